"""ctypes loader for libsharp_hip.so -- the only compute backend of this package.

There is deliberately no fallback: if the HIP library is missing or no gfx950 device
is visible, every entry point raises (the product path must fail loudly, never route
through a CPU path)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsharp_hip.so")
_lib = None
_initialised_device = None


class SharpError(RuntimeError):
    """Raised where the reference would stop(): carries the library's message."""


def so_path():
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise SharpError(
                f"{_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C sharp_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
        try:  # torch wheels bundle their own libamdhip64: load it first so the process has ONE HIP runtime
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(_SO)
        _lib.sharp_last_error.restype = C.c_char_p
    return _lib


def check(rc, allow=0):
    """rc == 0 ok; warning bits in `allow` are returned to the caller; anything else raises."""
    if rc == 0:
        return 0
    if rc & ~allow == 0:
        return rc
    msg = lib().sharp_last_error()
    raise SharpError(msg.decode() if msg else f"libsharp_hip error {rc}")


def init(device=None):
    """Select the GPU (default: LOCAL_RANK or 0) and create the library's stream."""
    global _initialised_device
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _initialised_device != device:
        check(lib().sharp_init(int(device)))
        _initialised_device = device
    return device


def shutdown():
    """sharp_shutdown(): the streams of every slot are destroyed (workspaces stay); the next call initialises the library again."""
    global _initialised_device
    check(lib().sharp_shutdown())
    _initialised_device = None


def ensure_init():
    if _initialised_device is None:
        init()


def reload_options():
    """Ask the library to read its SHARP_* environment switches again (it reads them once, at first use)."""
    check(lib().sharp_reload_options())
