"""sharp_amd -- MI355X-native implementation of SHARP's ensemble random-projection +
meta-clustering hot path (reference: shibiaowan/SHARP, an R package).

Python mirror of the reference's exported R functions on that path; all compute runs
in libsharp_hip.so (hand-written HIP for gfx950) through its C ABI (include/sharp_hip.h)."""
from ._lib import SharpError, init, lib, reload_options, shutdown, so_path  # noqa: F401
from .api import *  # noqa: F401,F403
