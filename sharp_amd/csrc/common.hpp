// common.hpp -- runtime plumbing shared by every translation unit of libsharp_hip.so:
// error reporting across the C ABI, the device context (one HIP stream per process),
// RAII device buffers and per-kernel HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sharp_hip.h"

namespace sharp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_error(const std::string &msg);

#define SHARP_HIP_CHECK(expr)                                                                   \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw sharp::Error(SHARP_ERR, std::string(#expr) + ": " + hipGetErrorString(_e) +   \
                                              " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define SHARP_REQUIRE(cond, msg)                                  \
    do {                                                          \
        if (!(cond)) throw sharp::Error(SHARP_ERR_ARG, (msg));    \
    } while (0)

#define SHARP_API_BEGIN try {
#define SHARP_API_END                                        \
    }                                                        \
    catch (const sharp::Error &e) {                          \
        sharp::set_error(e.what());                          \
        return e.code;                                       \
    }                                                        \
    catch (const std::exception &e) {                        \
        sharp::set_error(e.what());                          \
        return SHARP_ERR;                                    \
    }                                                        \
    return SHARP_OK;

// An expression block resident in HBM: genes x cells, column-major (a cell is a contiguous vector of ld values), stored as fp32
// (counts and every other fp32-exact input: half the bytes of the one pass over X) or as fp64 (TPM / CPM-like doubles, which fp32
// would perturb by 6e-8 relative: the reference computes log2(X + 1) and the projection in double, R/SHARP.R:343-345).
struct XRef {
    const void *p = nullptr;
    bool f64 = false;
    // fixed-point bits of the RP accumulation in log mode: 44 while |log2(1 + x)| < 128 (every fp32 value); an fp64 block with values
    // beyond FLT_MAX gets 41 (|log2(1 + x)| < 1024, < 2^11 terms per sum)
    int log_fix_bits = 44;
    XRef() = default;
    XRef(const float *q) : p(q), f64(false) {}      // (implicit: the *_dev entry points take fp32 blocks)
    XRef(const double *q) : p(q), f64(true) {}
    const float *f32() const { return static_cast<const float *>(p); }
    const double *d64() const { return static_cast<const double *>(p); }
};

// Tuning switches, read from the environment ONCE (the first sharp_init) and kept: the kernels' hosts never call getenv per launch.
struct Knobs {
    bool rp_dual = true;        // SHARP_RP_DUAL=0: signed row-list codes even where two accumulator arrays would fit
    int rp_two_streams = -1;    // SHARP_RP_SERIAL=0 / 1: the compaction of chunk c + 1 on a second stream beside the apply of chunk c: always / never (default: by kernel form)
    int rp_ahead = 2;           // SHARP_RP_AHEAD: the compaction of the block beside the per-call projector build: 0 not at all, 1 behind the draw kernel
                                // (beside the packing of the row lists), 2 from the start (beside the draw kernel too)
    int rp_chunk = 0;           // SHARP_RP_CHUNK: cells per chunk of the RP stage (0: sized by the library)
    int rp_cp_wgs = 8;          // SHARP_RP_CP_WGS / SHARP_RP_AP_WGS: workgroups per CU of the two RP kernels (upper bounds)
    int rp_ap_wgs = 4;
    int rp_kernel = 0;          // SHARP_RP_KERNEL: "fused" (1) the single-kernel RP form, "dense" (2) the MFMA form, "sparse" (3) never the dense form,
                                // "pc" (4) always the producer / consumer kernel (rp3.hip), "split" (5) always the two-kernel form (rp2.hip)
    int rp_pc_wgs = 2;          // SHARP_RP_PC_WGS: workgroups per CU of the producer / consumer kernel
    int rp_pc_shape = 0;        // SHARP_RP_PC_SHAPE=a / b: its workgroup shape (rp3.hip: 8 waves, 2 producers / 12 waves, 4 producers; default by row-list width)
    int x_storage = 0;          // SHARP_X_STORAGE=fp32 / fp64: force the storage of uploaded blocks (0: fp32 when exact, else fp64)
    bool block_prefetch = true; // SHARP_NO_BLOCK_PREFETCH=1: a block's front is not prepared under the previous block's tail
    bool unlimited_batch = true;   // SHARP_UNLIMITED_BATCH=0: SHARP_unlimited block after block instead of one pipelined batch per window
    int unlimited_window_mb = 0;   // SHARP_UNLIMITED_WINDOW_MB: projections per window of that batch (0: 16 GB)
    int ml_min_levels = 0;      // SHARP_ML_MIN_LEVELS: candidate levels from which the incremental statistics take over (0: 256)
    bool hc_mono = false, hc_seq = false;   // SHARP_HC_MONO=1 / SHARP_HC_SEQ=1 (cross-checks): one-launch agglomeration / sequential kernel only
    int hc_split = -1;          // SHARP_HC_SPLIT=1 / 0: round-per-launch agglomeration forced on / off (-1: by task count)
    int hc_ranges = 0, hc_wpt = 0, hc_finish_at = 15, hc_chunk = 0;   // SHARP_HC_RANGES / _WPT / _FINISH_AT / _CHUNK: ranges per chunk, workgroups per task, finishing round, tasks per chunk
    int tail_threads = 4;       // SHARP_TAIL_THREADS: host threads (each with its own slot) that run the blocks' tails of a batched SHARP_unlimited window;
                                // 0: from the batch's progress callback on the calling thread, one after the other (the round-3 form)
    bool tail_priority = true;  // SHARP_TAIL_PRIORITY=0: the helpers' streams in the normal priority class
    bool dist_i8 = false;       // SHARP_DIST_I8=1: the correlation-distance GEMM on the integer matrix cores (gemm_i8.hip) instead of the fp64 MFMA
    bool stats_sums = true;     // SHARP_STATS_SUMS=0: the finest level's cluster sums as a skinny GEMM over a one-hot matrix
    bool stats_lane = true;     // SHARP_STATS_LANE=0: stats_kernel (the walk re-read from LDS per cell, bitonic median) where stats_lane_kernel runs by default (cross-check)
    bool hc_tri = false;        // SHARP_HC_TRI=1: the upper-triangle agglomeration kernel (hclust_tri.inc: 44 % of the HBM bytes, the same time alone,
                                // 8 % slower inside the batched SHARP_unlimited pipeline) where the full-matrix one runs by default
    bool hc_prep_early = true;  // SHARP_HC_PREP_EARLY=0: a chunk's row preparation behind the previous chunk's distance GEMM instead of beside it
    bool hc_pipe = true;        // SHARP_HC_PIPE=0: one chunk of base-clustering tasks at a time
    bool mean_early = false;    // SHARP_MEAN_EARLY=1: the ensemble mean of SHARP_large behind the LAST agglomeration, beside the last statistics, instead of behind them (measured: 53.2 against 51.7 ms per cfg2 step, four interleaved pairs: the statistics slow down by more than the mean takes)
    int gemm_slice = 8;         // SHARP_GEMM_SLICE: workgroups per CU per slice of a distance GEMM prepared under another block's tail
    bool proj_host = false;     // SHARP_PROJ_HOST=1 (cross-check): the host build of the projectors
    int hc_front = 0;           // SHARP_HC_FRONT=c: the first c agglomeration rounds without rewriting the matrix (hclust_front.inc; an experiment)
    bool hc_nn_gemm = true;     // SHARP_HC_NN_GEMM=0: the agglomeration's first round scans the distance matrix instead of reading the GEMM's per-tile partials
    bool hc_half = false;       // SHARP_HC_HALF=1: the eight-wave agglomeration kernel (half a CU per task) also when there is at most one task per CU
    int host_threads = 0;       // SHARP_HOST_THREADS: cap on the host cores this process sizes its pools from (0: its affinity mask); host_cores()
    int upload_threads = 0;     // SHARP_UPLOAD_THREADS: host threads narrowing / copying an uploaded block (0: up to 32)
    std::vector<int> devices;   // SHARP_DEVICES=0,1,2,...: the GPUs sharp_SHARP_unlimited deals a list of blocks to (empty / one: the caller's device)
    int front_overlap = 0;      // SHARP_FRONT_OVERLAP=b (an experiment): in a batched SHARP_unlimited window the blocks from the b-th on are projected with ONE workgroup per CU and
                                // the chunks of base tasks wait for their own blocks' projections only, so that the first distance GEMM runs beside the later blocks' RP kernels
    int hc_first_chunk = 0;     // SHARP_HC_FIRST_CHUNK=n: tasks in the first chunk of a pipelined batch (0: by the library, -1: equal chunks)
    int host_group = 1;         // SHARP_HOST_GROUP=2..3: host blocks of a SHARP_unlimited list go in groups of up to this many as one pipelined batch instead of block after block
                                // (measured on cfg3's ten blocks: a pair as one batch takes 40-50 ms, two blocks one after the other -- each prepared under the other's tail -- 2 x 25: no gain; default 1)
    bool decision_log = false;  // SHARP_DECISION_LOG=1: every get_opt_hclust call leaves a row in the decision log from the start (sharp_last_decisions)
    bool free_later = true;     // SHARP_FREE_LATER=0: a buffer that grows inside a batched SHARP_unlimited window is freed at once (hipFree drains the device) instead of when the window ends
    bool step_marks = false;    // SHARP_STEP_MARKS=1: host timestamps of a SHARP_unlimited call's milestones (chunks fetched, blocks' tails, merge) on stderr at its end
    int rp_shape = 0;           // SHARP_RP_SHAPE=1: 8 lanes x 4 slots per gene where the default is 16 x 2 (A/B runs)
};
const Knobs &knobs();
void reload_knobs();
// SHARP_STEP_MARKS=1: (label, id, time) marks of the running call, from any of its threads; step_marks_dump() prints them, in ms from the first one, and clears them
void step_mark(const char *label, int id = -1);
void step_marks_dump();
#ifdef SHARP_LAB
inline const char *lab_env(const char *name) { return getenv(name); }   // lab builds (tools/build_variant.sh) only: ablation / timing switches
#endif

struct KernelStat {
    double ms = 0;
    long long launches = 0;
};

struct Ctx {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t main_stream = nullptr;   // `stream` outside every StreamScope
    int rp_wgs_cap = 0;                  // > 0: the producer / consumer RP kernel with at most this many workgroups per CU (a block projected beside the first chunk's distance GEMM)
    bool polite = false;                 // work enqueued now is a block prepared ahead of time under another block's tail (SHARP_unlimited):
                                         // long launches go out in slices, nothing goes to the high-priority second stream
    hipStream_t stream2 = nullptr;   // side stream: producer kernels that overlap with consumers on `stream`
    std::vector<hipStream_t> aux;    // extra streams for pipelined task ranges (created on demand, see aux_stream())
    hipStream_t aux_stream(int i);
    int num_cu = 256;
    size_t lds_per_block = 65536;
    bool profiling = false;
    std::map<std::string, KernelStat> stats;
    // pending (start, stop, name) event triples, resolved lazily at sync points
    struct Pending { hipEvent_t a, b; std::string name; hipStream_t st; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t get_event();
    void resolve_pending();
};
Ctx &ctx();          // throws SHARP_ERR_NO_DEVICE if sharp_init() has not succeeded
Ctx &ctx_unchecked();

// Device slots.  Everything the library keeps between calls -- context, streams, workspaces, projector handles, the front prepared for
// the next call -- exists once per SLOT, and a host thread works on the slot it is bound to (slot 0 unless told otherwise: the one
// sharp_init() sets up, which is all a single-GPU host ever sees).  The multi-GPU entry points (sharp_SHARP_unlimited_multi) start one
// host thread per device and bind each to a slot of its own, so several GPUs -- or, in the tests, several slots on ONE GPU -- run
// side by side in one process.
constexpr int kMaxSlots = 200;       // the caller's slot and its tail helpers' (up to 8) + (compute, upload, up to 8 tail helpers) of up to 16 GPUs, with room for repeated devices
int cur_slot();
void bind_slot(int slot);                   // the calling thread works on this slot from now on
void init_slot(int slot, int device, bool high_priority = false);   // (high_priority: the slot's main stream in the high class, tail helpers)
                                            // bind_slot + hipSetDevice + the slot's context (streams) on that device; idempotent per (slot, device)
int acquire_slot(int device, int occurrence, int role);   // the worker slot (>= 1) of that device / occurrence in the device list / role (0 compute, 1 upload, 2 tail helper occurrence % 4 of slot occurrence / 4)
void for_each_ready_slot(const std::function<void()> &fn);   // fn() with the calling thread bound to each initialised slot in turn
// the per-slot instance of a keep-between-calls object: `T &name() { return per_slot<T>(); }`
template <typename T>
T &per_slot() { static T w[kMaxSlots]; return w[cur_slot()]; }

// Times everything enqueued on the library stream during its lifetime (if profiling is on).
struct KernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    hipStream_t st = nullptr;
    explicit KernelTimer(const char *n, hipStream_t stream = nullptr);   // default: the library's main stream
    ~KernelTimer();
};

// Wall-clock time of a host-side section, reported through the same table with a "host:" prefix.
struct HostTimer {
    const char *name;
    double t0;
    bool on;
    explicit HostTimer(const char *n);
    ~HostTimer();
};

// A small cache of device blocks for buffers that are allocated and released on EVERY call (the per-call projectors of SHARP() and the
// temporaries of their build: twelve hipFree per call, each of which synchronises the device: 0.37 ms per cfg2 step).  Opt-in
// (DevBuf::alloc_pooled): a block handed back is handed out again without any synchronisation, so the owner must have synchronised
// the stream(s) that used it -- which is what every user of these buffers did before releasing them anyway.  Blocks up to 64 MB, at most
// 512 MB held per process; sharp_trim() / sharp_shutdown() empty it.
// hipFree drains the whole device.  While a pipelined SHARP_unlimited window runs (FreeLater scope: the device is busy until its last
// agglomeration ends) a DevBuf that lets go of a block below 64 MB parks it instead (at most 1 GB parked: beyond that it is freed at once); the parked blocks are freed when the last such scope ends.
struct FreeLater { FreeLater(); ~FreeLater(); FreeLater(const FreeLater &) = delete; FreeLater &operator=(const FreeLater &) = delete; };
bool free_later(void *p, size_t bytes);                // true: parked (a scope is open and the block is small); false: the caller frees it
void *pool_take(size_t bytes, size_t *cap_bytes);      // a cached block of the current device with capacity >= bytes (its size class), or a fresh one
void pool_give(void *p, size_t cap_bytes);
void pool_clear();

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t pooled_cap = 0;               // bytes of the pool block behind p (0: a plain hipMalloc)
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), pooled_cap(o.pooled_cap) { o.p = nullptr; o.n = 0; o.pooled_cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; pooled_cap = o.pooled_cap; o.p = nullptr; o.n = 0; o.pooled_cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) SHARP_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
        if (count) step_mark("        hipMalloc (after a hipFree if the buffer grew), KB", static_cast<int>(count * sizeof(T) >> 10));
    }
    void alloc_pooled(size_t count) {    // (see pool_take)
        release();
        n = count;
        if (count) p = static_cast<T *>(pool_take(count * sizeof(T), &pooled_cap));
    }
    // A buffer that has to grow is freed and allocated anew, and hipFree drains the WHOLE device: from a tail helper of a batched
    // SHARP_unlimited window that wait lasts until the pipeline's last agglomeration is over (a 524 KB label buffer kept block 2's tail,
    // and its helper thread, for 62 of a call's 172 ms: SHARP_STEP_MARKS).  Sizes that follow the data (clusters per fold, columns of
    // a soft matrix) differ by a few per cent from block to block, so a buffer below 64 MB grows by half again and stops growing.
    void ensure(size_t count) {
        if (count <= n) return;
        size_t want = count;
        if (n && count * sizeof(T) <= (64ull << 20)) want = count + count / 2;
        alloc(want);
    }
    void release() {
        if (!p) return;
        if (pooled_cap) pool_give(p, pooled_cap); else if (!free_later(p, n * sizeof(T))) (void)hipFree(p);
        p = nullptr; n = 0; pooled_cap = 0;
    }
    void upload(const T *h, size_t count) {
        SHARP_HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
    }
    void download(T *h, size_t count) const {
        SHARP_HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
        SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    }
    void zero() { if (n) SHARP_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), ctx().stream)); }
};

// Makes `s` the library's current stream for the lifetime of the object: every launch helper, DevBuf transfer and
// KernelTimer below it then works on `s` (used to pipeline independent task ranges on several streams).
struct StreamScope {
    hipStream_t saved;
    explicit StreamScope(hipStream_t s) : saved(ctx().stream) { ctx().stream = s; }
    ~StreamScope() { ctx_unchecked().stream = saved; }
    StreamScope(const StreamScope &) = delete;
    StreamScope &operator=(const StreamScope &) = delete;
};

// driver.hip: the side streams of the block pipeline (next block's front, ensemble mean) and the front prepared for the next call
void drain_side_streams();
void drop_pending_front();

inline void stream_sync() { SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream)); }

// fn(0) .. fn(n-1) on the calling thread plus up to max_threads - 1 workers of a persistent pool (items handed out dynamically).
// For the short host loops between kernels (per-fold relabelling and votes): starting std::threads anew cost more than the loops.
// fn must not call HIP and must not throw.
int host_cores();                   // cores this process may use (affinity mask), capped by SHARP_HOST_THREADS
void host_pool_threads_hint(int n);   // the calling thread's slot gets a pool of n workers if its pool does not exist yet (tail helpers: several run side by side)
void host_parallel_for(int n, int max_threads, const std::function<void(int)> &fn);
inline void launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(SHARP_ERR, std::string("launch of ") + what + " failed: " + hipGetErrorString(e));
}

}  // namespace sharp
