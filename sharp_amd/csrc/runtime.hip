// runtime.hip -- context, error string, profiling and device-memory helpers of the C ABI.
#include <cstring>
#include <ctime>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <pthread.h>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <sstream>
#include <tuple>
#include <vector>
#include <cstdio>

#include "common.hpp"

namespace sharp {

static thread_local std::string g_last_error;
static thread_local int t_slot = 0;
static Ctx g_ctxs[kMaxSlots];

void set_error(const std::string &msg) { g_last_error = msg; }

int cur_slot() { return t_slot; }
void bind_slot(int slot) {
    if (slot < 0 || slot >= kMaxSlots) throw Error(SHARP_ERR_ARG, "libsharp_hip: device slot out of range");
    t_slot = slot;
}
Ctx &ctx_unchecked() { return g_ctxs[t_slot]; }
Ctx &ctx() {
    Ctx &g_ctx = g_ctxs[t_slot];
    if (!g_ctx.ready)
        throw Error(SHARP_ERR_NO_DEVICE,
                    "libsharp_hip: no device context -- call sharp_init(device) first (a MI355X / gfx950 GPU is required; "
                    "there is no CPU fallback)");
    return g_ctx;
}

hipEvent_t Ctx::get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e;
    SHARP_HIP_CHECK(hipEventCreate(&e));
    return e;
}
hipStream_t Ctx::aux_stream(int i) {
    while (static_cast<int>(aux.size()) <= i) {
        hipStream_t s;
        // Streams of one priority share a small pool of hardware queues, dealt out by whatever else the process has created (after
        // a torch device-to-host copy, two "concurrent" ranges were measured running back to back: 110 ms per step instead of 83).
        // Streams of different priorities never share a queue, so neighbouring ranges alternate between the two classes.
        int lo = 0, hi = 0;
        SHARP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SHARP_HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, (aux.size() & 1) ? hi : 0));
        aux.push_back(s);
    }
    return aux[i];
}
void Ctx::resolve_pending() {
    if (pending.empty()) return;
    SHARP_HIP_CHECK(hipStreamSynchronize(stream));
    if (stream2) SHARP_HIP_CHECK(hipStreamSynchronize(stream2));
    for (hipStream_t s : aux) SHARP_HIP_CHECK(hipStreamSynchronize(s));
    for (auto &p : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &s = stats[p.name];
            s.ms += ms;
            s.launches += 1;
        }
        event_pool.push_back(p.a);
        event_pool.push_back(p.b);
    }
    pending.clear();
}

KernelTimer::KernelTimer(const char *n, hipStream_t stream) : name(n) {
    Ctx &c = ctx();
    if (!c.profiling) return;
    st = stream ? stream : c.stream;
    a = c.get_event();
    b = c.get_event();
    (void)hipEventRecord(a, st);
}
KernelTimer::~KernelTimer() {
    if (!a) return;
    Ctx &c = ctx_unchecked();
    (void)hipEventRecord(b, st);
    c.pending.push_back({a, b, name, st});
    if (c.pending.size() > 4096) {
        try { c.resolve_pending(); } catch (...) {}
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent host worker pool (host_parallel_for).  Created on first use and never destroyed: the workers sleep on a
// condition variable between jobs and die with the process.
// ---------------------------------------------------------------------------------------------
namespace {
struct HostPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *fn = nullptr;
    std::atomic<int> next{0};
    int n = 0, participants = 0, pending = 0;
    unsigned long generation = 0;
    std::exception_ptr failure;                           // the first exception of a job, rethrown on the caller's thread

    explicit HostPool(int nworkers) {
        for (int w = 0; w < nworkers; ++w) workers.emplace_back([this, w] { run(w); });
    }
    // A throwing item (bad_alloc in a resize, say) must not escape a worker (std::terminate inside the host R / Python process) nor
    // unwind the caller past the wait for the workers, which still hold `fn`: the first exception is kept, the remaining items are
    // skipped, everybody finishes, and parallel_for rethrows.
    void drain() {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) break;
            try { (*fn)(i); }
            catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                if (!failure) failure = std::current_exception();
                next.store(n, std::memory_order_relaxed);
            }
        }
    }
    void run(int w) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (w >= participants) continue;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    void parallel_for(int count, int max_threads, const std::function<void(int)> &f) {
        const int helpers = std::min<int>(static_cast<int>(workers.size()), std::min(max_threads, count) - 1);
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f; n = count; next.store(0, std::memory_order_relaxed);
            participants = helpers; pending = helpers;
            failure = nullptr;
            ++generation;
        }
        cv_work.notify_all();
        drain();                                          // the caller works too (never throws: see drain)
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
            fn = nullptr;
            err = failure;
            failure = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }
};
}  // namespace

namespace {
// One pool per device slot (the host tails of several GPUs -- one driver thread per slot in sharp_SHARP_unlimited_multi -- run side by
// side, and each slot has its one driver thread).  Pools, job locks and the table's own lock live behind pointers so that a forked
// child can start afresh: the child has the objects but not the threads, and a mutex that another thread of the parent held at
// fork() time would stay locked forever in the child.  Everything is created under the table lock: two slots' first calls may race.
struct PoolState { HostPool *pool = nullptr; std::mutex *one_job = nullptr; };
struct PoolTable { std::mutex *mu = new std::mutex; PoolState s[kMaxSlots]; bool atfork = false; };
PoolTable &pool_table() { static PoolTable *t = new PoolTable; return *t; }   // (never destroyed)
void pool_atfork_child() {
    PoolTable &T = pool_table();
    T.mu = new std::mutex;                                // the parent's objects are abandoned in the child (their threads do not exist there)
    for (PoolState &S : T.s) { S.pool = nullptr; S.one_job = nullptr; }
}
}  // namespace

// The cores this process may run on (its affinity mask: a GPU box hands a rank a share of the host; hardware_concurrency() counts the
// whole machine), capped by SHARP_HOST_THREADS -- bench.py sets that to cores / world size when it runs one process per GPU, so that eight
// ranks' upload pools, tail helpers and host loops together ask for the host's cores once, not eight times.
int host_cores() {
    static const int present = [] {
        cpu_set_t set;
        int c = 0;
        if (sched_getaffinity(0, sizeof set, &set) == 0) c = CPU_COUNT(&set);
        if (c <= 0) c = static_cast<int>(std::thread::hardware_concurrency());
        return c > 0 ? c : 4;
    }();
    const int cap = knobs().host_threads;
    return cap > 0 ? std::max(1, std::min(present, cap)) : present;
}

static thread_local int g_pool_threads_hint = 0;
void host_pool_threads_hint(int n) { g_pool_threads_hint = n; }

void host_parallel_for(int n, int max_threads, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    const unsigned hw = static_cast<unsigned>(host_cores());
    if (n == 1 || max_threads <= 1 || hw <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    PoolTable &T = pool_table();
    PoolState *S = nullptr;
    {
        std::lock_guard<std::mutex> lk(*T.mu);
        if (!T.atfork) { T.atfork = true; pthread_atfork(nullptr, nullptr, pool_atfork_child); }
        S = &T.s[cur_slot()];
        if (!S->one_job) S->one_job = new std::mutex;
        const unsigned want = g_pool_threads_hint > 0 ? static_cast<unsigned>(g_pool_threads_hint) : 15u;
        if (!S->pool) S->pool = new HostPool(static_cast<int>(std::min<unsigned>(hw - 1, want)));   // leaked on purpose
    }
    std::lock_guard<std::mutex> lk(*S->one_job);          // one job at a time per slot
    S->pool->parallel_for(n, max_threads, fn);
}

static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
// ---- the block cache behind DevBuf::alloc_pooled (common.hpp) --------------------------------------------------------------------
namespace {
struct BlockPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> blocks;     // (device, capacity) -> block
    size_t held = 0;
};
BlockPool &block_pool() { static BlockPool *P = new BlockPool; return *P; }   // (never destroyed: buffers of static objects come back late)
constexpr size_t kPoolMaxBlock = 64ull << 20, kPoolMaxHeld = 512ull << 20;
// four size classes per octave: a block serves requests down to 84 % of its capacity
size_t pool_class(size_t bytes) {
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    const size_t q = c / 8;
    for (size_t v = c / 2 + q; v < c; v += q) if (v >= bytes) return v;
    return c;
}
}  // namespace
namespace {
struct Parked { std::mutex mu; int scopes = 0; std::vector<void *> blocks; size_t bytes = 0; };
constexpr size_t kParkedMaxBytes = 1ull << 30;      // (overlapping windows on several GPUs keep the scope open: beyond 1 GB parked the block is freed at once)
Parked &parked() { static Parked *P = new Parked; return *P; }
}  // namespace
FreeLater::FreeLater() { Parked &P = parked(); std::lock_guard<std::mutex> lk(P.mu); ++P.scopes; }
FreeLater::~FreeLater() {
    std::vector<void *> gone;
    { Parked &P = parked(); std::lock_guard<std::mutex> lk(P.mu); if (--P.scopes == 0) { gone.swap(P.blocks); P.bytes = 0; } }
    for (void *p : gone) (void)hipFree(p);
}
bool free_later(void *p, size_t bytes) {
    if (bytes > (64ull << 20) || !knobs().free_later) return false;
    Parked &P = parked();
    std::lock_guard<std::mutex> lk(P.mu);
    if (P.scopes <= 0 || P.bytes + bytes > kParkedMaxBytes) return false;
    P.blocks.push_back(p);
    P.bytes += bytes;
    return true;
}
void *pool_take(size_t bytes, size_t *cap_bytes) {
    int dev = 0;
    SHARP_HIP_CHECK(hipGetDevice(&dev));
    const size_t cap = bytes <= kPoolMaxBlock ? pool_class(bytes) : bytes;
    *cap_bytes = cap;
    if (bytes <= kPoolMaxBlock) {
        BlockPool &P = block_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.blocks.find({dev, cap});
        if (it != P.blocks.end()) { void *p = it->second; P.blocks.erase(it); P.held -= cap; return p; }
    }
    void *p = nullptr;
    SHARP_HIP_CHECK(hipMalloc(&p, cap));
    return p;
}
void pool_give(void *p, size_t cap_bytes) {
    if (!p) return;
    int dev = 0;
    if (cap_bytes <= kPoolMaxBlock && hipGetDevice(&dev) == hipSuccess) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;   // (the block's own device, whoever hands it back)
        BlockPool &P = block_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.held + cap_bytes <= kPoolMaxHeld) { P.blocks.insert({{dev, cap_bytes}, p}); P.held += cap_bytes; return; }
    }
    (void)hipFree(p);
}
void pool_clear() {
    BlockPool &P = block_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto &kv : P.blocks) (void)hipFree(kv.second);
    P.blocks.clear(); P.held = 0;
}

HostTimer::HostTimer(const char *n) : name(n), t0(0), on(false) {
    Ctx &c = ctx_unchecked();
    if (c.ready && c.profiling) { on = true; t0 = now_s(); }
}
HostTimer::~HostTimer() {
    if (!on) return;
    Ctx &c = ctx_unchecked();
    auto &s = c.stats[std::string("host:") + name];
    s.ms += (now_s() - t0) * 1e3;
    s.launches += 1;
}

}  // namespace sharp

namespace sharp {
// The one place the library reads its environment: at the first use (sharp_init) and again only when sharp_reload_options() asks for it.
static Knobs read_knobs() {
    Knobs v;
    auto env = [](const char *name) { return getenv(name); };                    // (the library's only getenv outside lab_env())
    auto num = [&](const char *name, int dflt) { const char *e = env(name); return e && *e ? atoi(e) : dflt; };
    // what a user or a test of the product may want to set
    if (const char *kv = env("SHARP_RP_KERNEL")) v.rp_kernel = !strcmp(kv, "fused") ? 1 : !strcmp(kv, "dense") ? 2 : !strcmp(kv, "pc") ? 4 : 0;
    if (const char *xs = env("SHARP_X_STORAGE")) v.x_storage = !strcmp(xs, "fp32") ? 32 : !strcmp(xs, "fp64") ? 64 : 0;
    v.host_threads = num("SHARP_HOST_THREADS", 0);
    v.upload_threads = num("SHARP_UPLOAD_THREADS", 0);
    v.tail_threads = num("SHARP_TAIL_THREADS", 4);
    v.unlimited_batch = num("SHARP_UNLIMITED_BATCH", 1) != 0;
    v.unlimited_window_mb = num("SHARP_UNLIMITED_WINDOW_MB", 0);
    v.block_prefetch = num("SHARP_NO_BLOCK_PREFETCH", 0) == 0;
    v.decision_log = num("SHARP_DECISION_LOG", 0) != 0;
    v.host_group = num("SHARP_HOST_GROUP", 1);
    v.step_marks = num("SHARP_STEP_MARKS", 0) != 0;
    // cross-checks between the forms the library itself chooses from (tests/): sequential / one-launch / round-per-launch agglomeration, chunking,
    // the two statistics kernels, the incremental statistics, the host build of the projectors
    v.hc_seq = num("SHARP_HC_SEQ", 0) == 1;
    v.hc_mono = num("SHARP_HC_MONO", 0) == 1;
    v.hc_split = num("SHARP_HC_SPLIT", -1);
    v.hc_pipe = num("SHARP_HC_PIPE", 1) != 0;
    v.hc_chunk = num("SHARP_HC_CHUNK", 0);
    v.hc_first_chunk = num("SHARP_HC_FIRST_CHUNK", 0);
    v.hc_ranges = num("SHARP_HC_RANGES", 0);
    v.hc_nn_gemm = num("SHARP_HC_NN_GEMM", 1) != 0;
    v.stats_lane = num("SHARP_STATS_LANE", 1) != 0;
    v.ml_min_levels = num("SHARP_ML_MIN_LEVELS", 0);
    v.proj_host = num("SHARP_PROJ_HOST", 0) == 1;
    v.rp_pc_wgs = std::max(1, num("SHARP_RP_PC_WGS", 2));
#ifdef SHARP_LAB
    // lab builds (make LAB=1) only: alternative kernels and schedule experiments that were measured and not adopted (LAB_NOTES.md)
    if (const char *kv = env("SHARP_RP_KERNEL")) { if (!strcmp(kv, "sparse")) v.rp_kernel = 3; else if (!strcmp(kv, "split")) v.rp_kernel = 5; }
    v.rp_dual = num("SHARP_RP_DUAL", 1) != 0;
    { const int ser = num("SHARP_RP_SERIAL", -1); v.rp_two_streams = ser < 0 ? -1 : (ser == 0 ? 1 : 0); }
    v.rp_chunk = num("SHARP_RP_CHUNK", 0);
    v.rp_ahead = num("SHARP_RP_AHEAD", 2);
    v.rp_cp_wgs = std::max(1, num("SHARP_RP_CP_WGS", 8));
    v.rp_ap_wgs = std::max(1, num("SHARP_RP_AP_WGS", 4));
    v.rp_shape = num("SHARP_RP_SHAPE", 0);
    if (const char *ps = env("SHARP_RP_PC_SHAPE")) v.rp_pc_shape = (*ps == 'a' || *ps == 'A') ? 1 : (*ps == 'b' || *ps == 'B') ? 2 : 0;
    v.hc_wpt = num("SHARP_HC_WPT", 0);
    v.hc_finish_at = num("SHARP_HC_FINISH_AT", 15);
    v.mean_early = num("SHARP_MEAN_EARLY", 0) != 0;
    v.hc_prep_early = num("SHARP_HC_PREP_EARLY", 1) != 0;
    v.hc_tri = num("SHARP_HC_TRI", 0) != 0;
    v.hc_front = num("SHARP_HC_FRONT", 0);
    v.hc_half = num("SHARP_HC_HALF", 0) != 0;
    v.stats_sums = num("SHARP_STATS_SUMS", 1) != 0;
    v.dist_i8 = num("SHARP_DIST_I8", 0) != 0;
    v.free_later = num("SHARP_FREE_LATER", 1) != 0;
    v.front_overlap = num("SHARP_FRONT_OVERLAP", 0);
    v.tail_priority = num("SHARP_TAIL_PRIORITY", 1) != 0;
    v.gemm_slice = num("SHARP_GEMM_SLICE", 8);
#endif
    if (const char *dl = env("SHARP_DEVICES")) {
        for (const char *q = dl; *q;) {
            char *end = nullptr;
            const long dvc = strtol(q, &end, 10);
            if (end == q) break;
            v.devices.push_back(static_cast<int>(dvc));
            q = *end == ',' ? end + 1 : end;
        }
    }
    return v;
}
static Knobs &knobs_storage() { static Knobs k = read_knobs(); return k; }
const Knobs &knobs() { return knobs_storage(); }
void reload_knobs() { knobs_storage() = read_knobs(); }

namespace {
struct StepMarks { std::mutex mu; std::vector<std::tuple<const char *, int, double>> v; };
StepMarks &step_marks() { static StepMarks *M = new StepMarks; return *M; }
}  // namespace
void step_mark(const char *label, int id) {
    if (!knobs().step_marks) return;
    const double t = now_s();
    StepMarks &M = step_marks();
    std::lock_guard<std::mutex> lk(M.mu);
    M.v.emplace_back(label, id, t);
}
void step_marks_dump() {
    if (!knobs().step_marks) return;
    StepMarks &M = step_marks();
    std::lock_guard<std::mutex> lk(M.mu);
    if (M.v.empty()) return;
    const double t0 = std::get<2>(M.v.front());
    for (const auto &m : M.v) fprintf(stderr, "[step] %9.3f ms  %s %d\n", (std::get<2>(m) - t0) * 1e3, std::get<0>(m), std::get<1>(m));
    M.v.clear();
}
}  // namespace sharp

namespace sharp {
// The slot a worker of the in-process multi-GPU run uses: keyed on (device, which occurrence of that device in the caller's list, role),
// so that a later call with another device list -- {0, 1} then {1, 0}, or the tests' {0, 0} followed by a real {0, 1} -- finds the slots
// whose workspaces already live on the right GPU instead of rebinding one (a slot never changes its device).  Slot 0 is the caller's.
int acquire_slot(int device, int occurrence, int role) {
    static std::mutex mu;
    static int key[kMaxSlots][3];
    static int used = 1;
    std::lock_guard<std::mutex> lk(mu);
    for (int s = 1; s < used; ++s)
        if (key[s][0] == device && key[s][1] == occurrence && key[s][2] == role) return s;
    if (used >= kMaxSlots)
        throw Error(SHARP_ERR_ARG, "libsharp_hip: out of device slots (" + std::to_string(kMaxSlots - 1) + " worker contexts in one process)");
    key[used][0] = device; key[used][1] = occurrence; key[used][2] = role;
    return used++;
}
// fn() once for every initialised slot, the calling thread bound to it and to its device; the caller's slot and device are restored
void for_each_ready_slot(const std::function<void()> &fn) {
    const int keep = cur_slot();
    int keep_dev = -1;
    (void)hipGetDevice(&keep_dev);
    std::exception_ptr err;
    for (int s = 0; s < kMaxSlots; ++s) {
        bind_slot(s);
        Ctx &c = ctx_unchecked();
        if (!c.ready) continue;
        try {
            SHARP_HIP_CHECK(hipSetDevice(c.device));
            fn();
        } catch (...) { if (!err) err = std::current_exception(); }
    }
    bind_slot(keep);
    if (keep_dev >= 0) (void)hipSetDevice(keep_dev);
    if (err) std::rethrow_exception(err);
}
void init_slot(int slot, int device, bool high_priority) {
    bind_slot(slot);
    Ctx &c = ctx_unchecked();
    if (c.ready && c.device == device) { SHARP_HIP_CHECK(hipSetDevice(device)); return; }   // (the device is a per-thread setting)
    // every workspace a slot keeps between calls (distance matrices, E, pinned stages, events, projector handles) lives on the device
    // the slot was first initialised on: a slot never changes its device (sharp_shutdown() destroys the streams, not the workspaces)
    static int bound_device[kMaxSlots];
    static bool bound[kMaxSlots];
    if (bound[slot] && bound_device[slot] != device)
        throw Error(SHARP_ERR_ARG, "sharp_init: this slot is bound to device " + std::to_string(bound_device[slot]) +
                                   " (its workspaces live there); one device per slot -- several GPUs in one process go through "
                                   "sharp_SHARP_unlimited_multi, or start one process per GPU");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        throw Error(SHARP_ERR_NO_DEVICE, "libsharp_hip: no HIP device visible (MI355X / gfx950 required; no CPU fallback)");
    SHARP_REQUIRE(device >= 0 && device < n, "sharp_init: device index out of range");
    SHARP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SHARP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        throw Error(SHARP_ERR_NO_DEVICE, std::string("libsharp_hip is built for gfx950 only; device reports ") + prop.gcnArchName);
    if (c.ready && c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    if (c.ready && c.stream2) { (void)hipStreamDestroy(c.stream2); c.stream2 = nullptr; }
    for (hipStream_t s : c.aux) (void)hipStreamDestroy(s);
    c.aux.clear();
    int lo = 0, hi = 0;
    SHARP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    // (a tail helper's small kernels go ahead of the pipeline's GEMM workgroups when a CU frees up: they are what a finished block waits for)
    if (high_priority) SHARP_HIP_CHECK(hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi));
    else SHARP_HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    c.main_stream = c.stream;
    // the second stream of the RP stage: its own priority class, hence its own hardware queue (see aux_stream)
    SHARP_HIP_CHECK(hipStreamCreateWithPriority(&c.stream2, hipStreamNonBlocking, hi));
    c.device = device;
    bound_device[slot] = device;
    bound[slot] = true;
    c.num_cu = prop.multiProcessorCount;
    c.lds_per_block = prop.sharedMemPerBlock;
    c.ready = true;
}
}  // namespace sharp

using namespace sharp;

extern "C" {

const char *sharp_last_error(void) { return g_last_error.c_str(); }
int sharp_version(void) { return 100; }

int sharp_device_count(int *count) {
    SHARP_API_BEGIN
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *count = n;
    SHARP_API_END
}

int sharp_init(int device) {
    SHARP_API_BEGIN
    init_slot(cur_slot(), device);          // (slot 0 unless this thread was bound elsewhere)
    SHARP_API_END
}

int sharp_shutdown(void) {
    SHARP_API_BEGIN
    for_each_ready_slot([] {                              // the caller's slot and every worker slot of a multi-GPU run
        Ctx &c = ctx_unchecked();
        (void)hipStreamSynchronize(c.stream);
        drop_pending_front();
        drain_side_streams();
        for (auto &p : c.pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        c.pending.clear();
        for (auto e : c.event_pool) (void)hipEventDestroy(e);
        c.event_pool.clear();
        (void)hipStreamDestroy(c.stream);
        c.stream = nullptr;
        if (c.stream2) { (void)hipStreamSynchronize(c.stream2); (void)hipStreamDestroy(c.stream2); c.stream2 = nullptr; }
        for (hipStream_t s : c.aux) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        c.aux.clear();
        c.ready = false;
    });
    pool_clear();
    SHARP_API_END
}

int sharp_reload_options(void) {
    SHARP_API_BEGIN
    reload_knobs();
    SHARP_API_END
}

int sharp_synchronize(void) {
    SHARP_API_BEGIN
    stream_sync();
    drain_side_streams();                                // (the next block's front, the ensemble mean: they may still read caller buffers)
    if (ctx().stream2) SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream2));
    SHARP_API_END
}

int sharp_profile_enable(int on) {
    SHARP_API_BEGIN
    ctx().profiling = on != 0;
    SHARP_API_END
}
int sharp_profile_reset(void) {
    SHARP_API_BEGIN
    Ctx &c = ctx();
    c.resolve_pending();
    c.stats.clear();
    SHARP_API_END
}
int sharp_profile_get(const char *name, double *total_ms, long long *launches) {
    SHARP_API_BEGIN
    Ctx &c = ctx();
    c.resolve_pending();
    auto it = c.stats.find(name);
    *total_ms = it == c.stats.end() ? 0.0 : it->second.ms;
    *launches = it == c.stats.end() ? 0 : it->second.launches;
    SHARP_API_END
}
int sharp_profile_dump(char *buf, int buflen) {
    SHARP_API_BEGIN
    Ctx &c = ctx();
    c.resolve_pending();
    std::ostringstream os;
    for (auto &kv : c.stats) os << kv.first << ' ' << kv.second.ms << ' ' << kv.second.launches << '\n';
    std::string s = os.str();
    SHARP_REQUIRE(buflen > 0, "sharp_profile_dump: empty buffer");
    size_t n = s.size() < (size_t)buflen - 1 ? s.size() : (size_t)buflen - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    SHARP_API_END
}

int sharp_dev_alloc(long long bytes, void **dptr) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(bytes >= 0, "sharp_dev_alloc: negative size");
    SHARP_HIP_CHECK(hipMalloc(dptr, (size_t)bytes));
    SHARP_API_END
}
int sharp_dev_free(void *dptr) {
    SHARP_API_BEGIN
    ctx();
    SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    SHARP_HIP_CHECK(hipFree(dptr));
    SHARP_API_END
}
int sharp_dev_upload(void *dptr, const void *host, long long bytes) {
    SHARP_API_BEGIN
    SHARP_HIP_CHECK(hipMemcpyAsync(dptr, host, (size_t)bytes, hipMemcpyHostToDevice, ctx().stream));
    stream_sync();
    SHARP_API_END
}
int sharp_dev_download(void *host, const void *dptr, long long bytes) {
    SHARP_API_BEGIN
    SHARP_HIP_CHECK(hipMemcpyAsync(host, dptr, (size_t)bytes, hipMemcpyDeviceToHost, ctx().stream));
    stream_sync();
    SHARP_API_END
}

}  // extern "C"
