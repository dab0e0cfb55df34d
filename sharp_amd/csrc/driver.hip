// driver.hip -- the drivers of the hot path, one library call per reference entry point so that no
// foreach/%dopar% fork ever touches the HIP context (SURVEY.md 8b "Threading"):
//   SHARP()           R/SHARP.R:44-318    defaults + dispatch                       (row a12)
//   SHARP_small()     R/SHARP.R:339-454   K projections of all cells + wMetaC       (row a7)
//   SHARP_large()     R/SHARP.R:478-851   shuffle, 2000-cell folds, K*T tasks as ONE batched launch
//                                          sequence, per-fold wMetaC, cross-fold sMetaC   (row a8)
//   SHARP_unlimited() R/SHARP_unlimited.R:29-242  per-block SHARP + centroid-level sMetaC (row a11)
// X stays resident in HBM as fp32 (genes x cells, column-major); E, viE and every clustering
// intermediate stay on the device; only labels and a few KB of selection statistics cross PCIe.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <ctime>
#include <future>
#include <mutex>
#include <vector>

#include "meta.hpp"
#include "projector.hpp"
#include "rrng.hpp"
#include "upload.hpp"

namespace sharp {

struct SharpArgs {
    int K = 0, reduced_ndim = 0, base_ncells = 0, partition_ncells = 0, hmethod = 0;
    int N_cluster = 0, enpN = 0, indN = 0, minN = 0, maxN = 0;
    double sil_thre = -1, height_Ntimes = 0;
    int flag = 1;            // log-transform
    int projector = 0;       // handle of a shared rM list, 0 = draw from rN_seed
    double rN_seed = 0.5;    // 0.5 = the reference's "unseeded" sentinel
    bool want_viE = false, want_x0 = false;
    // the caller wants viE on the HOST (forview = TRUE, R/SHARP.R:46,844): SHARP_large then sends it on its way -- un-shuffle, copy into a pinned
    // staging block on the mean stream -- as soon as the ensemble mean exists, under the per-fold wMetaC and the sMetaC, instead of behind them
    bool view_to_host = false;
    // SHARP_fpart (R/SHARP_unlimited2.R:297-544): the large path with log10 (flag = 2), E1 rounded to one decimal before
    // clustering, maxN.cluster = 40 for the base tasks, and NO sMetaC: the per-fold ensemble labels go up to the caller
    bool fpart = false;
    int block = 0;           // the block's index in its SHARP_unlimited call (decision log only)
    // SHARP_unlimited with several blocks on this GPU: the block that comes next (same genes, projector, parameters).  Its projection,
    // row preparation and distance matrices are enqueued on a side stream before this block's agglomeration starts, so that they run
    // under this block's (HBM-bound, then host-bound) tail; the next call finds them done.
    XRef next_dX; long long next_n = 0, next_ld = 0;
};
struct SharpOut {
    std::vector<int> pred;           // 1..G, numbered by first appearance (R/SHARP.R:429-443,828-843)
    int n_pred = 0;
    // n x p, ORIGINAL cell order, on the device: a view of a workspace kept between calls (valid until the next call)
    struct View {
        double *p = nullptr;
        const double *staged = nullptr;      // pinned host copy on its way (SharpArgs::view_to_host), complete when `staged_done` has happened
        hipEvent_t staged_done = nullptr;
        void download(double *h, size_t count) const {
            if (staged) {
                SHARP_HIP_CHECK(hipEventSynchronize(staged_done));
                const size_t piece = (count + 15) / 16;
                const double *src = staged;
                host_parallel_for(16, 16, [&](int t) {       // (one thread copies ~10 GB/s out of pinned memory)
                    const size_t lo = std::min(count, piece * static_cast<size_t>(t)), hi = std::min(count, lo + piece);
                    if (hi > lo) memcpy(h + lo, src + lo, (hi - lo) * sizeof(double));
                });
                return;
            }
            SHARP_HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
            SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream));
        }
    } viE;
    std::vector<double> x0;          // n x x0_cols column-major
    int x0_cols = 0;
    int p = 0, K = 0, path = 0, rc = 0;
};

namespace {

// big per-call device buffers are kept between calls (hipMalloc/hipFree of multi-GB buffers costs up to tens of ms)
struct DriverWs {
    DevBuf<double> E, Eb, viE_sh, viE_out;       // Eb / posb: the block prepared ahead of time
    DevBuf<int> pos, posb;
    hipEvent_t mean_go = nullptr, mean_done = nullptr;
    hipStream_t mean_stream = nullptr;
    DevBuf<double> Ebatch;                       // Ebatch / posbatch: all blocks of a batched SHARP_unlimited window
    DevBuf<int> posbatch;
    DevBuf<double> fold_means, block_means;      // grow-only: a hipFree in a block's tail would wait for every stream
    double *view_pinned = nullptr;               // staging block of the early viE download (SharpArgs::view_to_host)
    size_t view_pinned_n = 0;
    hipEvent_t view_done = nullptr;
    std::vector<hipEvent_t> front_ev;            // SHARP_FRONT_OVERLAP: one event per block of a batched window (block q's projection is complete)
};
DriverWs &dws() { return per_slot<DriverWs>(); }

// allrpinfo of the last SHARP_small run (R/SHARP.R:350-387,446): the colour index of every cell under every random projection; the
// projections themselves (indE) are still in dws().E until the next SHARP call
struct LastSmall { bool valid = false; int n = 0, K = 0, p = 0; long long ldE = 0; std::vector<int> enrp; };
LastSmall &last_small() { return per_slot<LastSmall>(); }

inline bool lex_less_id(int a, int b) {
    char sa[16], sb[16];
    snprintf(sa, sizeof sa, "%d", a);
    snprintf(sb, sizeof sb, "%d", b);
    return strcmp(sa, sb) < 0;
}

// clusters with < 10 cells are merged into the one with the smallest id among them
// (R/SHARP.R:418-427,816-825; R/SHARP_unlimited.R:168-177)
void merge_small(std::vector<int> &lab) {
    int mx = 0;
    for (int v : lab) mx = std::max(mx, v);
    std::vector<long long> cnt(mx + 1, 0);
    for (int v : lab) ++cnt[v];
    int mn = -1;
    for (int q = 1; q <= mx; ++q) if (cnt[q] > 0 && cnt[q] < 10) { mn = q; break; }
    if (mn >= 0) for (int &v : lab) if (cnt[v] < 10) v = mn;
}
int relabel_first(std::vector<int> &lab) {
    int mx = 0;
    for (int v : lab) mx = std::max(mx, v);
    std::vector<int> map(mx + 1, 0);
    int k = 0;
    for (int &v : lab) { if (!map[v]) map[v] = ++k; v = map[v]; }
    return k;
}

std::shared_ptr<Projector> projector_for(const SharpArgs &a, int m, int p, int K, const std::function<void()> *after_draw = nullptr) {
    if (a.projector) {
        auto pr = get_projector(a.projector);
        SHARP_REQUIRE(pr->m == m && pr->p == p && pr->K >= K, "rM does not match the data (genes, reduced.ndim, ensize.K)");
        return pr;
    }
    std::vector<double> seeds(K);
    for (int k = 0; k < K; ++k) seeds[k] = (a.rN_seed == 0.5) ? 0.5 : 50 + a.rN_seed + (k + 1);   // R/SHARP.R:360,545
    return build_projector(m, p, K, seeds.data(), after_draw);
}

// R's round(x, 1) (nmath/fround.c, R >= 4.0.0, restated; R core is an unpinned third-party dependency of the reference):
// the closer of floor(10 x)/10 and ceil(10 x)/10 as doubles, the even multiple on a tie; used by R/SHARP_unlimited2.R:410
__host__ __device__ inline double r_round1(double x) {
    if (!(x == x) || x == 0.0 || x - x != 0.0) return x;            // NaN, 0, +-Inf
    const double sgn = x < 0.0 ? -1.0 : 1.0;
    x = x < 0.0 ? -x : x;
    if (x >= 1e14) return sgn * x;                                    // nothing to round beyond DBL_DIG
    const double x10 = 10.0 * x, i10 = floor(x10), xd = i10 / 10.0, xu = ceil(x10) / 10.0;
    const double du = xu - x, dd = x - xd;
    return sgn * ((dd < du || (dd == du && fmod(i10, 2.0) == 0.0)) ? xd : xu);
}
__global__ void round1_kernel(double *__restrict__ E, long long count) {
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < count; q += static_cast<long long>(gridDim.x) * blockDim.x)
        E[q] = r_round1(E[q]);
}

inline int colour_of(int j) { return j > 40 ? ((j - 1) % 40) + 1 : j; }   // R/getrowColor.R:59-68

__global__ void gather_rows_kernel(const double *__restrict__ src, const int *__restrict__ row_of, long long n, int p, double *__restrict__ dst) {
    const long long tot = n * p;
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < tot; q += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long i = q / p;
        dst[q] = src[static_cast<long long>(row_of[i]) * p + (q - i * p)];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// SHARP_small (R/SHARP.R:339-454)
// ---------------------------------------------------------------------------------------------
void sharp_small_dev(XRef dX, int m, int n, long long ld, const SharpArgs &a, int K, int p, HcParams base, SharpOut &out) {
    last_small().valid = false;                          // E is about to be reallocated / overwritten: valid again only once this call has completed
    auto pr = projector_for(a, m, p, K);
    const long long ldE = static_cast<long long>(pr->K) * p;
    DevBuf<double> &E = dws().E;
    E.ensure(static_cast<size_t>(n) * ldE);
    project_dev(*pr, dX, m, n, ld, a.flag, E.p, ldE, nullptr);                  // :350-363 for all k at once
    std::vector<HcTask> tasks(K);
    HcParams bp = base; bp.N_cluster = a.indN;
    for (int k = 0; k < K; ++k) {
        tasks[k].d_mat = E.p + static_cast<long long>(k) * p; tasks[k].ld = ldE; tasks[k].n = n; tasks[k].p = p; tasks[k].prm = bp;
        tasks[k].prm.dec_level = 0; tasks[k].prm.dec_k = k;
    }
    std::vector<HcResult> hr;
    get_opt_hclust_batch(tasks, false, hr);                                     // :366 getrowColor
    std::vector<int> enrp(static_cast<size_t>(n) * K);
    for (int k = 0; k < K; ++k) { out.rc |= hr[k].rc; for (int i = 0; i < n; ++i) enrp[static_cast<size_t>(k) * n + i] = colour_of(hr[k].f[i]); }
    { LastSmall &L = last_small(); L.n = n; L.K = K; L.p = p; L.ldE = ldE; L.enrp = enrp; }
    WmTask wt; wt.nC = enrp.data(); wt.N = n; wt.C = K; wt.prm = base; wt.prm.N_cluster = a.N_cluster;   // :401
    wt.prm.dec_level = 1;
    std::vector<WmTask> wts{wt};
    std::vector<WmResult> wr;
    wmetac_batch(wts, a.want_x0, false, wr);
    out.rc |= wr[0].rc;
    out.pred = wr[0].finalC;
    if (a.want_viE || true) { dws().viE_out.ensure(static_cast<size_t>(n) * p); out.viE.p = dws().viE_out.p; ensemble_mean_dev(E.p, ldE, n, p, K, out.viE.p); }   // :416
    if (a.N_cluster <= 0 && n > 10000) merge_small(out.pred);                   // :418-427
    out.n_pred = relabel_first(out.pred);                                       // :429-443
    if (a.want_x0) { out.x0 = wr[0].x0; out.x0_cols = wr[0].ncl; }
    stream_sync();
    last_small().valid = true;
}

// folds of partition.ncells cells, the last two balanced (R/SHARP.R:513-536)
static std::vector<int> fold_starts(int n, int ng) {
    const int T = (n + ng - 1) / ng;
    std::vector<int> st(T + 1, 0);
    if (T == 1) { st[1] = n; return st; }
    for (int t = 0; t < T - 2; ++t) st[t + 1] = (t + 1) * ng;
    const int nt = n - (T - 2) * ng;
    st[T - 1] = (T - 2) * ng + std::min(nt / 2, ng);
    st[T] = n;
    return st;
}

// ---------------------------------------------------------------------------------------------
// SHARP_large (R/SHARP.R:478-851)
// ---------------------------------------------------------------------------------------------
// The front of a block: shuffle, folds, projection (E rows straight into shuffled order), the K*T base-clustering tasks -- and, for
// a block prepared ahead of time, their row preparation and distance matrices (hc_prefetch_begin).
struct LargeFront {
    XRef dX; int m = 0, n = 0; long long ld = 0; int K = 0, p = 0, ng = 0, flag = 0, projector = 0; double rN_seed = 0; bool fpart = false;
    HcParams bp;
    bool shuffle = false;
    std::vector<int> reind, fst, pos;                                          // (pos: cell -> shuffled position; uploaded from here)
    int T = 0;
    std::shared_ptr<Projector> pr;
    long long ldE = 0;
    double *E = nullptr; int *dpos = nullptr;                                   // device: this block's projection and shuffle map
    std::vector<HcTask> tasks;
    std::shared_ptr<HcPrefetch> hc;                                             // distance matrices already enqueued
    bool matches(XRef X, int m_, int n_, long long ld_, int K_, int p_, int ng_, const SharpArgs &a, const HcParams &b) const {
        return dX.p == X.p && dX.f64 == X.f64 && m == m_ && n == n_ && ld == ld_ && K == K_ && p == p_ && ng == ng_ && flag == a.flag &&
               projector == a.projector && rN_seed == a.rN_seed && fpart == a.fpart && bp.hmethod == b.hmethod && bp.N_cluster == b.N_cluster &&
               bp.minN == b.minN && bp.maxN == b.maxN && bp.sil_thre == b.sil_thre && bp.height_Ntimes == b.height_Ntimes;
    }
};
namespace {
// A front prepared for the NEXT SHARP-large call.  It belongs to that call alone: `born` is the number of the call that made it and only
// call born + 1 may take it (a hinted call that never comes, or comes later, finds it dropped -- an address and a shape that happen to
// match prove nothing about the bytes behind them).  The side stream reads the hinted block until that next call, or
// sharp_synchronize(), returns: the caller keeps the buffer alive and unchanged until then.
struct PendingFront { std::unique_ptr<LargeFront> f; hipStream_t stream = nullptr; hipEvent_t main_done = nullptr; int parity = 0;
                      unsigned long long born = 0, calls = 0; };
PendingFront &pending_front() { return per_slot<PendingFront>(); }
}  // namespace

// Finishes whatever the side streams of this file still have in flight (sharp_synchronize, sharp_shutdown).
void drain_side_streams() {
    PendingFront &PF = pending_front();
    if (PF.stream) (void)hipStreamSynchronize(PF.stream);
    if (dws().mean_stream) (void)hipStreamSynchronize(dws().mean_stream);
}
// Forgets a prepared front (after its stream has drained): error exits, sharp_trim, sharp_shutdown, sharp_projector_destroy.
void drop_pending_front() {
    PendingFront &PF = pending_front();
    if (PF.stream) (void)hipStreamSynchronize(PF.stream);
    PF.f.reset();
}

// The shuffle of a block (:493-507): sample(n) after set.seed(50) -- 1 ms of host time for 50 000 cells, a function of n alone unless
// rN.seed = 0.5.  It is computed on a thread of its own beside whatever the device work of the front needs first (the projector build:
// 1.4 ms), and a SHARP_unlimited run asks for its first block's shuffle before it builds its projectors (shuffle_ahead).
struct Shuffle { std::vector<int> reind, pos; };
static Shuffle make_shuffle(int n, bool random_seed) {
    Shuffle S;
    RRng rng(random_seed ? static_cast<uint32_t>(std::random_device{}()) : 50u);   // :493-499
    S.reind = rng.permutation(n);
    S.pos.assign(n, 0);
    for (int i = 0; i < n; ++i) S.pos[S.reind[i] - 1] = i;
    return S;
}
namespace {
struct ShuffleAhead { int n = 0; std::future<Shuffle> fut; };
ShuffleAhead &shuffle_ahead_slot() { return per_slot<ShuffleAhead>(); }
}  // namespace
static void shuffle_ahead(long long n, double rN_seed) {
    ShuffleAhead &H = shuffle_ahead_slot();
    H.n = 0; H.fut = std::future<Shuffle>();
    if (n >= 100000 || n < 5000 || rN_seed == 0.5) return;
    H.n = static_cast<int>(n);
    H.fut = std::async(std::launch::async, make_shuffle, static_cast<int>(n), false);
}

static void large_front(LargeFront &F, const SharpArgs &a, bool ahead, double *E_into = nullptr, int *pos_into = nullptr,
                        const Shuffle *same_size = nullptr) {
    const int n = F.n, m = F.m, K = F.K, p = F.p;
    F.shuffle = n < 100000;                                                     // :504-507
    std::vector<int> &pos = F.pos;
    std::future<Shuffle> shuf;
    if (F.shuffle) {
        ShuffleAhead &H = shuffle_ahead_slot();
        if (same_size && a.rN_seed != 0.5 && static_cast<int>(same_size->pos.size()) == n) { F.reind = same_size->reind; pos = same_size->pos; }
        else if (H.n == n && H.fut.valid() && a.rN_seed != 0.5) { shuf = std::move(H.fut); H.n = 0; }
        else shuf = std::async(std::launch::async, make_shuffle, n, a.rN_seed == 0.5);
    } else pos.assign(n, 0);
    F.fst = fold_starts(n, F.ng);
    F.T = static_cast<int>(F.fst.size()) - 1;
    // projectors drawn for this call: the block's compaction (which needs X only) is enqueued behind the draw kernel, on the second stream,
    // and runs beside the packing of the row lists (small kernels and host round trips: the chip is nearly idle there; beside the draw
    // kernel itself, which holds every CU's registers, the compaction took 1.45 x as long)
    unsigned ahead_tok = 0u;
    const bool may_go_ahead = !a.projector && !ahead && knobs().rp_ahead > 0;
    const std::function<void()> start_compaction = [&] { if (may_go_ahead && !ahead_tok) ahead_tok = rp_compact_ahead(F.dX, m, n, F.ld, a.flag); };
    if (knobs().rp_ahead >= 2) start_compaction();
    { HostTimer ht("projector_build"); F.pr = projector_for(a, m, p, K, &start_compaction); }       // :539-549
    F.ldE = static_cast<long long>(F.pr->K) * p;
    if (shuf.valid()) { Shuffle S = shuf.get(); F.reind = std::move(S.reind); pos = std::move(S.pos); }
    if (E_into) {                                                               // (a block of a batch: rows of the batch's own buffers)
        F.E = E_into; F.dpos = F.shuffle ? pos_into : nullptr;
        if (F.shuffle) SHARP_HIP_CHECK(hipMemcpyAsync(pos_into, pos.data(), static_cast<size_t>(n) * sizeof(int), hipMemcpyHostToDevice, ctx().stream));
    } else {
        DevBuf<double> &E = ahead ? dws().Eb : dws().E;
        { HostTimer ht("alloc_E"); E.ensure(static_cast<size_t>(n) * F.ldE); }
        DevBuf<int> &dpos = ahead ? dws().posb : dws().pos;
        if (F.shuffle) { dpos.ensure(n); dpos.upload(pos.data(), n); }
        F.E = E.p; F.dpos = F.shuffle ? dpos.p : nullptr;
    }
    // E rows are written straight into shuffled order, so fold t is the contiguous row range [fst[t], fst[t+1])
    project_dev(*F.pr, F.dX, m, n, F.ld, a.flag, F.E, F.ldE, F.dpos, ahead_tok);  // :567-585 for every (k, t)
    if (a.fpart) {                                                              // newE1 = round(newE1, digits = 1)  (unlimited2 :410)
        Ctx &c = ctx();
        hipLaunchKernelGGL(round1_kernel, dim3(c.num_cu * 8), dim3(256), 0, c.stream, F.E, static_cast<long long>(n) * F.ldE);
        launch_check("round1_kernel");
    }
    // K*T base-clustering tasks in one batch (:554-618)
    F.tasks.assign(static_cast<size_t>(K) * F.T, HcTask());
    for (int k = 0; k < K; ++k)
        for (int t = 0; t < F.T; ++t) {
            HcTask &tk = F.tasks[static_cast<size_t>(k) * F.T + t];
            tk.d_mat = F.E + static_cast<long long>(F.fst[t]) * F.ldE + static_cast<long long>(k) * p;
            tk.ld = F.ldE; tk.n = F.fst[t + 1] - F.fst[t]; tk.p = p; tk.prm = F.bp;
            tk.prm.dec_level = 0; tk.prm.dec_k = k; tk.prm.dec_fold = t;           // (decision log; the block is stamped by the call that uses the front)
        }
}

// enE / K (R/SHARP.R:750,776) on the slot's mean stream; after: an event it has to wait for (nullptr: the main stream as it stands now)
static void enqueue_ensemble_mean(const double *E, long long ldE, int n, int p, int K, double *viE_sh, hipEvent_t after) {
    Ctx &c = ctx();
    hipEvent_t &mean_done = dws().mean_done;
    hipStream_t &ms = dws().mean_stream;                                        // (a stream of its own: the library's second stream may hold
    if (!mean_done) {                                                           // the next block's kernels, SHARP_unlimited)
        SHARP_HIP_CHECK(hipEventCreateWithFlags(&mean_done, hipEventDisableTiming));
        SHARP_HIP_CHECK(hipEventCreateWithFlags(&dws().mean_go, hipEventDisableTiming));
        int lo = 0, hi = 0;
        SHARP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SHARP_HIP_CHECK(hipStreamCreateWithPriority(&ms, hipStreamNonBlocking, hi));
    }
    SHARP_HIP_CHECK(hipEventRecord(dws().mean_go, c.stream));                  // E is complete (the RP stage ran on the main stream)
    SHARP_HIP_CHECK(hipStreamWaitEvent(ms, dws().mean_go, 0));
    if (after) SHARP_HIP_CHECK(hipStreamWaitEvent(ms, after, 0));
    {
        StreamScope scope(ms);
        ensemble_mean_dev(E, ldE, n, p, K, viE_sh);
    }
    SHARP_HIP_CHECK(hipEventRecord(mean_done, ms));
}

// Everything of SHARP_large behind the base clustering (R/SHARP.R:620-851): per-fold wMetaC, cross-fold sMetaC, un-shuffle, the
// small-cluster merge and the relabel.  hr: the K*T results of the block's base tasks, task (k, t) at k*T + t.
static void large_tail(const LargeFront &F, const SharpArgs &a, int K, int p, const HcParams &base, const HcResult *hr, SharpOut &out) {
    const int n = F.n, T = F.T;
    const bool shuffle = F.shuffle;
    const std::vector<int> &reind = F.reind, &fst = F.fst;
    const long long ldE = F.ldE;
    struct { double *p; } E{F.E};
    struct { int *p; } dpos{F.dpos};
    DevBuf<double> &viE_sh = dws().viE_sh;                                      // enE / K in shuffled order (:750,776)
    viE_sh.ensure(static_cast<size_t>(n) * p);
    // The ensemble mean streams all of E once (HBM-bound, 0.47 ms at cfg2) and is first needed by the final sMetaC: on the side stream
    // it runs beside the per-fold wMetaC kernels (LDS- and latency-bound) instead of in front of them -- unless it has gone out already,
    // behind the last chunk's agglomeration (enqueue_ensemble_mean from the base clustering's hook: beside the last statistics).
    hipEvent_t &mean_done = dws().mean_done;
    if (!hc_after_last_agglomeration_fired()) enqueue_ensemble_mean(E.p, ldE, n, p, K, viE_sh.p, nullptr);
    bool view_sent = false;
    if (a.view_to_host && !a.fpart) {
        // viE (:776-783) is final as soon as the mean is: back to the original cell order and off to the host on the mean stream, under the tail
        DriverWs &W = dws();
        const size_t cnt = static_cast<size_t>(n) * p;
        if (W.view_pinned_n < cnt) {
            if (W.view_pinned) (void)hipHostFree(W.view_pinned);
            W.view_pinned = nullptr; W.view_pinned_n = 0;
            SHARP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&W.view_pinned), cnt * sizeof(double), hipHostMallocDefault));
            W.view_pinned_n = cnt;
        }
        if (!W.view_done) SHARP_HIP_CHECK(hipEventCreateWithFlags(&W.view_done, hipEventDisableTiming));
        W.viE_out.ensure(cnt);
        {
            StreamScope scope(W.mean_stream);
            if (shuffle) {
                hipLaunchKernelGGL(gather_rows_kernel, dim3(ctx().num_cu * 8), dim3(256), 0, W.mean_stream, viE_sh.p, dpos.p, static_cast<long long>(n), p, W.viE_out.p);
                launch_check("gather_rows_kernel");
            } else {
                SHARP_HIP_CHECK(hipMemcpyAsync(W.viE_out.p, viE_sh.p, cnt * 8, hipMemcpyDeviceToDevice, W.mean_stream));
            }
            SHARP_HIP_CHECK(hipMemcpyAsync(W.view_pinned, W.viE_out.p, cnt * 8, hipMemcpyDeviceToHost, W.mean_stream));
        }
        SHARP_HIP_CHECK(hipEventRecord(W.view_done, W.mean_stream));
        out.viE.p = W.viE_out.p; out.viE.staged = W.view_pinned; out.viE.staged_done = W.view_done;
        view_sent = true;
    }
    // enrp per fold (:627-635); labels "<colour>p<t>" only need to be distinct per (k, t): the colour id does
    std::vector<std::vector<int>> enrp(T);
    for (int q = 0; q < K * T; ++q) out.rc |= hr[q].rc;
    {
    HostTimer ht("tail_enrp");
    host_parallel_for(T, 16, [&](int t) {
        const int nt = fst[t + 1] - fst[t];
        enrp[t].resize(static_cast<size_t>(nt) * K);
        for (int k = 0; k < K; ++k) {
            const HcResult &r = hr[static_cast<size_t>(k) * T + t];
            for (int i = 0; i < nt; ++i) enrp[t][static_cast<size_t>(k) * nt + i] = colour_of(r.f[i]);
        }
    });
    }
    // per-fold wMetaC (:692-709)
    std::vector<WmTask> wts(T);
    for (int t = 0; t < T; ++t) {
        wts[t].nC = enrp[t].data(); wts[t].N = fst[t + 1] - fst[t]; wts[t].C = K;
        wts[t].prm = base; wts[t].prm.N_cluster = a.enpN;
        wts[t].prm.dec_level = 1; wts[t].prm.dec_fold = t;
    }
    std::vector<WmResult> wr;
    { HostTimer ht("wmetac_total"); wmetac_batch(wts, a.want_x0, false, wr); }
    step_mark("  tail: per-fold wMetaC done, folds", T);
    SHARP_HIP_CHECK(hipStreamWaitEvent(ctx().stream, mean_done, 0));           // every use of viE_sh is below
    std::vector<int> Slab(n);                                                   // SrowColor in shuffled order
    std::vector<int> stf;                                                       // meta id per (fold, cluster) column of sx0
    std::vector<int> uid(n);
    int nCu = 0;
    std::vector<int> col0(T + 1, 0);                                            // first sx0 column of fold t
    for (int t = 0; t < T; ++t) {
        out.rc |= wr[t].rc;
        // fColor "<id>en<t>": unique() order = fold by fold, first appearance inside the fold
        std::vector<int> u;
        const int nu = first_appearance_ids(wr[t].finalC.data(), wr[t].finalC.size(), u);
        for (size_t i = 0; i < u.size(); ++i) uid[fst[t] + i] = nCu + u[i];
        col0[t] = nCu;
        nCu += nu;
    }
    col0[T] = nCu;
    if (a.fpart) {
        // fColor[reind] = fColor (unlimited2 :522-526): labels back in the original cell order; the caller's sMetaC takes
        // unique() of the concatenated blocks, i.e. first appearance in THAT order
        std::vector<int> lab(n);
        for (int i = 0; i < n; ++i) lab[shuffle ? reind[i] - 1 : i] = uid[i] + 1;
        out.pred = lab;
        out.n_pred = relabel_first(out.pred);
        dws().viE_out.ensure(static_cast<size_t>(n) * p); out.viE.p = dws().viE_out.p;
        if (shuffle) {
            Ctx &c = ctx();
            hipLaunchKernelGGL(gather_rows_kernel, dim3(c.num_cu * 8), dim3(256), 0, c.stream, viE_sh.p, dpos.p, static_cast<long long>(n), p, out.viE.p);
            launch_check("gather_rows_kernel");
        } else {
            SHARP_HIP_CHECK(hipMemcpyAsync(out.viE.p, viE_sh.p, static_cast<size_t>(n) * p * 8, hipMemcpyDeviceToDevice, ctx().stream));
        }
        stream_sync();
        return;
    }
    if (T == 1) {
        // :738-746 then :828: as.numeric("<id>en1") is NA for every cell -> a single cluster (reference quirk 2)
        std::fill(Slab.begin(), Slab.end(), 1);
        stf.assign(nCu, 1);
    } else {
        DevBuf<double> &means = dws().fold_means;                               // (kept: hipMalloc / hipFree synchronise the whole device, and in a
        means.ensure(static_cast<size_t>(nCu) * p);                             // batched SHARP_unlimited later chunks' agglomeration is in flight)
        { HostTimer ht("tail_fold_means"); cluster_means_dev(viE_sh.p, p, n, p, uid, nCu, means.p); }   // sMetaC :58-63 on E1 = enE/K
        HcParams sp = base; sp.N_cluster = a.N_cluster;
        sp.dec_level = 2;
        SmResult sr;
        step_mark("  tail: fold means done, clusters", nCu);
        { HostTimer ht("smetac_total"); sr = smetac_from_means(means.p, nCu, p, n, sp); }   // :754
        step_mark("  tail: sMetaC done", sr.optN);
        out.rc |= sr.rc;
        stf = sr.tf;
        for (int i = 0; i < n; ++i) Slab[i] = stf[uid[i]];
    }
    HostTimer ht_fin("tail_finish");
    out.pred.resize(n);
    for (int i = 0; i < n; ++i) out.pred[shuffle ? reind[i] - 1 : i] = Slab[i];   // :775-783
    if (a.want_x0) {
        // sx0: block-diagonal per-fold soft matrices (:717-731), columns merged by stf (:761-772), un-shuffled
        if (T == 1) for (int q = 0; q < nCu; ++q) stf[q] = q + 1;               // x0 = sx0 (:746): one fold, no columns merged
        const int sn = *std::max_element(stf.begin(), stf.end());
        out.x0.assign(static_cast<size_t>(n) * sn, 0.0);
        out.x0_cols = sn;
        for (int t = 0; t < T; ++t) {
            const int nt = fst[t + 1] - fst[t], ncl = wr[t].ncl;
            for (int q = 0; q < ncl; ++q) {
                const int dstc = stf[col0[t] + q] - 1;
                for (int i = 0; i < nt; ++i) {
                    const int cell = shuffle ? reind[fst[t] + i] - 1 : fst[t] + i;
                    out.x0[static_cast<size_t>(dstc) * n + cell] += wr[t].x0[static_cast<size_t>(q) * nt + i];
                }
            }
        }
    }
    // viE back to the original cell order (:776-783) -- unless that went out with the early download above
    if (!view_sent) {
        dws().viE_out.ensure(static_cast<size_t>(n) * p); out.viE.p = dws().viE_out.p;
        if (shuffle) {
            Ctx &c = ctx();
            hipLaunchKernelGGL(gather_rows_kernel, dim3(c.num_cu * 8), dim3(256), 0, c.stream, viE_sh.p, dpos.p, static_cast<long long>(n), p, out.viE.p);
            launch_check("gather_rows_kernel");
        } else {
            SHARP_HIP_CHECK(hipMemcpyAsync(out.viE.p, viE_sh.p, static_cast<size_t>(n) * p * 8, hipMemcpyDeviceToDevice, ctx().stream));
        }
    }
    if (a.N_cluster <= 0 && n > 10000) merge_small(out.pred);                   // :816-825
    out.n_pred = relabel_first(out.pred);                                       // :828-843
    stream_sync();
}

static void sharp_large_dev_body(XRef dX, int m, int n, long long ld, const SharpArgs &a, int K, int p, int ng, HcParams base, SharpOut &out);
void sharp_large_dev(XRef dX, int m, int n, long long ld, const SharpArgs &a, int K, int p, int ng, HcParams base, SharpOut &out) {
    try {
        sharp_large_dev_body(dX, m, n, ld, a, K, p, ng, base, out);
    } catch (...) {                                                             // nothing prepared ahead outlives a failed call
        drop_pending_front();
        rp_compact_ahead_drop();
        if (ctx_unchecked().stream2) (void)hipStreamSynchronize(ctx_unchecked().stream2);   // (the block's compaction may still be reading the caller's X)
        // the main stream too: the front's per-call projector goes back to the block pool during unwinding WITHOUT a synchronisation of
        // its own, and another slot on this device could take those blocks while this slot's kernels still read them
        if (ctx_unchecked().stream) (void)hipStreamSynchronize(ctx_unchecked().stream);
        throw;
    }
}
static void sharp_large_dev_body(XRef dX, int m, int n, long long ld, const SharpArgs &a, int K, int p, int ng, HcParams base, SharpOut &out) {
    HostTimer ht_all("sharp_large_total");
    last_small().valid = false;                                                 // E is about to be overwritten
    HcParams bp = base; bp.N_cluster = a.indN;
    if (a.fpart) bp.maxN = 40;                                                  // "for partition clustering" (unlimited2 :421)
    // the front: prepared ahead of time by the previous block's call, or now
    std::unique_ptr<LargeFront> Fp;
    PendingFront &PF = pending_front();
    const unsigned long long call_no = ++PF.calls;
    if (PF.f) {
        if (PF.born + 1 == call_no && PF.f->matches(dX, m, n, ld, K, p, ng, a, bp)) {
            Fp = std::move(PF.f);
            std::swap(dws().E, dws().Eb);                                       // its buffers become the current block's
            std::swap(dws().pos, dws().posb);
        } else {
            SHARP_HIP_CHECK(hipStreamSynchronize(PF.stream));                   // (a prepared block that is not the one asked for: dropped)
            PF.f.reset();
        }
    }
    if (!Fp) {
        Fp.reset(new LargeFront);
        Fp->dX = dX; Fp->m = m; Fp->n = n; Fp->ld = ld; Fp->K = K; Fp->p = p; Fp->ng = ng; Fp->flag = a.flag; Fp->projector = a.projector;
        Fp->rN_seed = a.rN_seed; Fp->fpart = a.fpart; Fp->bp = bp;
        large_front(*Fp, a, false);
    }
    LargeFront &F = *Fp;
    // The NEXT block's front on a side stream.  If this block's own front was prepared ahead, its agglomeration is enqueued first and
    // the next front waits for it: it then runs under this block's statistics and host-bound tail instead of beside the HBM-bound
    // agglomeration (which it only slowed down: 259 -> 246 ms for the ten blocks of cfg3 ungated).
    hipEvent_t after_agglo = nullptr;
    if (F.hc && a.next_dX.p) { HostTimer ht("base_clustering_total"); after_agglo = hc_prefetch_agglomerate(*F.hc); }
    if (a.next_dX.p && a.projector && knobs().block_prefetch) {
        const int nn = static_cast<int>(a.next_n);
        const int Tn = (nn + ng - 1) / ng;
        if (a.next_n >= 5000 && a.next_n < (1LL << 31) && static_cast<long long>(K) * Tn <= ctx().num_cu && Tn > 1) {
            std::unique_ptr<LargeFront> N(new LargeFront);
            N->dX = a.next_dX; N->m = m; N->n = nn; N->ld = a.next_ld; N->K = K; N->p = p; N->ng = ng; N->flag = a.flag;
            N->projector = a.projector; N->rN_seed = a.rN_seed; N->fpart = a.fpart; N->bp = bp;
            if (a.maxN <= 0 && !a.fpart) N->bp.maxN = std::max(40, (nn + 4999) / 5000);   // the next block's own default (R/SHARP.R:144-146)
            // The prefetch is an optimisation: whatever fails in it (an allocation on a GPU shared with another process, say) must not fail
            // THIS block, which needs none of it.
            try {
            if (!PF.stream) {
                // lowest priority: the tail's small kernels (on the critical path) go first whenever they are ready; at equal priority the
                // next block's distance GEMM kept every CU busy and they waited (the tail took 12 ms instead of 6.6)
                int lo = 0, hi = 0;
                SHARP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                SHARP_HIP_CHECK(hipStreamCreateWithPriority(&PF.stream, hipStreamNonBlocking, lo));
            }
            {
                // (the main stream has nothing pending that the next block depends on; its buffers are its own)
                // This block's own front ran on the main stream (first block of a call): the side stream must not start before it has
                // finished -- the RP stage's chunk buffers (rp2.hip) are shared, and the next block's compaction would overwrite
                // entries this block's apply kernels are still reading.  (A front prepared ahead ran on the side stream itself.)
                if (!F.hc) {
                    if (!PF.main_done) SHARP_HIP_CHECK(hipEventCreateWithFlags(&PF.main_done, hipEventDisableTiming));
                    SHARP_HIP_CHECK(hipEventRecord(PF.main_done, ctx().stream));
                    SHARP_HIP_CHECK(hipStreamWaitEvent(PF.stream, PF.main_done, 0));
                }
                StreamScope scope(PF.stream);
                struct Polite { Polite() { ctx().polite = true; } ~Polite() { ctx_unchecked().polite = false; } } polite;
                if (after_agglo) SHARP_HIP_CHECK(hipStreamWaitEvent(PF.stream, after_agglo, 0));
                large_front(*N, a, true);
                if (hc_prefetch_possible(N->tasks)) N->hc = hc_prefetch_begin(N->tasks, PF.parity ^= 1);
            }
            if (N->hc) { PF.f = std::move(N); PF.born = call_no; }
            else SHARP_HIP_CHECK(hipStreamSynchronize(PF.stream));              // (cannot be kept: finish it, the projection is simply redone)
            } catch (...) {
                if (PF.stream) (void)hipStreamSynchronize(PF.stream);           // what was enqueued for the next block drains; the block is
                (void)hipGetLastError();                                        // simply prepared by its own call
                PF.f.reset();
            }
        }
    }
    std::vector<HcResult> hr;
    {
        HostTimer ht("base_clustering_total");
        if (F.hc) { hc_set_after_last_agglomeration(nullptr); hc_prefetch_stamp_block(*F.hc, a.block); hc_prefetch_finish(*F.hc, false, hr); }
        else {
            // the ensemble mean needs E only: it goes out behind the LAST chunk's agglomeration, beside that chunk's statistics
            // (latency-bound), instead of behind them in the serial tail (-0.5 ms per cfg2 step)
            dws().viE_sh.ensure(static_cast<size_t>(n) * p);
            const double *Ep = F.E;
            const long long ldEp = F.ldE;
            double *vsh = dws().viE_sh.p;
            if (knobs().mean_early) hc_set_after_last_agglomeration([=](hipEvent_t ev) { enqueue_ensemble_mean(Ep, ldEp, n, p, K, vsh, ev); });
            else hc_set_after_last_agglomeration(nullptr);
            for (HcTask &tk : F.tasks) tk.prm.dec_block = a.block;
            try { get_opt_hclust_batch(F.tasks, false, hr); } catch (...) { hc_set_after_last_agglomeration(nullptr); throw; }
        }
    }
    large_tail(F, a, K, p, base, hr.data(), out);
    hc_set_after_last_agglomeration(nullptr);
}

// ---------------------------------------------------------------------------------------------
// SHARP front door (R/SHARP.R:44-318) for a prepared matrix resident on the device
// ---------------------------------------------------------------------------------------------
void sharp_front_dev(XRef dX, int m, long long n_, long long ld, SharpArgs a, SharpOut &out) {
    SHARP_REQUIRE(dX.p, "No expression data is provided!");
    SHARP_REQUIRE(n_ >= 3 && n_ < (1LL << 31) && m >= 2, "SHARP: need at least 3 cells and 2 genes");
    const int n = static_cast<int>(n_);
    if (a.rN_seed != 0.5) SHARP_REQUIRE(std::fmod(a.rN_seed, 1.0) == 0.0, "The rN.seed should be an integer!");   // :169-179
    int p = a.reduced_ndim > 0 ? a.reduced_ndim : static_cast<int>(std::ceil(std::log2(static_cast<double>(n)) / (0.2 * 0.2)));   // :119-122
    int base_ncells = a.base_ncells > 0 ? a.base_ncells : 5000;                 // :124-128
    int part = a.partition_ncells > 0 ? a.partition_ncells : 2000;              // :130-132
    HcParams base;
    base.hmethod = a.hmethod > 0 ? a.hmethod : 1;                               // :134-136
    base.minN = a.minN > 0 ? a.minN : 2;                                        // :139-141
    base.maxN = a.maxN > 0 ? a.maxN : std::max(40, (n + 4999) / 5000);          // :144-146
    base.sil_thre = a.sil_thre >= 0 ? a.sil_thre : 0.35;                        // :149-151
    base.height_Ntimes = a.height_Ntimes > 0 ? a.height_Ntimes : 2.0;           // :154-156
    base.dec_block = a.block;
    int K = a.K;
    if (a.N_cluster > 0 && n < base_ncells) {                                   // :181-191
        a.indN = a.N_cluster;
        base_ncells = (n + 1) / 2;
        part = (n + 1) / 2;
        if (K <= 0) K = 15;
    }
    if (a.fpart) {                                                              // SHARP_fpart has no small path
        if (K <= 0) K = 5;
        SHARP_REQUIRE(part >= 3 && part <= kHcMaxN, "partition.ncells must be between 3 and 16384");
        out.path = 1;
        sharp_large_dev(dX, m, n, ld, a, K, p, part, base, out);
    } else if (n < base_ncells) {
        if (K <= 0) K = 15;                                                     // :254-257
        SHARP_REQUIRE(n <= kHcMaxN, "SHARP_small: more than 16384 cells in one unpartitioned clustering task");
        out.path = 0;
        sharp_small_dev(dX, m, n, ld, a, K, p, base, out);
    } else {
        if (K <= 0) K = 5;                                                      // :268-271
        SHARP_REQUIRE(part >= 3 && part <= kHcMaxN, "partition.ncells must be between 3 and 16384");
        out.path = 1;
        sharp_large_dev(dX, m, n, ld, a, K, p, part, base, out);
    }
    out.p = p; out.K = K;
}

// ---------------------------------------------------------------------------------------------
// SHARP_unlimited pieces (R/SHARP_unlimited.R:29-242)
// ---------------------------------------------------------------------------------------------
// one block: y[[i]] = SHARP(mat, reduced.ndim = p, prep = FALSE, logflag = FALSE, rM = rM, ensize.K, rN.seed)
// (:135) and the colMeans of its viE per predicted cluster -- all sMetaC ever uses of E1 (:163, R/sMetaC.R:58-63)
// The view reduction of ONE SHARP_unlimited call (R/SHARP_unlimited.R:216-228), fixed when the call begins and handed down by value to
// everything that works for it (block loops, tail helpers, device workers): kdim > 0 = the caller's viE takes kdim columns per cell, the
// product E1 %*% ranM2(p, kdim, seed) taken per block on the device; seed = the ONE seed of that call's z0 (an integer, also for an
// unseeded run: the reference draws z0 once and multiplies all of E1 by it, :219-225).
struct ViewCall {
    int kdim = 0;
    double seed = 0;
    int cols(int p) const { return kdim > 0 ? kdim : p; }
};
void unlimited_block_summary(const SharpOut &o, long long nb, int p, std::vector<int> &pred, std::vector<double> &means,
                             std::vector<long long> &counts, double *viE_host, const ViewCall &view);
void unlimited_block_dev(XRef dX, int m, long long nb, long long ld, int p, int projector, int K, double rN_seed,
                         std::vector<int> &pred, std::vector<double> &means, std::vector<long long> &counts, double *viE_host,
                         const ViewCall &view, int flag = 1, const SharpArgs *fpart_args = nullptr, XRef next_dX = XRef(), long long next_n = 0,
                         long long next_ld = 0, int block = 0) {
    SharpArgs a;
    if (fpart_args) a = *fpart_args;
    a.block = block;                                            // SHARP_unlimited2: every SHARP_fpart parameter
    a.next_dX = next_dX; a.next_n = next_n; a.next_ld = next_ld;                // the block after this one (prepared under this one's tail)
    a.K = K; a.reduced_ndim = p; a.flag = flag; a.projector = projector; a.rN_seed = rN_seed; a.want_viE = true;
    a.fpart = fpart_args != nullptr;
    SharpOut o;
    sharp_front_dev(dX, m, nb, ld, a, o);
    unlimited_block_summary(o, nb, p, pred, means, counts, viE_host, view);
}

// viewflag above 1e5 cells (R/SHARP_unlimited.R:216-228): enresults$viE = 1/sqrt(kdim) * E1 %*% ranM2(p, kdim, seed), kdim = 50.  E1 (ncells x p) exists
// only to be multiplied: the product is taken PER BLOCK on the device, by the RP kernels themselves (the block's viE as an fp64 "expression"
// block of p genes, raw mode), as soon as the block's tail has its viE, and ncells x kdim doubles leave the GPU instead of ncells x p
// (cfg3: 0.2 GB instead of 1.9 GB).  The entries that carry a view_dim argument say so themselves; for the others sharp_unlimited_view_dim()
// arms it for the NEXT SHARP_unlimited call OF THE CALLING THREAD (the arm is thread-local: a call on another thread never takes it, and
// nothing of a running call is shared).  take_view() at the top of a run turns the arm into that run's ViewCall -- whatever the run then
// does, the arm is spent.
thread_local int tl_view_pending = 0;
static double fresh_view_seed() {                        // an unseeded run: one integer seed for the call's z0 (set.seed() of a random word)
    std::random_device rd;
    return static_cast<double>(rd() & 0x7fffffffu);
}
static ViewCall make_view(int kdim, double rN_seed, int K) {
    ViewCall v;
    v.kdim = kdim > 0 ? kdim : 0;
    // `50 + rN.seed + k` at :222 reads a `k` that only exists inside the foreach at :96 (an R error whenever a seed
    // is given); fixed as the next seed of that sequence, k = ensize.K + 1
    if (v.kdim > 0) v.seed = rN_seed == 0.5 ? fresh_view_seed() : 50 + rN_seed + K + 1;
    return v;
}
static ViewCall take_view(double rN_seed, int K) {
    const int kdim = tl_view_pending;
    tl_view_pending = 0;
    return make_view(kdim, rN_seed, K);
}
struct ViewProj { std::shared_ptr<Projector> pr; int p = 0, kdim = 0; double seed = 0; DevBuf<double> out, pad; };

// labels, per-cluster means of viE and cluster sizes of one finished block (what the cross-block sMetaC needs, :153-163)
void unlimited_block_summary(const SharpOut &o, long long nb, int p, std::vector<int> &pred, std::vector<double> &means,
                             std::vector<long long> &counts, double *viE_host, const ViewCall &view) {
    pred = o.pred;
    const int G = o.n_pred;
    std::vector<int> uid(pred.size());
    counts.assign(G, 0);
    for (size_t i = 0; i < pred.size(); ++i) { uid[i] = pred[i] - 1; ++counts[pred[i] - 1]; }   // ids are first-appearance ordered
    DevBuf<double> &dm = dws().block_means;
    dm.ensure(static_cast<size_t>(G) * p);
    cluster_means_dev(o.viE.p, p, static_cast<int>(nb), p, uid, G, dm.p);
    means.resize(static_cast<size_t>(G) * p);
    dm.download(means.data(), means.size());
    if (!viE_host) return;
    const int kdim = view.kdim;
    if (kdim <= 0) { o.viE.download(viE_host, static_cast<size_t>(nb) * p); return; }   // E1 rows of this block (:153), viewflag only
    ViewProj &V = per_slot<ViewProj>();                                         // (a tail helper's slot builds its own, once)
    const double seed = view.seed;                                              // (one z0 per call: every block, slot and device builds the same one)
    if (!V.pr || V.p != p || V.kdim != kdim || V.seed != seed) {
        V.pr = build_projector(p, kdim, 1, &seed);                              // ranM2(p, kdim, seed): the same draw as ranM (R/ranM2.R:11-35)
        V.p = p; V.kdim = kdim; V.seed = seed;
    }
    V.out.ensure(static_cast<size_t>(nb) * kdim);
    HostTimer ht("tail_view_reduce");
    // (the block's viE is the library's own: finite, 16-byte aligned, |x| far below FLT_MAX -- no validation pass as for a caller's fp64 block;
    // an odd p gets a padded copy: the RP kernels read an fp64 block with 16-byte loads, i.e. an even leading dimension)
    const double *src = o.viE.p;
    long long ldv = p;
    if (p % 2) {
        ldv = p + 1;
        V.pad.ensure(static_cast<size_t>(nb) * ldv);
        SHARP_HIP_CHECK(hipMemcpy2DAsync(V.pad.p, static_cast<size_t>(ldv) * 8, o.viE.p, static_cast<size_t>(p) * 8, static_cast<size_t>(p) * 8, static_cast<size_t>(nb),
                                         hipMemcpyDeviceToDevice, ctx().stream));
        src = V.pad.p;
    }
    project_dev(*V.pr, XRef(src), p, static_cast<int>(nb), ldv, 0, V.out.p, kdim, nullptr);
    V.out.download(viE_host, static_cast<size_t>(nb) * kdim);
}

// Blocks [b0, b1) of a SHARP_unlimited call whose base-clustering tasks run as ONE pipelined batch (get_opt_hclust_batch: chunks of
// at most one task per CU, chunk j + 1's distance GEMM beside chunk j's agglomeration, chunk j's statistics beside chunk j + 1's
// agglomeration) instead of block after block: the tasks of all blocks are independent, and a block's own tail -- per-fold wMetaC,
// sMetaC, relabels: small kernels and host loops -- runs from the batch's progress callback while later chunks' agglomeration keeps
// the HBM busy.  Every block is the same SHARP() call as in the block-by-block form (same shuffle, folds, parameters): same labels.
// deliver(b, out): called once per block; with SHARP_TAIL_THREADS > 0 (the default) from helper threads, in no particular order.
static void unlimited_batch_window(const XRef *dX, const long long *ncb, const long long *ldb, int b0, int b1, int m, int p, int proj, int K,
                                   double rN_seed, const std::function<void(int, const SharpOut &)> &deliver, const int *gids = nullptr) {
    FreeLater park_frees;                                 // (a buffer that grows in a tail helper or between chunks: no hipFree, i.e. no drain of the device, before the window is through)
    PendingFront &PF = pending_front();
    hc_set_after_last_agglomeration(nullptr);             // (every block's tail enqueues its own ensemble mean here)
    if (PF.f) { SHARP_HIP_CHECK(hipStreamSynchronize(PF.stream)); PF.f.reset(); }   // (no block-by-block front may be pending)
    const int nbk = b1 - b0;
    std::vector<std::unique_ptr<LargeFront>> F(nbk);
    std::vector<SharpArgs> A(nbk);
    std::vector<HcParams> base(nbk);
    long long rows = 0;
    for (int q = 0; q < nbk; ++q) rows += ncb[b0 + q];
    const long long ldE = static_cast<long long>(get_projector(proj)->K) * p;
    DriverWs &W = dws();
    W.Ebatch.ensure(static_cast<size_t>(rows) * ldE);
    W.posbatch.ensure(static_cast<size_t>(rows));
    last_small().valid = false;
    std::vector<HcTask> tasks;
    std::vector<size_t> first(nbk + 1, 0);
    std::vector<long long> row0(nbk + 1, 0);
#ifdef SHARP_LAB
    const int fo = knobs().front_overlap;                 // (SHARP_FRONT_OVERLAP, an experiment: lab builds only)
#else
    constexpr int fo = 0;
#endif
    for (int q = 0; q < nbk; ++q) {
        const int n = static_cast<int>(ncb[b0 + q]);
        SharpArgs &a = A[q];
        a.K = K; a.reduced_ndim = p; a.flag = 1; a.projector = proj; a.rN_seed = rN_seed; a.want_viE = true;   // as unlimited_block_dev
        HcParams &bs = base[q];                                                 // as sharp_front_dev for the large path
        bs.hmethod = 1; bs.minN = 2; bs.maxN = std::max(40, (n + 4999) / 5000); bs.sil_thre = 0.35; bs.height_Ntimes = 2.0;
        a.block = gids ? gids[b0 + q] : b0 + q; bs.dec_block = a.block;
        HcParams bp = bs; bp.N_cluster = a.indN;
        F[q].reset(new LargeFront);
        LargeFront &f = *F[q];
        f.dX = dX[b0 + q]; f.m = m; f.n = n; f.ld = ldb[b0 + q]; f.K = K; f.p = p; f.ng = 2000; f.flag = a.flag; f.projector = proj;
        f.rN_seed = rN_seed; f.fpart = false; f.bp = bp;
        row0[q + 1] = row0[q] + n;
    }
    // A block's front: its shuffle, its projection into its rows of the batch's E (the RP kernel), its K x T task descriptors.
    int fronts_done = 0;
    Shuffle same_size;                                    // (the shuffle of the latest block: blocks of one size share it)
    const auto front_of = [&](int q) {
        if (fo >= 2 && q >= 2) ctx().rp_wgs_cap = 1;
        large_front(*F[q], A[q], false, W.Ebatch.p + row0[q] * ldE, W.posbatch.p + row0[q], &same_size);
        ctx().rp_wgs_cap = 0;
        if (F[q]->shuffle && static_cast<int>(same_size.pos.size()) != F[q]->n) { same_size.reind = F[q]->reind; same_size.pos = F[q]->pos; }
    };
    std::function<hipEvent_t(size_t)> prepare;
    if (fo <= 0) {
        for (int q = 0; q < nbk; ++q) {
            front_of(q);
            tasks.insert(tasks.end(), F[q]->tasks.begin(), F[q]->tasks.end());
            first[q + 1] = tasks.size();
            for (size_t i = first[q]; i < first[q + 1]; ++i) tasks[i].prm.dec_block = A[q].block;
        }
        fronts_done = nbk;
    } else {
        // SHARP_FRONT_OVERLAP (an experiment): the fronts go out just in time -- a chunk of base tasks asks for the blocks it reads before its
        // own work is enqueued, so the first distance GEMM starts behind the first blocks' RP kernels and the later blocks are projected beside the
        // pipeline.  The task descriptors need no device work: fold boundaries and row offsets only (the same as large_front builds).
        for (int q = 0; q < nbk; ++q) {
            const LargeFront &f = *F[q];
            const std::vector<int> fst = fold_starts(f.n, f.ng);
            const int T = static_cast<int>(fst.size()) - 1;
            const double *E0 = W.Ebatch.p + row0[q] * ldE;
            for (int k = 0; k < K; ++k)
                for (int t = 0; t < T; ++t) {
                    HcTask tk;
                    tk.d_mat = E0 + static_cast<long long>(fst[t]) * ldE + static_cast<long long>(k) * p;
                    tk.ld = ldE; tk.n = fst[t + 1] - fst[t]; tk.p = p; tk.prm = f.bp;
                    tk.prm.dec_level = 0; tk.prm.dec_k = k; tk.prm.dec_fold = t; tk.prm.dec_block = A[q].block;
                    tasks.push_back(tk);
                }
            first[q + 1] = tasks.size();
        }
        prepare = [&](size_t upto) -> hipEvent_t {
            while (fronts_done < nbk && first[fronts_done] < upto) {
                const int q = fronts_done++;
                front_of(q);
                SHARP_REQUIRE(F[q]->tasks.size() == first[q + 1] - first[q] && F[q]->tasks[0].d_mat == tasks[first[q]].d_mat, "SHARP_unlimited: a block's tasks are not where its front put them");
                while (W.front_ev.size() <= static_cast<size_t>(q)) { hipEvent_t e; SHARP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); W.front_ev.push_back(e); }
                SHARP_HIP_CHECK(hipEventRecord(W.front_ev[q], ctx().stream));
            }
            return W.front_ev[fronts_done - 1];
        };
    }
    std::vector<HcResult> hr;
    const auto tail_of = [&](int q) {
        SharpOut o;
        o.path = 1;
        step_mark("tail begins, block", b0 + q);
        large_tail(*F[q], A[q], K, p, base[q], hr.data() + first[q], o);
        o.p = p; o.K = K;
        step_mark("tail's labels ready, block", b0 + q);
        deliver(b0 + q, o);
        step_mark("tail delivered, block", b0 + q);
        F[q].reset();
    };
    int next = 0;
    if (knobs().tail_threads <= 0 || nbk < 2) {
        const std::function<void(size_t)> progress = [&](size_t done) {
            while (next < nbk && first[next + 1] <= done) tail_of(next++);
        };
        {
            HostTimer ht("base_clustering_total");
            get_opt_hclust_batch(tasks, false, hr, &progress, fo > 0 ? &prepare : nullptr);
        }
        progress(tasks.size());
        return;
    }
    // The blocks' tails on host threads of their own, each bound to its own slot (own streams, own workspaces) on the same GPU.  A tail is a
    // chain of small kernels with host decisions in between: 3.8 ms of wall time on an idle GPU, 6-10 ms beside a saturating
    // agglomeration.  Run from the progress callback they kept THIS thread from enqueueing the next chunk's work, and the blocks whose
    // last tasks sit in the final chunks -- four of cfg3's ten -- ran their tails one after the other behind the last agglomeration
    // (27 ms of a 186 ms call, profiles/r04_cfg3_timeline.txt); several helpers run those side by side.  A tail reads only what is final
    // when its block is reported done (its rows of E, its entries of hr) and writes only its own block's outputs; deliver() is called
    // from the helpers, in no particular order.
    struct TailQueue {
        std::mutex mu;
        std::condition_variable cv;
        int ready = 0, taken = 0;    // blocks [0, ready) may run their tails; [0, taken) have been handed to a helper
        bool stop = false;
        std::exception_ptr err;
    } Q;
    Ctx &mc = ctx();
    const int dev = mc.device, owner = cur_slot();
    const bool prof = mc.profiling;
    const int H = std::max(1, std::min({knobs().tail_threads, nbk, 8, std::max(1, host_cores() / 4)}));   // (a helper brings a five-thread host pool)
    std::vector<int> tslot(H);
    for (int h = 0; h < H; ++h) tslot[h] = acquire_slot(dev, owner * 8 + h, 2);
    std::vector<std::thread> helpers;
    helpers.reserve(H);
    auto helper_body = [&](int h) {
            try {
                init_slot(tslot[h], dev, knobs().tail_priority);
                // (several helpers per GPU run their tails' host loops side by side: 25 folds on five threads take five rounds, on thirteen two)
                host_pool_threads_hint(std::max(5, std::min(12, host_cores() / (2 * (H + 1)))));
                ctx().profiling = prof;
                for (;;) {
                    int q;
                    {
                        std::unique_lock<std::mutex> lk(Q.mu);
                        Q.cv.wait(lk, [&] { return Q.ready > Q.taken || Q.stop; });
                        if (Q.ready <= Q.taken) break;
                        q = Q.taken++;
                    }
                    tail_of(q);
                }
                SHARP_HIP_CHECK(hipStreamSynchronize(ctx().stream));
                ctx().resolve_pending();
            } catch (...) {
                // The caller rethrows after the join and unwinds F[], the batch's E and pooled buffers; this slot's kernels may still be
                // reading them, and pool_give() hands blocks to other slots of the GPU without a synchronisation: drain first.
                Ctx &hc = ctx_unchecked();
                if (hc.stream) (void)hipStreamSynchronize(hc.stream);
                if (hc.stream2) (void)hipStreamSynchronize(hc.stream2);
                try { drain_side_streams(); } catch (...) {}
                std::lock_guard<std::mutex> lk(Q.mu);
                if (!Q.err) Q.err = std::current_exception();
            }
    };
    try {
        for (int h = 0; h < H; ++h) helpers.emplace_back(helper_body, h);
    } catch (...) {                                        // std::thread could not start: the started, joinable helpers must not be destroyed
        { std::lock_guard<std::mutex> lk(Q.mu); Q.stop = true; }
        Q.cv.notify_all();
        for (std::thread &t : helpers) t.join();
        throw;
    }
    const std::function<void(size_t)> progress = [&](size_t done) {
        int r = next;
        step_mark("base tasks reported done", static_cast<int>(done));
        while (r < nbk && first[r + 1] <= done) ++r;
        if (r == next) return;
        next = r;
        { std::lock_guard<std::mutex> lk(Q.mu); Q.ready = r; }
        Q.cv.notify_all();
    };
    std::exception_ptr err;
    try {
        {
            HostTimer ht("base_clustering_total");
            get_opt_hclust_batch(tasks, false, hr, &progress, fo > 0 ? &prepare : nullptr);
        }
        progress(tasks.size());
    } catch (...) { err = std::current_exception(); }
    { std::lock_guard<std::mutex> lk(Q.mu); Q.stop = true; }
    Q.cv.notify_all();
    for (std::thread &t : helpers) t.join();
    if (prof) {                                            // the tails' stage times belong to this call's table
        for (int h = 0; h < H; ++h) {
            bind_slot(tslot[h]);
            std::map<std::string, KernelStat> ts;
            ts.swap(ctx_unchecked().stats);
            bind_slot(owner);
            for (const auto &kv : ts) { KernelStat &d = mc.stats[kv.first]; d.ms += kv.second.ms; d.launches += kv.second.launches; }
        }
    }
    if (err) std::rethrow_exception(err);
    if (Q.err) std::rethrow_exception(Q.err);
}

// cross-block sMetaC on the gathered centroids, small-cluster merge and size-ordered relabel (:163-183)
void unlimited_merge(const double *means, const long long *counts, int nC, int p, long long ncells, int N_cluster, int minN, int maxN,
                     std::vector<int> &final_id, int &n_final, int hmethod = 1, double sil_thre = 0.35, double height_Ntimes = 2.0) {
    HcParams prm;                                                               // hmethod/sil.thre/height.Ntimes of y[[1]]$paras = defaults
    prm.hmethod = hmethod; prm.N_cluster = N_cluster;
    prm.minN = minN > 0 ? minN : 2;                                             // :70-72
    prm.maxN = maxN > 0 ? maxN : static_cast<int>(std::max<long long>(40, (ncells + 4999) / 5000));   // :75-77
    prm.sil_thre = sil_thre; prm.height_Ntimes = height_Ntimes;
    prm.dec_level = 3;
    DevBuf<double> dm;                                                          // (from the block cache: no hipMalloc / hipFree per call; smetac_from_means
    dm.alloc_pooled(static_cast<size_t>(nC) * p);                               //  returns with its stream drained, so the block may go back)
    dm.upload(means, static_cast<size_t>(nC) * p);
    SmResult sr = smetac_from_means(dm.p, nC, p, ncells, prm);
    final_id = sr.tf;
    int mx = *std::max_element(final_id.begin(), final_id.end());
    std::vector<long long> cnt(mx + 1, 0);
    for (int q = 0; q < nC; ++q) cnt[final_id[q]] += counts[q];
    if (N_cluster <= 0 && ncells > 10000) {                                     // :168-177
        int mn = -1;
        for (int q = 1; q <= mx; ++q) if (cnt[q] > 0 && cnt[q] < 10) { mn = q; break; }
        if (mn >= 0) {
            for (int &v : final_id) if (cnt[v] < 10) v = mn;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int q = 0; q < nC; ++q) cnt[final_id[q]] += counts[q];
        }
    }
    // x = sort(table(finalrowColor), decreasing = TRUE): size order, ties in string order of the ids (:180-183)
    std::vector<int> ids;
    for (int q = 1; q <= mx; ++q) if (cnt[q] > 0) ids.push_back(q);
    std::stable_sort(ids.begin(), ids.end(), [&](int x, int y) { return lex_less_id(x, y); });
    std::stable_sort(ids.begin(), ids.end(), [&](int x, int y) { return cnt[x] > cnt[y]; });
    std::vector<int> map(mx + 1, 0);
    for (size_t q = 0; q < ids.size(); ++q) map[ids[q]] = static_cast<int>(q) + 1;
    for (int &v : final_id) v = map[v];
    n_final = static_cast<int>(ids.size());
}

}  // namespace sharp

using namespace sharp;

namespace {

// The resident copy of a host matrix is kept between calls like every other workspace (freshly allocated HBM costs ~ 30 ms per GB
// at first touch; run_Mtimes_SHARP and testlog + SHARP call in on the same matrix again and again).
HostBlock &host_block() { return per_slot<HostBlock>(); }

struct NextHint { const float *dX = nullptr; long long nb = 0, ld = 0; };
NextHint &next_hint() { return per_slot<NextHint>(); }

// the body shared by the host-matrix, CSC and device entry points of SHARP()
int sharp_run_block(XRef dX, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells, int partition_ncells,
                    int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN, double sil_thre,
                    double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred, int *n_pred, double *viE, double *x0,
                    int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path) {
    int warn = 0;
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(pred, "sharp_SHARP: null output");
    SharpArgs a;
    a.K = ensize_K; a.reduced_ndim = reduced_ndim; a.base_ncells = base_ncells; a.partition_ncells = partition_ncells;
    a.hmethod = hmethod; a.N_cluster = N_cluster; a.enpN = enpN_cluster; a.indN = indN_cluster; a.minN = minN; a.maxN = maxN;
    a.sil_thre = sil_thre; a.height_Ntimes = height_Ntimes; a.flag = log_flag; a.projector = projector; a.rN_seed = rN_seed;
    a.want_viE = viE != nullptr; a.want_x0 = x0 != nullptr;
    a.view_to_host = viE != nullptr;
    SharpOut o;
    sharp_front_dev(dX, m, n, ld, a, o);
    warn = o.rc;
    std::copy(o.pred.begin(), o.pred.end(), pred);
    if (n_pred) *n_pred = o.n_pred;
    if (viE) o.viE.download(viE, static_cast<size_t>(n) * o.p);
    if (x0) {
        SHARP_REQUIRE(o.x0_cols <= x0_cap_cols, "sharp_SHARP: x0 buffer has too few columns");
        std::copy(o.x0.begin(), o.x0.end(), x0);
    }
    if (x0_cols) *x0_cols = o.x0_cols;
    if (p_used) *p_used = o.p;
    if (K_used) *K_used = o.K;
    if (path) *path = o.path;
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return warn;
}

}  // namespace

extern "C" {

int sharp_SHARP_dev(const float *dX, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                    int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                    double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred, int *n_pred,
                    double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path) {
    return sharp_run_block(XRef(dX), m, n, ld, ensize_K, reduced_ndim, base_ncells, partition_ncells, hmethod, N_cluster, enpN_cluster,
                           indN_cluster, minN, maxN, sil_thre, height_Ntimes, log_flag, projector, rN_seed, pred, n_pred, viE, x0,
                           x0_cap_cols, x0_cols, p_used, K_used, path);
}

int sharp_SHARP_dev64(const double *dX, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                      int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                      double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred, int *n_pred,
                      double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path) {
    XRef r;
    try { ctx(); r = dev64_ref(dX, m, n, ld); }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return sharp_run_block(r, m, n, ld, ensize_K, reduced_ndim, base_ncells, partition_ncells, hmethod, N_cluster, enpN_cluster,
                           indN_cluster, minN, maxN, sil_thre, height_Ntimes, log_flag, projector, rN_seed, pred, n_pred, viE, x0,
                           x0_cap_cols, x0_cols, p_used, K_used, path);
}

int sharp_SHARP(const double *X, int m, long long n, long long ld, int ensize_K, int reduced_ndim, int base_ncells,
                int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN, int maxN,
                double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred, int *n_pred,
                double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path) {
    HostBlock &hb = host_block();
    try {
        ctx();
        if (!X || ld < m || n < 1) throw sharp::Error(SHARP_ERR_ARG, "No expression data is provided!");
        upload_block(X, m, n, ld, hb);
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return sharp_run_block(hb.ref(), m, n, hb.ld, ensize_K, reduced_ndim, base_ncells, partition_ncells, hmethod, N_cluster, enpN_cluster,
                           indN_cluster, minN, maxN, sil_thre, height_Ntimes, log_flag, projector, rN_seed, pred, n_pred, viE, x0,
                           x0_cap_cols, x0_cols, p_used, K_used, path);
}

/* 32 or 64: how the most recent host-matrix entry point (sharp_SHARP, sharp_SHARP_csc, sharp_SHARP_unlimited*, sharp_project) stored
 * its block in HBM -- fp32 when every value survives the round trip through float, else fp64 (0: no upload yet). */
int sharp_x_storage(void) { return upload_last_storage(); }
int sharp_x_wire(void) { return upload_last_wire(); }

/* allrpinfo of the most recent SHARP_small run (R/SHARP.R:350-387: per random projection k the rowColor of every cell and the projected
 * matrix indE = tmp$mat).  enrp: n x K column-major colour indices; indE: n x (K p) row-major, projection k in columns [k p, (k+1) p). */
int sharp_last_rpinfo(int *n, int *K, int *p, int *enrp, double *indE) {
    SHARP_API_BEGIN
    ctx();
    LastSmall &L = last_small();
    SHARP_REQUIRE(L.valid, "sharp_last_rpinfo: the last SHARP call did not take the SHARP_small path (allrpinfo exists there only, R/SHARP.R:446)");
    if (n) *n = L.n;
    if (K) *K = L.K;
    if (p) *p = L.p;
    if (enrp) std::copy(L.enrp.begin(), L.enrp.end(), enrp);
    if (indE) {
        const size_t w = static_cast<size_t>(L.K) * L.p * sizeof(double);
        SHARP_HIP_CHECK(hipMemcpy2DAsync(indE, w, dws().E.p, static_cast<size_t>(L.ldE) * sizeof(double), w, static_cast<size_t>(L.n),
                                         hipMemcpyDeviceToHost, ctx().stream));
        stream_sync();
    }
    SHARP_API_END
}

/* Gives back what the host-matrix entry points keep between calls: the resident fp32 copy of the last host matrix and the pinned
 * staging buffers.  The next call simply allocates them again. */
int sharp_trim(void) {
    SHARP_API_BEGIN
    ctx();
    for_each_ready_slot([] {                             // the caller's slot and every worker slot a multi-GPU run has left behind
        SHARP_HIP_CHECK(hipDeviceSynchronize());
        drop_pending_front();
        host_block().release();
        { HostBlockPair &hp = per_slot<HostBlockPair>(); for (HostBlock &h : hp.hb) h.release(); }
        upload_release_staging();
        dws().Ebatch.release();                          // a batched SHARP_unlimited window's projections (up to 16 GB)
        dws().posbatch.release();
        if (dws().view_pinned) { (void)hipHostFree(dws().view_pinned); dws().view_pinned = nullptr; dws().view_pinned_n = 0; }   // staging of the early viE download
        rp_pc_trim();
        rp_trim();                                       // the per-chunk entry buffers of a block compacted ahead (up to 16 GB)
        // worker and helper slots (in-process multi-GPU runs, tail helpers) also give back their clustering workspaces -- tens of GB per
        // slot, which add up when several slots share one GPU; the caller's own slot keeps them (the next SHARP() call would pay for them again)
        if (cur_slot() != 0) { hc_release_workspaces(); dws().E.release(); dws().Eb.release(); dws().viE_sh.release(); dws().viE_out.release(); }
    });
    pool_clear();
    SHARP_API_END
}

/* dgCMatrix entry points (colptr: n+1 ints, rowidx: 0-based, val: fp64; canonical CSC).  sharp_csc_to_dense_dev fills a caller-owned
 * device block (m x n fp32, column stride ld >= m) that every *_dev entry point then accepts. */
int sharp_csc_to_dense_dev(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ld) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(dX && ld >= m && m > 0 && n >= 0, "sharp_csc_to_dense_dev: bad block");
    upload_csc_into_f32(colptr, rowidx, val, m, n, dX, ld);
    SHARP_API_END
}

int sharp_csc_packed_expand_dev(const long long *d_colptr, const void *d_idx, int idx_bits, const void *d_val, int val_bits, int m,
                                long long n, void *dX, long long ld, int dx_is_f64) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(d_colptr && dX && m > 0 && n >= 0 && ld >= m, "sharp_csc_packed_expand_dev: bad block");
    SHARP_REQUIRE((idx_bits == 16 && m <= 65536) || idx_bits == 32, "sharp_csc_packed_expand_dev: row indices of 16 (at most 65 536 genes) or 32 bits");
    SHARP_REQUIRE(val_bits == 8 || val_bits == 16 || val_bits == 32 || val_bits == 64, "sharp_csc_packed_expand_dev: values of 8 or 16 (unsigned integers), 32 (float) or 64 (double) bits");
    SHARP_REQUIRE(val_bits != 64 || dx_is_f64, "sharp_csc_packed_expand_dev: 64-bit values need an fp64 block");
    expand_packed_csc_dev(d_colptr, d_idx, idx_bits, d_val, val_bits, m, n, dX, ld, dx_is_f64 != 0);
    SHARP_API_END
}

int sharp_SHARP_csc(const int *colptr, const int *rowidx, const double *val, int m, long long n, int ensize_K, int reduced_ndim,
                    int base_ncells, int partition_ncells, int hmethod, int N_cluster, int enpN_cluster, int indN_cluster, int minN,
                    int maxN, double sil_thre, double height_Ntimes, int log_flag, int projector, double rN_seed, int *pred,
                    int *n_pred, double *viE, double *x0, int x0_cap_cols, int *x0_cols, int *p_used, int *K_used, int *path) {
    HostBlock &hb = host_block();
    try {
        ctx();
        if (m <= 0 || n <= 0) throw sharp::Error(SHARP_ERR_ARG, "No expression data is provided!");
        upload_block_csc(colptr, rowidx, val, m, n, hb);
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return sharp_run_block(hb.ref(), m, n, hb.ld, ensize_K, reduced_ndim, base_ncells, partition_ncells, hmethod, N_cluster, enpN_cluster,
                           indN_cluster, minN, maxN, sil_thre, height_Ntimes, log_flag, projector, rN_seed, pred, n_pred, viE, x0,
                           x0_cap_cols, x0_cols, p_used, K_used, path);
}

/* One-shot hint for a caller that runs its blocks one call at a time (a rank of the sharded run with several blocks): the block that
 * the NEXT-BUT-ONE call will bring (same genes, projector and parameters as the next call's block), already resident in HBM.  The next
 * sharp_unlimited_block*_dev call prepares it under its own tail.  dX = NULL clears the hint. */
int sharp_unlimited_next_block_dev(const float *dX_next, long long nb_next, long long ld_next) {
    NextHint &h = next_hint();
    h.dX = dX_next; h.nb = nb_next; h.ld = ld_next;
    return SHARP_OK;
}

static int unlimited_block_view_entry(const void *dXv, bool f64, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                      double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                      long long *counts, double *viE, const ViewCall *view_in) {
    SHARP_API_BEGIN
    // (a caller that arms sharp_unlimited_view_dim per block gets its E1 rows back reduced; block by block an UNSEEDED run must say which z0
    // its blocks share: sharp_unlimited_block_viewk_dev)
    const ViewCall view = view_in ? *view_in : take_view(rN_seed, ensize_K > 0 ? ensize_K : 5);
    ctx();
    SHARP_REQUIRE(pred && n_clusters && means && counts, "sharp_unlimited_block_view_dev: null output");
    std::vector<int> pr;
    std::vector<double> mn;
    std::vector<long long> cn;
    const NextHint h = next_hint();
    next_hint() = NextHint();
    const XRef dX = f64 ? dev64_ref(static_cast<const double *>(dXv), m, nb, ld) : XRef(static_cast<const float *>(dXv));
    unlimited_block_dev(dX, m, nb, ld, p, projector, ensize_K > 0 ? ensize_K : 5, rN_seed, pr, mn, cn, viE, view, flag != 0, nullptr,
                        h.dX ? XRef(h.dX) : XRef(), h.nb, h.ld);
    SHARP_REQUIRE(static_cast<int>(cn.size()) <= cap_rows, "sharp_unlimited_block_view_dev: centroid buffer too small");
    std::copy(pr.begin(), pr.end(), pred);
    std::copy(mn.begin(), mn.end(), means);
    std::copy(cn.begin(), cn.end(), counts);
    *n_clusters = static_cast<int>(cn.size());
    SHARP_API_END
}

int sharp_unlimited_block_view_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                   double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                   long long *counts, double *viE) {
    return unlimited_block_view_entry(dX, false, m, nb, ld, p, projector, ensize_K, rN_seed, flag, pred, n_clusters, means, cap_rows, counts, viE, nullptr);
}

int sharp_unlimited_block_viewk_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                    double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                    long long *counts, int view_dim, double view_seed, double *viE) {
    if (view_dim < 0 || view_dim > 4096 || (view_dim > 0 && std::fmod(view_seed, 1.0) != 0.0)) {
        sharp::set_error("sharp_unlimited_block_viewk_dev: view_dim must lie in 0 .. 4096 and view_seed must be an integer (the seed of the run's z0)");
        return SHARP_ERR_ARG;
    }
    ViewCall v;
    v.kdim = viE ? view_dim : 0;
    v.seed = view_seed;
    return unlimited_block_view_entry(dX, false, m, nb, ld, p, projector, ensize_K, rN_seed, flag, pred, n_clusters, means, cap_rows, counts, viE, &v);
}

/* the same for a resident fp64 block (TPM / CPM-like values: 16-byte aligned, even leading dimension) */
int sharp_unlimited_block_viewk_dev64(const double *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K,
                                      double rN_seed, int flag, int *pred, int *n_clusters, double *means, int cap_rows,
                                      long long *counts, int view_dim, double view_seed, double *viE) {
    if (view_dim < 0 || view_dim > 4096 || (view_dim > 0 && std::fmod(view_seed, 1.0) != 0.0)) {
        sharp::set_error("sharp_unlimited_block_viewk_dev64: view_dim must lie in 0 .. 4096 and view_seed must be an integer (the seed of the run's z0)");
        return SHARP_ERR_ARG;
    }
    ViewCall v;
    v.kdim = viE ? view_dim : 0;
    v.seed = view_seed;
    return unlimited_block_view_entry(dX, true, m, nb, ld, p, projector, ensize_K, rN_seed, flag, pred, n_clusters, means, cap_rows, counts, viE, &v);
}

int sharp_unlimited_block_dev(const float *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K, double rN_seed,
                              int *pred, int *n_clusters, double *means, int cap_rows, long long *counts) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(pred && n_clusters && means && counts, "sharp_unlimited_block_dev: null output");
    std::vector<int> pr;
    std::vector<double> mn;
    std::vector<long long> cn;
    unlimited_block_dev(dX, m, nb, ld, p, projector, ensize_K > 0 ? ensize_K : 5, rN_seed, pr, mn, cn, nullptr, ViewCall());
    SHARP_REQUIRE(static_cast<int>(cn.size()) <= cap_rows, "sharp_unlimited_block_dev: centroid buffer too small");
    std::copy(pr.begin(), pr.end(), pred);
    std::copy(mn.begin(), mn.end(), means);
    std::copy(cn.begin(), cn.end(), counts);
    *n_clusters = static_cast<int>(cn.size());
    SHARP_API_END
}

int sharp_unlimited_block_dev64(const double *dX, int m, long long nb, long long ld, int p, int projector, int ensize_K, double rN_seed,
                                int *pred, int *n_clusters, double *means, int cap_rows, long long *counts) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(pred && n_clusters && means && counts, "sharp_unlimited_block_dev64: null output");
    std::vector<int> pr;
    std::vector<double> mn;
    std::vector<long long> cn;
    unlimited_block_dev(dev64_ref(dX, m, nb, ld), m, nb, ld, p, projector, ensize_K > 0 ? ensize_K : 5, rN_seed, pr, mn, cn, nullptr, ViewCall());
    SHARP_REQUIRE(static_cast<int>(cn.size()) <= cap_rows, "sharp_unlimited_block_dev64: centroid buffer too small");
    std::copy(pr.begin(), pr.end(), pred);
    std::copy(mn.begin(), mn.end(), means);
    std::copy(cn.begin(), cn.end(), counts);
    *n_clusters = static_cast<int>(cn.size());
    SHARP_API_END
}

int sharp_unlimited_merge(const double *means, const long long *counts, int nC, int p, long long ncells, int N_cluster, int minN,
                          int maxN, int *final_id, int *n_final) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(means && counts && final_id && n_final, "sharp_unlimited_merge: null argument");
    std::vector<int> fid;
    int nf = 0;
    unlimited_merge(means, counts, nC, p, ncells, N_cluster, minN, maxN, fid, nf);
    std::copy(fid.begin(), fid.end(), final_id);
    *n_final = nf;
    SHARP_API_END
}

}  // extern "C"

// The blocks of a SHARP_unlimited call (R/SHARP_unlimited.R:125-149) on one GPU, each one the SHARP() call of :135: several large-path
// blocks whose base tasks outnumber the CUs go as ONE pipelined batch (unlimited_batch_window), in windows of at most ~16 GB of
// projections; SHARP_UNLIMITED_BATCH=0 or blocks that do not qualify: block after block, each preparing the next under its tail.
// take(b, labels, cluster means, cluster sizes) is called once per block, in block order.
static void unlimited_blocks_loop(const XRef *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m, int p, int proj, int K,
                                  double rN_seed, double *const *viE_of,   // NULL, or per block where its E1 rows go (NULL: not wanted)
                                  const ViewCall &view,
                                  const std::function<void(int, std::vector<int> &, std::vector<double> &, std::vector<long long> &)> &take,
                                  const int *gids = nullptr,     // (decision log: the blocks' indices in the caller's list; NULL: 0, 1, ...)
                                  bool small_windows = false) {  // a batch also for a few blocks whose tasks merely fill the chip (host blocks taken as they arrive)
    {
        int b = 0;
        while (b < nblocks) {                                                                                  // :125-149
            // Several large-path blocks whose base tasks outnumber the CUs: one pipelined batch (unlimited_batch_window), in windows of
            // at most ~16 GB of projections.  SHARP_UNLIMITED_BATCH=0: block after block, each preparing the next under its tail.
            int e = b;
            long long rows = 0, ntasks = 0;
            long long window_bytes = 16LL << 30;
            if (knobs().unlimited_window_mb > 0) window_bytes = static_cast<long long>(knobs().unlimited_window_mb) << 20;   // (tests: several windows)
            if (knobs().unlimited_batch) {
                while (e < nblocks && ncb[e] >= 5000 && ncb[e] < (1LL << 31) &&
                       (rows + ncb[e]) * static_cast<long long>(K) * p * 8 <= window_bytes) {
                    rows += ncb[e];
                    ntasks += static_cast<long long>(K) * ((ncb[e] + 1999) / 2000);
                    ++e;
                }
            }
            if (e - b >= 2 && ntasks > (small_windows ? 3LL * ctx().num_cu / 4 : 2LL * ctx().num_cu)) {
                struct Got { std::vector<int> pb; std::vector<double> mb; std::vector<long long> cb; };
                std::vector<Got> got(e - b);                                    // (the window's tails finish on helper threads, in any order)
                unlimited_batch_window(dX_blocks, ncb, ldb, b, e, m, p, proj, K, rN_seed, [&](int bb, const SharpOut &o) {
                    Got &g = got[bb - b];
                    unlimited_block_summary(o, ncb[bb], p, g.pb, g.mb, g.cb, viE_of ? viE_of[bb] : nullptr, view);
                }, gids);
                for (int q = b; q < e; ++q) take(q, got[q - b].pb, got[q - b].mb, got[q - b].cb);
                b = e;
                continue;
            }
            std::vector<int> pb;
            std::vector<double> mb;
            std::vector<long long> cb;
            const bool more = b + 1 < nblocks;
            unlimited_block_dev(dX_blocks[b], m, ncb[b], ldb[b], p, proj, K, rN_seed, pb, mb, cb,
                                viE_of ? viE_of[b] : nullptr, view, 1, nullptr, more ? dX_blocks[b + 1] : XRef(),
                                more ? ncb[b + 1] : 0, more ? ldb[b + 1] : 0, gids ? gids[b] : b);
            take(b, pb, mb, cb);
            ++b;
        }
    }
}

// SHARP_unlimited on resident blocks (fp32 or fp64 each)
static int unlimited_run(const XRef *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                         int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used,
                         double *viE, const ViewCall *view_in = nullptr) {
    SHARP_API_BEGIN
    const ViewCall view = view_in ? *view_in : take_view(rN_seed, ensize_K > 0 ? ensize_K : 5);   // (first: a call that fails its argument checks spends the arm too)
    ctx();
    SHARP_REQUIRE(dX_blocks && ncb && ldb && pred, "The input should be a LIST of partitioned scRNA-seq expression matrices!");
    SHARP_REQUIRE(nblocks >= 2, "SHARP is used instead of SHARP_unlimited because the length of the input is 1!");
    if (rN_seed != 0.5) SHARP_REQUIRE(std::fmod(rN_seed, 1.0) == 0.0, "The rN.seed should be an integer!");   // :80-90
    long long ncells = 0;
    for (int b = 0; b < nblocks; ++b) ncells += ncb[b];
    const int p = static_cast<int>(std::ceil(std::log2(static_cast<double>(ncells)) / (0.2 * 0.2)));           // :65-66
    const int K = ensize_K > 0 ? ensize_K : 5;                                                                 // :92-94
    std::vector<double> seeds(K);
    for (int k = 0; k < K; ++k) seeds[k] = (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + (k + 1);                    // :97-104
    step_mark("SHARP_unlimited begins, blocks", nblocks);
    shuffle_ahead(ncb[0], rN_seed);                        // (the first block's shuffle beside the projector build)
    const int proj = register_projector(build_projector(m, p, K, seeds.data()));
    std::vector<double> means;
    std::vector<long long> counts;
    std::vector<int> first(nblocks + 1, 0);
    long long off = 0;
    try {
        auto take = [&](int b, std::vector<int> &pb, std::vector<double> &mb, std::vector<long long> &cb) {
            std::copy(pb.begin(), pb.end(), pred + off);
            means.insert(means.end(), mb.begin(), mb.end());
            counts.insert(counts.end(), cb.begin(), cb.end());
            first[b + 1] = first[b] + static_cast<int>(cb.size());
            off += ncb[b];
        };
        std::vector<double *> viE_of(nblocks, nullptr);
        long long at = 0;
        for (int b = 0; b < nblocks; ++b) { if (viE) viE_of[b] = viE + static_cast<size_t>(at) * view.cols(p); at += ncb[b]; }
        unlimited_blocks_loop(dX_blocks, ncb, ldb, nblocks, m, p, proj, K, rN_seed, viE ? viE_of.data() : nullptr, view, take);
    } catch (...) { drop_projector(proj); throw; }
    step_mark("blocks done, projector dropped: merge of centroids", first[nblocks]);
    drop_projector(proj);
    std::vector<int> fid;
    int nf = 0;
    unlimited_merge(means.data(), counts.data(), first[nblocks], p, ncells, N_cluster, minN, maxN, fid, nf);
    step_mark("merge done, clusters", nf);
    {
        std::vector<long long> offs(nblocks + 1, 0);
        for (int b = 0; b < nblocks; ++b) offs[b + 1] = offs[b] + ncb[b];
        host_parallel_for(nblocks, 16, [&](int b) {              // (every block's cells through the merge's map: 0.3 ms for 500 000 cells on one thread)
            int *pb = pred + offs[b];
            const int *map = fid.data() + first[b];
            for (long long i = 0; i < ncb[b]; ++i) pb[i] = map[pb[i] - 1];
        });
    }
    if (n_pred) *n_pred = nf;
    if (p_used) *p_used = p;
    step_mark("SHARP_unlimited returns", 0);
    step_marks_dump();
    SHARP_API_END
}

// SHARP_unlimited2 (R/SHARP_unlimited2.R:29-292): SHARP_fpart per block, then ONE sMetaC over the fold-level ensemble
// clusters of all blocks -- which, like every sMetaC, only needs their centroids in E1 = enE/K.
static int unlimited2_run(const XRef *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                          int ensize_K, int reduced_ndim, int partition_ncells, int hmethod, int N_cluster, int enpN, int indN,
                          int minN, int maxN, double sil_thre, double height_Ntimes, int flag, double rN_seed, int *pred,
                          int *n_pred, int *p_used, double *viE) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(dX_blocks && ncb && ldb && pred && nblocks >= 1, "No expression data is provided!");
    if (rN_seed != 0.5) SHARP_REQUIRE(std::fmod(rN_seed, 1.0) == 0.0, "The rN.seed should be an integer!");   // :117-127
    long long ncells = 0;
    for (int b = 0; b < nblocks; ++b) ncells += ncb[b];
    const int p = reduced_ndim > 0 ? reduced_ndim : static_cast<int>(std::ceil(std::log2(static_cast<double>(ncells)) / (0.2 * 0.2)));   // :42-44
    const int K = ensize_K > 0 ? ensize_K : 5;                                                                 // :39-41
    SharpArgs fa;
    fa.partition_ncells = partition_ncells; fa.hmethod = hmethod; fa.enpN = enpN; fa.indN = indN;
    fa.minN = minN > 0 ? minN : 2;                                                                             // :51-53
    fa.maxN = maxN > 0 ? maxN : static_cast<int>(std::max<long long>(40, (ncells + 4999) / 5000));             // :54-56 (total cells)
    fa.sil_thre = sil_thre; fa.height_Ntimes = height_Ntimes;
    fa.N_cluster = 0;                                                                                          // fpart does not cut; the final sMetaC does
    std::vector<double> seeds(K);
    for (int k = 0; k < K; ++k) seeds[k] = (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + (k + 1);                    // :130-137
    step_mark("SHARP_unlimited begins, blocks", nblocks);
    const int proj = register_projector(build_projector(m, p, K, seeds.data()));
    std::vector<double> means;
    std::vector<long long> counts;
    std::vector<int> first(nblocks + 1, 0);
    long long off = 0;
    try {
        for (int b = 0; b < nblocks; ++b) {                                                                    // :146-163
            std::vector<int> pb;
            std::vector<double> mb;
            std::vector<long long> cb;
            const bool more = b + 1 < nblocks;
            unlimited_block_dev(dX_blocks[b], m, ncb[b], ldb[b], p, proj, K, rN_seed, pb, mb, cb,
                                viE ? viE + static_cast<size_t>(off) * p : nullptr, ViewCall(), flag ? 2 : 0, &fa, more ? dX_blocks[b + 1] : XRef(),
                                more ? ncb[b + 1] : 0, more ? ldb[b + 1] : 0);
            std::copy(pb.begin(), pb.end(), pred + off);
            means.insert(means.end(), mb.begin(), mb.end());
            counts.insert(counts.end(), cb.begin(), cb.end());
            first[b + 1] = first[b] + static_cast<int>(cb.size());
            off += ncb[b];
        }
    } catch (...) { drop_projector(proj); throw; }
    drop_projector(proj);
    std::vector<int> fid;
    int nf = 0;
    unlimited_merge(means.data(), counts.data(), first[nblocks], p, ncells, N_cluster, minN, maxN, fid, nf, hmethod > 0 ? hmethod : 1,
                    sil_thre >= 0 ? sil_thre : 0.35, height_Ntimes > 0 ? height_Ntimes : 2.0);               // :184-205
    off = 0;
    for (int b = 0; b < nblocks; ++b) {
        for (long long i = 0; i < ncb[b]; ++i) pred[off + i] = fid[first[b] + pred[off + i] - 1];
        off += ncb[b];
    }
    if (n_pred) *n_pred = nf;
    if (p_used) *p_used = p;
    SHARP_API_END
}

extern "C" {
// SEVERAL blocks of one rank of a sharded SHARP_unlimited run in one call (sharp_unlimited_block_dev per block, but with the base
// clustering of all of them as one pipelined batch and their tails on helper threads: 17 instead of 25 ms per 50 000-cell block).
int sharp_unlimited_blocks_dev(const void *const *dX_blocks, const int *is_f64, const long long *ncb, const long long *ldb, int nblocks, int m,
                               int p, int projector, int ensize_K, double rN_seed, int *pred, int *n_clusters, double *means, int cap_rows,
                               long long *counts) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(dX_blocks && ncb && ldb && pred && n_clusters && means && counts && nblocks >= 1, "sharp_unlimited_blocks_dev: null argument");
    SHARP_REQUIRE(p >= 1 && cap_rows >= 1, "sharp_unlimited_blocks_dev: bad sizes");
    const int K = ensize_K > 0 ? ensize_K : 5;
    std::vector<XRef> refs(nblocks);
    for (int b = 0; b < nblocks; ++b) {
        SHARP_REQUIRE(dX_blocks[b] && ncb[b] >= 1 && ldb[b] >= m, "sharp_unlimited_blocks_dev: bad block");
        refs[b] = (is_f64 && is_f64[b]) ? dev64_ref(static_cast<const double *>(dX_blocks[b]), m, ncb[b], ldb[b])
                                        : XRef(static_cast<const float *>(dX_blocks[b]));
    }
    long long off = 0;
    int rows = 0;
    unlimited_blocks_loop(refs.data(), ncb, ldb, nblocks, m, p, projector, K, rN_seed, nullptr, ViewCall(),
                          [&](int b, std::vector<int> &pb, std::vector<double> &mb, std::vector<long long> &cb) {
        SHARP_REQUIRE(rows + static_cast<int>(cb.size()) <= cap_rows, "sharp_unlimited_blocks_dev: centroid buffer too small");
        std::copy(pb.begin(), pb.end(), pred + off);
        std::copy(mb.begin(), mb.end(), means + static_cast<size_t>(rows) * p);
        std::copy(cb.begin(), cb.end(), counts + rows);
        n_clusters[b] = static_cast<int>(cb.size());
        rows += static_cast<int>(cb.size());
        off += ncb[b];
    });
    SHARP_API_END
}
}  // extern "C"

namespace {
// host blocks of a list-of-matrices call: each is stored as fp32 or fp64 on its own merits
struct HostBlocks {
    std::vector<HostBlock> bufs;
    std::vector<XRef> refs;
    std::vector<long long> lds;
    int upload(const double *const *X_blocks, const long long *ncb, int nblocks, int m) {
        try {
            ctx();
            if (!X_blocks || !ncb || nblocks < 1) throw sharp::Error(SHARP_ERR_ARG, "No expression data is provided!");
            bufs.resize(nblocks);
            int any64 = 0;
            for (int b = 0; b < nblocks; ++b) {
                upload_block(X_blocks[b], m, ncb[b], m, bufs[b]);
                refs.push_back(bufs[b].ref());
                lds.push_back(bufs[b].ld);
                any64 |= bufs[b].f64;
            }
            (void)any64;
        }
        catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
        catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
        return SHARP_OK;
    }
};
// One block of a list-of-blocks call, wherever it lives: a dense host matrix (R's numeric matrix), the three slots of a dgCMatrix,
// or a block already resident on one of the call's GPUs.
struct BlockSrc {
    enum Kind { HostDense, HostCsc, DevF32, DevF64 } kind = HostDense;
    const double *host = nullptr;                          // HostDense: m x n doubles, column stride m
    const int *colptr = nullptr, *rowidx = nullptr;        // HostCsc: @p (n + 1), @i
    const double *val = nullptr;                           //          @x
    const void *dev = nullptr;                             // DevF32 / DevF64
    long long ld = 0;
    int worker = -1;                                       // DevF32 / DevF64: index into the call's device list (the GPU the block lives on)
    long long n = 0;
    bool on_host() const { return kind == HostDense || kind == HostCsc; }
};

HostBlockPair &host_block_pair() { return per_slot<HostBlockPair>(); }

// What the last in-process multi-device call did when: one row per block -- worker, block, upload start / end, clustering start / end,
// seconds since the call began (sharp_multi_timeline; profiles/r04_multi_timeline.txt).
struct MultiTimeline { std::mutex mu; std::vector<double> rows; };
MultiTimeline &multi_timeline() { static MultiTimeline *t = new MultiTimeline; return *t; }
double wall_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

// SHARP_unlimited over several GPUs inside one process (R/SHARP_unlimited.R:125-183): the serial block loop of the reference
// (:125-163) is dealt out, block b to worker b mod W (a resident block: to the worker of the GPU it lives on).  A worker is a host thread
// bound to a device slot of its own (context, streams, workspaces, projector handles: common.hpp): it builds the K projectors on its GPU
// (a pure function of m, p and the seeds, :97-104; p from the GLOBAL cell count, :65-66) and clusters its blocks one after the other, each
// prepared under the previous one's tail when it is already resident.  Host blocks -- dense or sparse -- reach the GPU through a second
// thread per worker with a slot of its own (own stream, pinned staging, three resident copies in rotation): block i + W crosses PCIe while
// block i is clustered.  Per block the worker keeps the labels and the per-cluster centroid means and sizes -- all the final sMetaC uses
// of E1 (R/sMetaC.R:58-63); the tables (a few hundred rows x p doubles per block) meet in host memory, worker 0 runs the merge (:163-183)
// once the others are done, and every block's labels are mapped through it.  No other data crosses between devices.
int unlimited_run_multi(const std::vector<BlockSrc> &blocks, int m, int ensize_K, int N_cluster, int minN, int maxN, double rN_seed,
                        const int *devices, int ndev, int *pred, int *n_pred, int *p_used, double *viE) {
    SHARP_API_BEGIN
    const ViewCall view = take_view(rN_seed, ensize_K > 0 ? ensize_K : 5);       // (by value into every worker: one z0 on all devices)
    const int nblocks = static_cast<int>(blocks.size());
    SHARP_REQUIRE(pred, "The input should be a LIST of partitioned scRNA-seq expression matrices!");
    SHARP_REQUIRE(nblocks >= 2, "SHARP is used instead of SHARP_unlimited because the length of the input is 1!");
    SHARP_REQUIRE(devices && ndev >= 1 && ndev <= 16, "SHARP_unlimited: the device list must name between 1 and 16 GPUs");
    if (rN_seed != 0.5) SHARP_REQUIRE(std::fmod(rN_seed, 1.0) == 0.0, "The rN.seed should be an integer!");   // :80-90
    long long ncells = 0;
    for (int b = 0; b < nblocks; ++b) {
        const BlockSrc &s = blocks[b];
        SHARP_REQUIRE(s.n >= 3 && (s.kind == BlockSrc::HostDense ? s.host != nullptr : s.kind == BlockSrc::HostCsc ? s.colptr != nullptr : s.dev != nullptr),
                      "SHARP_unlimited: empty block");
        SHARP_REQUIRE(s.on_host() || (s.worker >= 0 && s.worker < ndev), "SHARP_unlimited: a resident block names a device outside the device list");
        ncells += s.n;
    }
    const int p = static_cast<int>(std::ceil(std::log2(static_cast<double>(ncells)) / (0.2 * 0.2)));           // :65-66
    const int K = ensize_K > 0 ? ensize_K : 5;                                                                 // :92-94
    std::vector<double> seeds(K);
    for (int k = 0; k < K; ++k) seeds[k] = (rN_seed == 0.5) ? 0.5 : 50 + rN_seed + (k + 1);                    // :97-104
    bool resident = false;
    for (const BlockSrc &s : blocks) resident |= !s.on_host();
    const int W = resident ? ndev : std::min(ndev, nblocks);
    SHARP_REQUIRE(rN_seed != 0.5 || W == 1, "SHARP_unlimited on several GPUs needs a seed: unseeded projectors would differ between the devices");
    std::vector<long long> cell0(nblocks + 1, 0);
    for (int b = 0; b < nblocks; ++b) cell0[b + 1] = cell0[b] + blocks[b].n;
    std::vector<std::vector<int>> mine(W);                // the blocks of each worker, in block order
    for (int b = 0; b < nblocks; ++b) mine[blocks[b].on_host() ? b % W : blocks[b].worker].push_back(b);
    std::vector<int> occ(W, 0);                           // which occurrence of its device a worker is (a device may be listed repeatedly)
    for (int w = 0; w < W; ++w) for (int v = 0; v < w; ++v) occ[w] += devices[v] == devices[w];
    std::vector<std::vector<int>> pb(nblocks);
    std::vector<std::vector<double>> mb(nblocks);
    std::vector<std::vector<long long>> cb(nblocks);
    std::vector<std::string> err(W);
    std::vector<int> rcs(W, SHARP_OK);
    const double t_begin = wall_s();
    std::vector<double> tl(static_cast<size_t>(nblocks) * 6, 0.0);
    std::mutex done_mu;
    std::condition_variable done_cv;
    int done_workers = 0;
    std::atomic<int> failed{0};
    std::vector<int> fid;
    int nf = 0;
    std::vector<int> first(nblocks + 1, 0);
    std::vector<double> means;
    std::vector<long long> counts;

    auto worker = [&](int w) {
        // ---- the upload thread of this worker: host blocks into the resident copies of its own slot, in block order
        struct Feed {
            std::mutex mu;
            std::condition_variable cv;
            std::vector<char> ready;                      // per position in mine[w]: uploaded
            std::vector<XRef> ref;
            std::vector<long long> ld;
            int consumed = 0;                             // host blocks the compute thread has finished with
            bool stop = false;
            std::string err;
            int rc = SHARP_OK;
        } feed;
        const std::vector<int> &my = mine[w];
        feed.ready.assign(my.size(), 0);
        feed.ref.resize(my.size());
        feed.ld.assign(my.size(), 0);
        std::vector<int> hostpos;                         // positions in `my` that hold host blocks
        for (size_t i = 0; i < my.size(); ++i) if (blocks[my[i]].on_host()) hostpos.push_back(static_cast<int>(i));
        // resident copies the upload rotates through: three (block after block, each prepared under the previous one's tail: the default), four when
        // SHARP_HOST_GROUP >= 2 asks for groups of arrived blocks as one pipelined batch (measured: no gain at two or three blocks)
        long long big = 0;
        for (int i : hostpos) big = std::max<long long>(big, blocks[my[i]].n);
        const int group_max = std::max(1, std::min(knobs().host_group, kHostRing - 1));
        // (three when block follows block: with two, block i + 2 could only start its upload when block i was done, i.e. when block i + 1 began -- so
        // no block was ever on the GPU in time to have its front prepared under its predecessor's tail, and a cfg3 block cost 26 ms instead of 20)
        const int ring = static_cast<double>(big) * m * 4.0 > 8e9 ? 2 : (group_max == 1 ? 3 : kHostRing);
        std::thread up;
        if (!hostpos.empty())
            up = std::thread([&] {
                try {
                    init_slot(acquire_slot(devices[w], occ[w], 1), devices[w], true);   // (high-priority stream: a block's expansion takes the CUs the clustering leaves free ahead of the next chunk's distance GEMM, which nothing waits for yet)
                    HostBlockPair &P = host_block_pair();
                    for (size_t h = 0; h < hostpos.size(); ++h) {
                        {   // copy h % ring is free once the compute thread is done with host block h - ring
                            std::unique_lock<std::mutex> lk(feed.mu);
                            feed.cv.wait(lk, [&] { return feed.stop || feed.consumed + ring > static_cast<int>(h); });
                            if (feed.stop) return;
                        }
                        const int i = hostpos[h], b = my[i];
                        const BlockSrc &s = blocks[b];
                        HostBlock &hb = P.hb[h % ring];
                        tl[static_cast<size_t>(b) * 6 + 2] = wall_s() - t_begin;
                        if (s.kind == BlockSrc::HostDense) upload_block(s.host, m, s.n, m, hb);
                        else upload_block_csc(s.colptr, s.rowidx, s.val, m, s.n, hb);
                        stream_sync();
                        tl[static_cast<size_t>(b) * 6 + 3] = wall_s() - t_begin;
                        std::lock_guard<std::mutex> lk(feed.mu);
                        feed.ref[i] = hb.ref(); feed.ld[i] = hb.ld; feed.ready[i] = 1;
                        feed.cv.notify_all();
                    }
                }
                catch (const sharp::Error &e) { std::lock_guard<std::mutex> lk(feed.mu); feed.err = e.what(); feed.rc = e.code; feed.cv.notify_all(); }
                catch (const std::exception &e) { std::lock_guard<std::mutex> lk(feed.mu); feed.err = e.what(); feed.rc = SHARP_ERR; feed.cv.notify_all(); }
            });
        auto stop_feed = [&] {
            { std::lock_guard<std::mutex> lk(feed.mu); feed.stop = true; }
            feed.cv.notify_all();
            if (up.joinable()) up.join();
        };
        // where block position i of this worker is, once it is on the GPU (waits for its upload; a resident block is there already)
        auto block_ref = [&](size_t i, bool wait, XRef &ref, long long &ld) -> bool {
            const BlockSrc &s = blocks[my[i]];
            if (!s.on_host()) {
                ref = s.kind == BlockSrc::DevF64 ? dev64_ref(static_cast<const double *>(s.dev), m, s.n, s.ld) : XRef(static_cast<const float *>(s.dev));
                ld = s.ld;
                return true;
            }
            std::unique_lock<std::mutex> lk(feed.mu);
            if (wait) feed.cv.wait(lk, [&] { return feed.ready[i] || feed.rc != SHARP_OK; });
            if (feed.rc != SHARP_OK) throw sharp::Error(feed.rc, feed.err);
            if (!feed.ready[i]) return false;
            ref = feed.ref[i]; ld = feed.ld[i];
            return true;
        };
        try {
            init_slot(acquire_slot(devices[w], occ[w], 0), devices[w]);
            const int proj = my.empty() ? 0 : register_projector(build_projector(m, p, K, seeds.data()));
            try {
                if (hostpos.empty() && my.size() >= 2) {
                    // all of this worker's blocks are on its GPU already: one call for the lot (pipelined batch windows where the blocks
                    // qualify, tails on helper threads), instead of block after block
                    std::vector<XRef> refs(my.size());
                    std::vector<long long> nn(my.size()), ll(my.size());
                    std::vector<double *> vo(my.size(), nullptr);
                    for (size_t i = 0; i < my.size(); ++i) {
                        block_ref(i, true, refs[i], ll[i]);
                        nn[i] = blocks[my[i]].n;
                        if (viE) vo[i] = viE + static_cast<size_t>(cell0[my[i]]) * view.cols(p);
                    }
                    const double t_go = wall_s() - t_begin;
                    unlimited_blocks_loop(refs.data(), nn.data(), ll.data(), static_cast<int>(my.size()), m, p, proj, K, rN_seed, viE ? vo.data() : nullptr, view,
                                          [&](int i, std::vector<int> &a, std::vector<double> &c, std::vector<long long> &d) {
                        const int b = my[i];
                        pb[b].swap(a); mb[b].swap(c); cb[b].swap(d);
                        tl[static_cast<size_t>(b) * 6 + 0] = w; tl[static_cast<size_t>(b) * 6 + 1] = b;
                        tl[static_cast<size_t>(b) * 6 + 4] = t_go; tl[static_cast<size_t>(b) * 6 + 5] = wall_s() - t_begin;
                    }, my.data());
                }
                for (size_t i = hostpos.empty() && my.size() >= 2 ? my.size() : 0; i < my.size(); ++i) {
                    if (failed.load()) break;
                    const int b = my[i];
                    XRef ref, nref;
                    long long ld = 0, nld = 0;
                    block_ref(i, true, ref, ld);
                    // SHARP_HOST_GROUP >= 2: the blocks that arrived go TOGETHER as one pipelined batch (unlimited_batch_window).  Whatever the
                    // grouping, every block is the same SHARP() call: same labels.
                    if (group_max > 1 && ring > 2 && blocks[b].on_host()) {
                        std::vector<XRef> refs{ref};
                        std::vector<long long> nn{blocks[b].n}, ll{ld};
                        size_t j = i + 1;
                        // (the SECOND block of a group is waited for -- one upload, 15-20 ms, against the 25 ms a lone block costs while half of the
                        // chip idles; its successor's upload then runs under the pair's 44 ms -- a third is taken only if it is there already)
                        while (j < my.size() && static_cast<int>(j - i) < std::min(group_max, ring - 1) && blocks[my[j]].on_host() &&
                               block_ref(j, j - i < 2 && blocks[my[j]].n >= 5000 && blocks[b].n >= 5000, nref, nld)) {
                            refs.push_back(nref); nn.push_back(blocks[my[j]].n); ll.push_back(nld);
                            ++j;
                        }
                        if (j - i >= 2) {
                            std::vector<double *> vo(j - i, nullptr);
                            for (size_t q = i; q < j; ++q) if (viE) vo[q - i] = viE + static_cast<size_t>(cell0[my[q]]) * view.cols(p);
                            const double t_go = wall_s() - t_begin;
                            unlimited_blocks_loop(refs.data(), nn.data(), ll.data(), static_cast<int>(j - i), m, p, proj, K, rN_seed, viE ? vo.data() : nullptr, view,
                                                  [&](int q, std::vector<int> &a, std::vector<double> &c, std::vector<long long> &d) {
                                const int bb = my[i + q];
                                pb[bb].swap(a); mb[bb].swap(c); cb[bb].swap(d);
                                tl[static_cast<size_t>(bb) * 6 + 0] = w; tl[static_cast<size_t>(bb) * 6 + 1] = bb;
                                tl[static_cast<size_t>(bb) * 6 + 4] = t_go; tl[static_cast<size_t>(bb) * 6 + 5] = wall_s() - t_begin;
                            }, my.data() + i, true);
                            {
                                std::lock_guard<std::mutex> lk(feed.mu);
                                feed.consumed += static_cast<int>(j - i);
                                feed.cv.notify_all();
                            }
                            i = j - 1;
                            continue;
                        }
                    }
                    // the next block's front goes under this block's tail if that block is on the GPU by now (a resident block always is)
                    const bool more = i + 1 < my.size() && block_ref(i + 1, false, nref, nld);
                    tl[static_cast<size_t>(b) * 6 + 4] = wall_s() - t_begin;
                    unlimited_block_dev(ref, m, blocks[b].n, ld, p, proj, K, rN_seed, pb[b], mb[b], cb[b],
                                        viE ? viE + static_cast<size_t>(cell0[b]) * view.cols(p) : nullptr,   // E1 rows of this block (:153), viewflag only
                                        view, 1, nullptr, more ? nref : XRef(), more ? blocks[my[i + 1]].n : 0, more ? nld : 0, b);
                    tl[static_cast<size_t>(b) * 6 + 5] = wall_s() - t_begin;
                    tl[static_cast<size_t>(b) * 6 + 0] = w; tl[static_cast<size_t>(b) * 6 + 1] = b;
                    if (blocks[b].on_host()) {
                        std::lock_guard<std::mutex> lk(feed.mu);
                        ++feed.consumed;
                        feed.cv.notify_all();
                    }
                }
            } catch (...) { drop_pending_front(); if (proj) drop_projector(proj); throw; }
            drop_pending_front();
            if (proj) drop_projector(proj);
            stream_sync();
            stop_feed();
            if (w == 0) {
                // the one exchange step: worker 0 waits for the others, then the centroid tables in global block order and the merge on its GPU
                {
                    std::unique_lock<std::mutex> lk(done_mu);
                    done_cv.wait(lk, [&] { return done_workers == W - 1; });
                }
                if (!failed.load()) {
                    for (int b = 0; b < nblocks; ++b) {
                        means.insert(means.end(), mb[b].begin(), mb[b].end());
                        counts.insert(counts.end(), cb[b].begin(), cb[b].end());
                        first[b + 1] = first[b] + static_cast<int>(cb[b].size());
                    }
                    unlimited_merge(means.data(), counts.data(), first[nblocks], p, ncells, N_cluster, minN, maxN, fid, nf);
                    stream_sync();
                }
            }
        }
        catch (const sharp::Error &e) { err[w] = e.what(); rcs[w] = e.code; failed.store(1); stop_feed(); }
        catch (const std::exception &e) { err[w] = e.what(); rcs[w] = SHARP_ERR; failed.store(1); stop_feed(); }
        if (w != 0) {
            std::lock_guard<std::mutex> lk(done_mu);
            ++done_workers;
            done_cv.notify_all();
        } else if (rcs[0] != SHARP_OK) {                  // (worker 0 failed before it waited: the others still have to be counted out)
            std::unique_lock<std::mutex> lk(done_mu);
            done_cv.wait(lk, [&] { return done_workers == W - 1; });
        }
    };
    {
        std::vector<std::thread> th;
        for (int w = 0; w < W; ++w) th.emplace_back(worker, w);
        for (auto &t : th) t.join();
    }
    {
        MultiTimeline &T = multi_timeline();
        std::lock_guard<std::mutex> lk(T.mu);
        T.rows = tl;
    }
    step_marks_dump();
    for (int w = 0; w < W; ++w)
        if (rcs[w] != SHARP_OK) throw sharp::Error(rcs[w], "device " + std::to_string(devices[w]) + ": " + err[w]);
    long long off = 0;
    for (int b = 0; b < nblocks; ++b) {                                         // labels scattered back (:165-183 through the merge's map)
        for (long long i = 0; i < blocks[b].n; ++i) pred[off + i] = fid[first[b] + pb[b][i] - 1];
        off += blocks[b].n;
    }
    if (n_pred) *n_pred = nf;
    if (p_used) *p_used = p;
    SHARP_API_END
}

std::vector<BlockSrc> dense_sources(const double *const *X_blocks, const long long *ncb, int nblocks) {
    std::vector<BlockSrc> v;
    for (int b = 0; X_blocks && ncb && b < nblocks; ++b) { BlockSrc s; s.kind = BlockSrc::HostDense; s.host = X_blocks[b]; s.n = ncb[b]; v.push_back(s); }
    return v;
}
std::vector<BlockSrc> csc_sources(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb, int nblocks) {
    std::vector<BlockSrc> v;
    for (int b = 0; colptr && rowidx && val && ncb && b < nblocks; ++b) {
        BlockSrc s;
        s.kind = BlockSrc::HostCsc; s.colptr = colptr[b]; s.rowidx = rowidx[b]; s.val = val[b]; s.n = ncb[b];
        v.push_back(s);
    }
    return v;
}
// the devices of a host-block call: the caller's list, else SHARP_DEVICES, else the caller's own device (one worker: upload and clustering still overlap)
std::vector<int> call_devices(const int *devices, int ndevices) {
    if (devices && ndevices > 0) return std::vector<int>(devices, devices + ndevices);
    if (!knobs().devices.empty()) return knobs().devices;
    return std::vector<int>(1, ctx().device);
}

std::vector<XRef> f32_refs(const float *const *dX_blocks, int nblocks) {
    std::vector<XRef> r;
    for (int b = 0; dX_blocks && b < nblocks; ++b) r.emplace_back(dX_blocks[b]);
    return r;
}
}  // namespace

extern "C" {

int sharp_SHARP_unlimited_view_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                   int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used,
                                   double *viE) {
    const std::vector<XRef> refs = f32_refs(dX_blocks, nblocks);
    return unlimited_run(dX_blocks ? refs.data() : nullptr, ncb, ldb, nblocks, m, ensize_K, N_cluster, minN, maxN, rN_seed, pred, n_pred, p_used, viE);
}

int sharp_unlimited_view_dim(int kdim) {
    SHARP_API_BEGIN
    SHARP_REQUIRE(kdim >= 0 && kdim <= 4096, "sharp_unlimited_view_dim: the view dimension must lie in 0 .. 4096");
    tl_view_pending = kdim;
    SHARP_API_END
}

int sharp_SHARP_unlimited_viewk_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                    int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used,
                                    int view_dim, double *viE) {
    if (view_dim < 0 || view_dim > 4096) { sharp::set_error("sharp_SHARP_unlimited_viewk_dev: the view dimension must lie in 0 .. 4096"); return SHARP_ERR_ARG; }
    tl_view_pending = 0;                                                       // (the explicit argument wins over an earlier arm)
    const ViewCall view = make_view(viE ? view_dim : 0, rN_seed, ensize_K > 0 ? ensize_K : 5);
    const std::vector<XRef> refs = f32_refs(dX_blocks, nblocks);
    return unlimited_run(dX_blocks ? refs.data() : nullptr, ncb, ldb, nblocks, m, ensize_K, N_cluster, minN, maxN, rN_seed, pred, n_pred, p_used, viE, &view);
}

int sharp_SHARP_unlimited2_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                               int ensize_K, int reduced_ndim, int partition_ncells, int hmethod, int N_cluster, int enpN, int indN,
                               int minN, int maxN, double sil_thre, double height_Ntimes, int flag, double rN_seed, int *pred,
                               int *n_pred, int *p_used, double *viE) {
    const std::vector<XRef> refs = f32_refs(dX_blocks, nblocks);
    return unlimited2_run(dX_blocks ? refs.data() : nullptr, ncb, ldb, nblocks, m, ensize_K, reduced_ndim, partition_ncells, hmethod, N_cluster,
                          enpN, indN, minN, maxN, sil_thre, height_Ntimes, flag, rN_seed, pred, n_pred, p_used, viE);
}

int sharp_SHARP_unlimited2(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K, int reduced_ndim,
                           int partition_ncells, int hmethod, int N_cluster, int enpN, int indN, int minN, int maxN, double sil_thre,
                           double height_Ntimes, int flag, double rN_seed, int *pred, int *n_pred, int *p_used, double *viE) {
    HostBlocks H;
    if (const int rc = H.upload(X_blocks, ncb, nblocks, m)) return rc;
    return unlimited2_run(H.refs.data(), ncb, H.lds.data(), nblocks, m, ensize_K, reduced_ndim, partition_ncells, hmethod,
                          N_cluster, enpN, indN, minN, maxN, sil_thre, height_Ntimes, flag, rN_seed, pred, n_pred, p_used, viE);
}

int sharp_SHARP_unlimited_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                              int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used) {
    return sharp_SHARP_unlimited_view_dev(dX_blocks, ncb, ldb, nblocks, m, ensize_K, N_cluster, minN, maxN, rN_seed, pred, n_pred,
                                          p_used, nullptr);
}

int sharp_SHARP_unlimited_view(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K, int N_cluster,
                               int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used, double *viE) {
    // A list of host matrices never sits in HBM as a whole: the blocks cross PCIe one after the other into a ring of resident copies while the
    // earlier ones are clustered (unlimited_run_multi), on the caller's GPU or dealt to the GPUs of SHARP_DEVICES=0,1,2,... (one host thread
    // each; an unseeded call stays on the caller's GPU: every device would draw its own projectors).
    if (!X_blocks || !ncb || nblocks < 1) { sharp::set_error("No expression data is provided!"); return SHARP_ERR_ARG; }
    std::vector<int> dv;
    try { dv = call_devices(nullptr, 0); }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    if (rN_seed == 0.5 && dv.size() > 1) dv.resize(1);
    return unlimited_run_multi(dense_sources(X_blocks, ncb, nblocks), m, ensize_K, N_cluster, minN, maxN, rN_seed, dv.data(),
                               static_cast<int>(dv.size()), pred, n_pred, p_used, viE);
}

int sharp_SHARP_unlimited_multi(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K, int N_cluster,
                                int minN, int maxN, double rN_seed, const int *devices, int ndevices, int *pred, int *n_pred, int *p_used,
                                double *viE) {
    if (!X_blocks || !ncb) { sharp::set_error("The input should be a LIST of partitioned scRNA-seq expression matrices!"); return SHARP_ERR_ARG; }
    return unlimited_run_multi(dense_sources(X_blocks, ncb, nblocks), m, ensize_K, N_cluster, minN, maxN, rN_seed, devices, ndevices, pred, n_pred, p_used, viE);
}

/* a list of dgCMatrix blocks: only the non-zeros of a block cross PCIe, block i + W while block i is clustered */
int sharp_SHARP_unlimited_csc_multi(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                                    int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN, double rN_seed,
                                    const int *devices, int ndevices, int *pred, int *n_pred, int *p_used, double *viE) {
    if (!colptr || !rowidx || !val || !ncb) { sharp::set_error("The input should be a LIST of partitioned scRNA-seq expression matrices!"); return SHARP_ERR_ARG; }
    std::vector<int> dv;
    try { dv = call_devices(devices, ndevices); }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    if (rN_seed == 0.5 && dv.size() > 1) dv.resize(1);
    return unlimited_run_multi(csc_sources(colptr, rowidx, val, ncb, nblocks), m, ensize_K, N_cluster, minN, maxN, rN_seed, dv.data(),
                               static_cast<int>(dv.size()), pred, n_pred, p_used, viE);
}
int sharp_SHARP_unlimited_csc(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                              int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN, double rN_seed, int *pred,
                              int *n_pred, int *p_used, double *viE) {
    return sharp_SHARP_unlimited_csc_multi(colptr, rowidx, val, ncb, nblocks, m, ensize_K, N_cluster, minN, maxN, rN_seed, nullptr, 0,
                                           pred, n_pred, p_used, viE);
}

/* blocks already resident on the GPUs of the device list: block b lives on devices[device_of_block[b]] */
int sharp_SHARP_unlimited_multi_dev(const void *const *dX_blocks, const int *is_f64, const long long *ncb, const long long *ldb,
                                    const int *device_of_block, int nblocks, int m, int ensize_K, int N_cluster, int minN, int maxN,
                                    double rN_seed, const int *devices, int ndevices, int *pred, int *n_pred, int *p_used, double *viE) {
    if (!dX_blocks || !ncb || !ldb || !device_of_block) { sharp::set_error("The input should be a LIST of partitioned scRNA-seq expression matrices!"); return SHARP_ERR_ARG; }
    std::vector<BlockSrc> v;
    for (int b = 0; b < nblocks; ++b) {
        BlockSrc s;
        s.kind = (is_f64 && is_f64[b]) ? BlockSrc::DevF64 : BlockSrc::DevF32;
        s.dev = dX_blocks[b]; s.ld = ldb[b]; s.worker = device_of_block[b]; s.n = ncb[b];
        v.push_back(s);
    }
    return unlimited_run_multi(v, m, ensize_K, N_cluster, minN, maxN, rN_seed, devices, ndevices, pred, n_pred, p_used, viE);
}

/* the last in-process multi-device call, block by block: rows of (worker, block, upload start, upload end, clustering start, clustering end),
 * seconds since the call began */
int sharp_multi_timeline(double *rows, int cap_rows, int *nrows) {
    MultiTimeline &T = multi_timeline();
    std::lock_guard<std::mutex> lk(T.mu);
    const int n = static_cast<int>(T.rows.size() / 6);
    if (nrows) *nrows = n;
    if (rows) std::copy(T.rows.begin(), T.rows.begin() + static_cast<size_t>(std::min(n, cap_rows)) * 6, rows);
    return SHARP_OK;
}

int sharp_SHARP_unlimited(const double *const *X_blocks, const long long *ncb, int nblocks, int m, int ensize_K, int N_cluster,
                          int minN, int maxN, double rN_seed, int *pred, int *n_pred, int *p_used) {
    return sharp_SHARP_unlimited_view(X_blocks, ncb, nblocks, m, ensize_K, N_cluster, minN, maxN, rN_seed, pred, n_pred, p_used, nullptr);
}

}  // extern "C"
