// hclust.hip -- batched get_opt_hclust on the GPU (R/get_opt_hclust.R:33-244):
//   a3  distance build      row_prep + fp64-MFMA correlation GEMM (linalg.hip)       :66-74
//   a4  stats::hclust       one 512-thread workgroup per task, NN-list algorithm      :76-83
//   a5  cutree k=min..max, median silhouette, get_CH("1-corr"), model selection        :90-231
// Third-party algorithms restated (not vendored by the reference): stats::hclust's Fortran NN-list
// agglomeration with Lance-Williams updates (fp64, same operation order, lowest-index tie-breaks),
// cutree's first-appearance numbering, cluster::silhouette, clues::get_CH per SURVEY.md App. A.4-A.6.
#include "hclust.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <type_traits>

#include "linalg.hpp"

namespace sharp {

struct HcMeta {
    int n, p, nld, kmin, kmax, nk, kpad, method;
    int symmetric, pad0;
    long long oD, oD0;        // working distance matrix; pristine copy (symmetric tasks only)
    long long oCr, oCt, oNrm;
    long long oM;             // ia / ib / height: n entries per task
    long long oLab;           // nk * n ints
    long long oH, oT, oG;     // n * kpad doubles each
    long long oCSt, oQ;       // p * kpad, kpad * kpad
    long long oOut;           // msil[nk] then CH[nk]
    const double *nn;         // row minima per 128-column tile written by the distance GEMM ([nld / 128 slots][nld rows], at most 16 slots); nullptr: scan D
};

constexpr int HC_THREADS = 512;
constexpr double HC_INF = 1.0e300;
constexpr int HC_RS = 16;   // loads in flight per lane in a rescan pass

struct MinPair { double v; int i; };
__device__ __forceinline__ MinPair mp_better(MinPair a, MinPair b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
// Wave-wide lexicographic min of (value, index) with DPP moves (no LDS traffic: a ds_bpermute butterfly costs six dependent
// LDS round trips, which was 40 % of the unloaded merge latency).  The result is valid in lane 63 and broadcast from there.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ MinPair mp_dpp_step(MinPair x) {
    const int lo = __double2loint(x.v), hi = __double2hiint(x.v);
    // lanes outside ROW_MASK (and lanes whose source is invalid) keep their own value: op(x, x) = x
    const int ylo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int yhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    MinPair y;
    y.i = __builtin_amdgcn_update_dpp(x.i, x.i, CTRL, ROW_MASK, 0xf, false);
    y.v = __hiloint2double(yhi, ylo);
    return mp_better(x, y);
}
__device__ __forceinline__ MinPair mp_wave(MinPair x) {
    x = mp_dpp_step<0xB1, 0xf>(x);     // quad_perm [1,0,3,2]
    x = mp_dpp_step<0x4E, 0xf>(x);     // quad_perm [2,3,0,1]
    x = mp_dpp_step<0x141, 0xf>(x);    // row_half_mirror
    x = mp_dpp_step<0x140, 0xf>(x);    // row_mirror: every lane of a 16-lane row holds the row's result
    x = mp_dpp_step<0x142, 0xa>(x);    // row_bcast15 into rows 1 and 3
    x = mp_dpp_step<0x143, 0xc>(x);    // row_bcast31 into rows 2 and 3: lane 63 holds the wave's result
    MinPair r;
    r.i = __builtin_amdgcn_readlane(x.i, 63);
    r.v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x.v), 63), __builtin_amdgcn_readlane(__double2loint(x.v), 63));
    return r;
}
// block-wide min; pv/pi: LDS scratch [32]; every thread returns the result.  The caller must have a
// barrier between two uses of the same scratch (there always is one in the merge loop).
__device__ __forceinline__ MinPair mp_block(MinPair x, double *pv, int *pi) {
    x = mp_wave(x);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { pv[w] = x.v; pi[w] = x.i; }
    __syncthreads();
    MinPair r; r.v = pv[0]; r.i = pi[0];
    const int nw = blockDim.x >> 6;
    for (int q = 1; q < nw; ++q) { MinPair y; y.v = pv[q]; y.i = pi[q]; r = mp_better(r, y); }
    return r;
}

__device__ __forceinline__ double lance_williams(int method, double d1, double d2, double d12, double mi, double mj, double mk) {
    switch (method) {
        case 1: case 8: {   // ward.D / ward.D2 (squared input)
            double dn = (mi + mk) * d1 + (mj + mk) * d2 - mk * d12;
            return dn / (mi + mj + mk);
        }
        case 2: return d1 < d2 ? d1 : d2;
        case 3: return d1 > d2 ? d1 : d2;
        case 4: return (mi * d1 + mj * d2) / (mi + mj);
        case 5: return (d1 + d2) / 2;
        case 6: return ((d1 + d2) - d12 / 2) / 2;
        default: return (mi * d1 + mj * d2 - mi * mj * d12 / (mi + mj)) / (mi + mj);
    }
}

// ---------------------------------------------------------------------------------------------
// a4: agglomeration.  State in LDS: disnn (nearest neighbour to the right), nn, membr, flag.
// D is the full symmetric matrix in HBM (row reads coalesced; the mirrored column write is strided).
// ---------------------------------------------------------------------------------------------
// GS: the nearest-neighbour state lives in global memory (gstate, gstride bytes per task) instead of LDS: tasks of more than
// kHcLdsMaxN observations (a cross-block sMetaC over thousands of block-level clusters).  Same code, same order of operations; the
// workgroup barriers order the global accesses as they order the LDS ones (all waves of a workgroup share the CU's L1).
template <bool GS>
__global__ __launch_bounds__(HC_THREADS) void hclust_kernel(const HcMeta *__restrict__ metas, double *__restrict__ Dall,
                                                            int *__restrict__ ia_all, int *__restrict__ ib_all,
                                                            double *__restrict__ h_all, int ablate, long long *__restrict__ dbg,
                                                            const int *__restrict__ only_if, unsigned char *gstate, long long gstride) {
    if (only_if && only_if[blockIdx.x] == 0) return;      // the bulk-synchronous kernel already did this task
    const HcMeta M = metas[blockIdx.x];
    const int n = M.n, nld = M.nld, method = M.method;
    double *D = Dall + M.oD;
    int *ia = ia_all + M.oM, *ib = ib_all + M.oM;
    double *crit = h_all + M.oM;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_lds[];
    unsigned char *const sm = GS ? gstate + static_cast<long long>(blockIdx.x) * gstride : sm_lds;
    const int nal = (n + 1) & ~1;
    double *disnn = reinterpret_cast<double *>(sm);
    double *pv = disnn + nal;                 // [32]
    int *nn = reinterpret_cast<int *>(pv + 32);
    int *membr = nn + nal;
    int *list = membr + nal;
    int *pi = list + nal;                     // [32]
    int *cnt = pi + 32;                       // [2]
    unsigned char *flag = reinterpret_cast<unsigned char *>(cnt + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = HC_THREADS / 64;

    if (method == 8) {
        for (long long q = tid; q < static_cast<long long>(n) * nld; q += HC_THREADS) {
            const int r = static_cast<int>(q / nld), c = static_cast<int>(q % nld);
            if (c < n) { const double d = D[static_cast<long long>(r) * nld + c]; D[static_cast<long long>(r) * nld + c] = d * d; }
        }
    }
    for (int i = tid; i < n; i += HC_THREADS) { flag[i] = 1; membr[i] = 1; nn[i] = 0; disnn[i] = HC_INF; }
    __syncthreads();
    // initial NN list: nearest neighbour to the RIGHT of i, lowest j on ties
    for (int i = wave; i < n - 1; i += nwave) {
        const double *row = D + static_cast<long long>(i) * nld;
        MinPair b; b.v = HC_INF; b.i = 0x7fffffff;
        for (int j = i + 1 + lane; j < n; j += 64) { MinPair c; c.v = row[j]; c.i = j; if (c.v < b.v) b = c; }
        b = mp_wave(b);
        if (lane == 0) { nn[i] = b.i; disnn[i] = b.v; }
    }
    __syncthreads();

    // Four workgroup barriers per merge: the two block reductions use alternating scratch so that no
    // "scratch is free again" barrier is needed, the merged pair's bookkeeping is done by the thread that
    // owns index i2 right before it looks at its own entries, and d(i2,j2) is the NN distance just found.
    double *pvB = reinterpret_cast<double *>(flag + ((n + 7) & ~7));
    int *piB = reinterpret_cast<int *>(pvB + 32);
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    for (int step = 0; step < n - 1; ++step) {
        const long long tt0 = dbg ? __builtin_readcyclecounter() : 0;
        // (1) least dissimilarity over the NN list (strict <, lowest index)
        MinPair b; b.v = HC_INF; b.i = 0x7fffffff;
        for (int i = tid; i < n - 1; i += HC_THREADS)
            if (flag[i]) { MinPair c; c.v = disnn[i]; c.i = i; if (c.v < b.v) b = c; }
        b = mp_block(b, pv, pi);
        const long long tt1 = dbg ? __builtin_readcyclecounter() : 0;
        const int i2 = b.i < n ? b.i : 0;       // NN lists look to the right, so im < nn[im]
        const int j2 = nn[i2];
        const double d12 = b.v;                  // DISNN(im) == D(im, NN(im)) is an invariant of the algorithm
        const double mi = membr[i2], mj = membr[j2];
        if (tid == 0) {
            ia[step] = i2 + 1; ib[step] = j2 + 1;
            crit[step] = method == 8 ? sqrt(b.v) : b.v;
            *cnt = 0;
        }
        // (2) Lance-Williams update of row/column i2; new NN of i2 among k > i2
        MinPair nb; nb.v = HC_INF; nb.i = 0x7fffffff;
        const double *ri = D + static_cast<long long>(i2) * nld, *rj = D + static_cast<long long>(j2) * nld;
        // loads are issued unconditionally and in batches: a load under a data-dependent branch would make
        // every iteration a separate dependent HBM round trip
        for (int k0 = tid; k0 < n; k0 += 4 * HC_THREADS) {
            double a1[4], a2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * HC_THREADS;
                const int kk = k < n ? k : n - 1;
                a1[u] = ri[kk]; a2[u] = rj[kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * HC_THREADS;
                if (k < n && k != i2 && k != j2 && flag[k]) {
                    const double dn = lance_williams(method, a1[u], a2[u], d12, mi, mj, static_cast<double>(membr[k]));
                    D[static_cast<long long>(i2) * nld + k] = dn;
                    if (!(ablate & 1)) D[static_cast<long long>(k) * nld + i2] = dn;
                    if (i2 < k) { if (dn < nb.v) { nb.v = dn; nb.i = k; } }
                    else if (dn < disnn[k]) { disnn[k] = dn; nn[k] = i2; }
                }
            }
        }
        const long long tt2 = dbg ? __builtin_readcyclecounter() : 0;
        nb = mp_block(nb, pvB, piB);
        const long long tt3 = dbg ? __builtin_readcyclecounter() : 0;
        if (tid == (i2 % HC_THREADS)) {          // owner of i2: merge bookkeeping before it scans its own entries
            membr[i2] = membr[i2] + membr[j2];
            disnn[i2] = nb.v;
            if (nb.i < n) nn[i2] = nb.i;
        }
        if (tid == (j2 % HC_THREADS)) flag[j2] = 0;
        // (3) rows whose nearest neighbour was i2 or j2 look again to their right
        for (int i = tid; i < n - 1; i += HC_THREADS)
            if (i != j2 && flag[i] && (nn[i] == i2 || nn[i] == j2)) list[atomicAdd(cnt, 1)] = i;
        __syncthreads();
        const long long tt4 = dbg ? __builtin_readcyclecounter() : 0;
        const int nl = (ablate & 2) ? 0 : *cnt;
        for (int q = wave; q < nl; q += nwave) {
            const int i = list[q];
            const double *row = D + static_cast<long long>(i) * nld;
            MinPair c; c.v = HC_INF; c.i = 0x7fffffff;
            for (int j0 = i + 1 + lane; j0 < n; j0 += 64 * HC_RS) {   // one pass (one HBM round trip) covers 2048 entries
                double v[HC_RS];
#pragma unroll
                for (int u = 0; u < HC_RS; ++u) { const int j = j0 + 64 * u; v[u] = row[j < n ? j : n - 1]; }
#pragma unroll
                for (int u = 0; u < HC_RS; ++u) {
                    const int j = j0 + 64 * u;
                    if (j < n && flag[j] && v[u] < c.v) { c.v = v[u]; c.i = j; }
                }
            }
            c = mp_wave(c);
            if (lane == 0) { disnn[i] = c.v; if (c.i < n) nn[i] = c.i; }
        }
        const long long tt5 = dbg ? __builtin_readcyclecounter() : 0;
        __syncthreads();
        if (dbg) { const long long tt6 = __builtin_readcyclecounter(); tacc[0] += tt1 - tt0; tacc[1] += tt2 - tt1; tacc[2] += tt3 - tt2; tacc[3] += tt4 - tt3; tacc[4] += tt5 - tt4; tacc[5] += tt6 - tt5; }
    }
    if (dbg && tid == 0) for (int q = 0; q < 6; ++q) dbg[blockIdx.x * 6 + q] = tacc[q];
}

// ---------------------------------------------------------------------------------------------
// a4, bulk-synchronous form for the reducible methods (ward.D, ward.D2, single, complete, average, mcquitty).
// For a reducible Lance-Williams update, merging a reciprocal-nearest-neighbour (RNN) pair never brings anything closer to
// any other cluster than that cluster's current nearest neighbour, so every RNN pair of the current matrix is a merge of
// the sequential algorithm, at the same height.  A round therefore (1) pairs up all RNN pairs, (2) ranks them by
// (height, lowest index) -- the order in which the sequential algorithm would perform them -- and (3) writes the next
// distance matrix compacted to the survivors, one wave per new row, reading whole old rows and writing whole new rows:
// pure streaming instead of one scattered 8-byte column write per (merge, cluster).  The nearest neighbour of every new
// row falls out of the same pass.  Entries between two clusters merged in the same round apply the two updates in rank
// order, exactly the arithmetic of the sequential algorithm; across rounds the association order can differ from the
// sequential one, so heights agree to rounding (1e-15), not bit for bit.  The merges are sorted by (height, index) at the
// end.  Any exact tie for a row minimum (or a round without a pair) abandons the task: status = 1, and the host runs
// hclust_kernel on it (R breaks ties by index order inside its nearest-neighbour lists; that is only restated there).
// D is left untouched; the rounds ping-pong between two scratch matrices.
// ---------------------------------------------------------------------------------------------
constexpr int HR_MAXN = 4096;
constexpr size_t HR_LDS_CU = 160 * 1024;     // LDS of a gfx950 compute unit
constexpr uint16_t HR_NONE = 0xffffu;

struct HrBest { double v; int i; int tie; };
__device__ __forceinline__ HrBest hr_combine(HrBest x, HrBest y) {
    HrBest r;
    if (y.v < x.v) r = y; else if (x.v < y.v) r = x;
    else { r.v = x.v; r.i = x.i < y.i ? x.i : y.i; r.tie = (x.i != y.i && x.i < 0x7fffffff && y.i < 0x7fffffff) ? 1 : (x.tie | y.tie); }
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ HrBest hr_dpp_step(HrBest x) {
    const int lo = __double2loint(x.v), hi = __double2hiint(x.v);
    HrBest y;
    y.v = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false),
                           __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false));
    y.i = __builtin_amdgcn_update_dpp(x.i, x.i, CTRL, ROW_MASK, 0xf, false);
    y.tie = __builtin_amdgcn_update_dpp(x.tie, x.tie, CTRL, ROW_MASK, 0xf, false);
    return hr_combine(x, y);
}
__device__ __forceinline__ HrBest hr_wave(HrBest x) {
    x = hr_dpp_step<0xB1, 0xf>(x);
    x = hr_dpp_step<0x4E, 0xf>(x);
    x = hr_dpp_step<0x141, 0xf>(x);
    x = hr_dpp_step<0x140, 0xf>(x);
    x = hr_dpp_step<0x142, 0xa>(x);
    x = hr_dpp_step<0x143, 0xc>(x);
    HrBest r;
    r.i = __builtin_amdgcn_readlane(x.i, 63);
    r.tie = __builtin_amdgcn_readlane(x.tie, 63);
    r.v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x.v), 63), __builtin_amdgcn_readlane(__double2loint(x.v), 63));
    return r;
}

// the same over the 16 lanes of a DPP row: every lane of the row ends with the row's result
__device__ __forceinline__ HrBest hr_row16(HrBest x) {
    x = hr_dpp_step<0xB1, 0xf>(x);
    x = hr_dpp_step<0x4E, 0xf>(x);
    x = hr_dpp_step<0x141, 0xf>(x);
    x = hr_dpp_step<0x140, 0xf>(x);
    return x;
}
template <int CTRL>
__device__ __forceinline__ double hr_min_step(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return fmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ double hr_min16(double x) {
    x = hr_min_step<0xB1>(x);
    x = hr_min_step<0x4E>(x);
    x = hr_min_step<0x141>(x);
    x = hr_min_step<0x140>(x);
    return x;
}

// HR_THREADS = 512: two tasks per CU (LDS state 37 B per observation, <= 128 VGPRs) when there are more tasks than CUs;
// 1024: one task per CU with twice the loads in flight when there are not (a task streams ~300 MB through ONE workgroup).
// MODE 0: the whole agglomeration in one launch, one workgroup per task.
// MODE 1 / 2: one ROUND per pair of launches, so that a task is no longer confined to the ~34 GB/s one CU can move: the
// LDS state lives as an image in global memory between launches; MODE 1 (one workgroup per task) loads it, applies the
// previous round's transition, finds and ranks the reciprocal pairs, builds the column maps and stores it back; MODE 2
// (gridDim.y workgroups per task) loads it read-only and rebuilds its share of the rows (work is handed out by counters in
// the image), writing the new rows' nearest neighbours straight into the image.
// MODE 3: picks a task up from its image and runs ALL its remaining rounds in this one launch (one workgroup per task, like MODE 0):
// once a few hundred clusters are left a round's two launches cost more than its work -- the last ~33 of the 45 rounds of a
// 2000-observation task took 3.3 ms as 66 launches.
// GS (MODE 1 / 2 / 3 only): tasks beyond HR_MAXN observations, whose state does not fit a CU's LDS -- the state arrays ARE the global
// image (no copy in or out; the same code addresses them), only the stage of the rebuild stays in LDS.
typedef __attribute__((address_space(1))) const double *hr_gcd;   // the distance matrices, in the global address space
typedef __attribute__((address_space(1))) double *hr_gd;
template <int HR_THREADS, int MODE, bool GS = false>
__global__ __launch_bounds__(HR_THREADS) void hclust_rnn_kernel(const HcMeta *__restrict__ metas, const double *__restrict__ Dall,
                                                                double *__restrict__ S0all, double *__restrict__ S1all,
                                                                int *__restrict__ ia_all, int *__restrict__ ib_all,
                                                                double *__restrict__ h_all, int *__restrict__ status,
                                                                unsigned char *__restrict__ images, long long image_stride,
                                                                int lds_bytes, int round, int *__restrict__ remaining, int lds_launch) {
    const HcMeta M = metas[blockIdx.x];
    const int n = M.n, nld = M.nld, method = M.method;
    // (global address space spelled out: left generic, every access of the matrices compiled to a FLAT instruction, which takes an LDS issue slot
    // as well and counts on lgkmcnt -- each wait for the column map in LDS then also waited for the 16 row loads in flight)
    const hr_gcd D = (hr_gcd)(Dall + M.oD);
    const hr_gd Sb[2] = {(hr_gd)(S0all + M.oD), (hr_gd)(S1all + M.oD)};
    int *ia = ia_all + M.oM, *ib = ib_all + M.oM;
    double *crit = h_all + M.oM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = HR_THREADS / 64;
    unsigned char *img = MODE ? images + static_cast<long long>(blockIdx.x) * image_stride : nullptr;
    if (method == 6 || method == 7 || n > (GS ? kHcMaxN : HR_MAXN)) {   // centroid / median are not reducible; large n: LDS
        if (MODE != 2 && (MODE == 0 || round == 0) && tid == 0) { status[blockIdx.x] = 1; if (MODE == 1) atomicSub(remaining, 1); }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_lds[];
    unsigned char *const sm = GS ? img : sm_lds;               // where the state arrays live
    const int nal = (n + 3) & ~3;
    double *dnnA = reinterpret_cast<double *>(sm);             // [2][nal]  NN distance (also the pair's height)
    uint16_t *cidA = reinterpret_cast<uint16_t *>(dnnA + 2 * nal);   // [2][nal]  smallest original member
    uint16_t *cszA = cidA + 2 * nal;                           // [2][nal]  cluster size
    uint16_t *nn = cszA + 2 * nal;                             // [nal]
    uint16_t *partner = nn + nal;                              // [nal]     old index of the RNN partner or NONE
    uint16_t *pseq = partner + nal;                            // [nal]     rank of the pair in the round
    uint16_t *oldidx = pseq + nal;                             // [nal]     new index -> old index
    uint16_t *newidx = oldidx + nal;                           // [nal]     old index -> new index (survivors)
    uint16_t *plist = newidx + nal;                            // [nal]     first members of the pairs
    uint16_t *colmap = plist + nal;                            // [nal]     old column -> new column, or 0x8000 | (2 rank + member) for the two members of a pair
    int *ctl = reinterpret_cast<int *>(colmap + nal);            // [16]: 0 npairs, 1 abort, 2/3 work counters (plain / merged rows), 4 nsingle,
                                                                //       5 cur, 6 na, 7 done, 8 src + 1, 9 nb, 10 state, 11 pending  (5..11: MODE 1/2)
    int *wsum = ctl + 16;                                       // [nwave + 1]
    unsigned char *tie = reinterpret_cast<unsigned char *>(wsum + nwave + 1);   // [nal]
    // what is left of the workgroup's LDS stages the pair members' entries of the rows being copied (see the rebuild below)
    const int stage_off = GS ? 0 : static_cast<int>((tie + nal - sm + 15) & ~static_cast<long>(15));
    double *stage = reinterpret_cast<double *>(sm_lds + stage_off);
    const int stage_pairs = lds_launch > stage_off ? (lds_launch - stage_off) / (nwave * 32) : 0;   // 2 rows x 2 members x 8 B per pair and wave

    int cur = 0, na = n, done = 0;
    int src = -1;                                               // -1: D (pristine), else scratch index
    const bool fresh = MODE == 0 || (MODE == 1 && round == 0);
    if (!fresh) {                                               // the state image of the previous launches
        if (!GS) {
            const uint4 *gi = reinterpret_cast<const uint4 *>(img);
            uint4 *li = reinterpret_cast<uint4 *>(sm);
            for (int q = tid; q < lds_bytes / 16; q += HR_THREADS) li[q] = gi[q];
            __syncthreads();
        }
        if (ctl[10] != 0 || (MODE == 2 && !ctl[11])) return;   // finished / abandoned, or nothing pending
        cur = ctl[5]; na = ctl[6]; done = ctl[7]; src = ctl[8] - 1;
        if ((MODE == 1 || MODE == 3) && ctl[11]) {              // apply the transition of the round that MODE 2 just rebuilt
            done += ctl[0]; na = ctl[9]; cur ^= 1; src = src < 0 ? 0 : (src ^ 1);
            __syncthreads();
            if (tid == 0) { ctl[0] = 0; ctl[11] = 0; }
            __syncthreads();
        }
    }
    if (fresh) {
    for (int i = tid; i < n; i += HR_THREADS) { cidA[i] = static_cast<uint16_t>(i); cszA[i] = 1; }
    if (tid == 0) { for (int q = 0; q < 16; ++q) ctl[q] = 0; }
    __syncthreads();
    // round 0 nearest neighbours.  With the row minima the distance GEMM left per 128-column tile (HcMeta::nn, already squared for ward.D2): a row's
    // minimum is the smallest of its tiles' minima, and only the tile(s) that hold it are scanned for the lowest column and a second one (tie) --
    // 1 KB per row instead of 16 KB.  Sixteen lanes per row, four rows per wave.
    if (M.nn) {
        const int slots = nld / 128;                            // <= 16 (setup_chunk)
        const int g = lane >> 4, l = lane & 15;
        for (int a0 = wave * 4; a0 < n; a0 += nwave * 4) {
            const int a = a0 + g < n ? a0 + g : n - 1;          // (a group beyond the last row repeats it and stores nothing)
            const double pm = l < slots ? M.nn[static_cast<long long>(l) * nld + a] : HC_INF;
            const double m = hr_min16(pm);
            unsigned cand = static_cast<unsigned>(__ballot(l < slots && pm == m) >> (16 * g)) & 0xffffu;   // this row's tiles that hold its minimum
            const hr_gcd row = D + static_cast<long long>(a) * nld;
            HrBest b; b.v = HC_INF; b.i = 0x7fffffff; b.tie = 0;
            while (__any(cand != 0u)) {
                if (cand) {
                    const int j0 = (__ffs(cand) - 1) * 128 + l;
                    cand &= cand - 1;
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int j = j0 + 16 * u; v[u] = row[j < n ? j : n - 1]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + 16 * u;
                        if (j < n && j != a) {
                            const double x = method == 8 ? v[u] * v[u] : v[u];
                            if (x < b.v) { b.v = x; b.i = j; b.tie = 0; } else if (x == b.v) b.tie = 1;
                        }
                    }
                }
            }
            b = hr_row16(b);
            if (l == 0 && a0 + g < n) { nn[a] = static_cast<uint16_t>(b.i < n ? b.i : 0); dnnA[a] = b.v; tie[a] = static_cast<unsigned char>(b.tie); }
        }
    } else
    // ... or from the pristine matrix (squared for ward.D2)
    for (int a = wave; a < n; a += nwave) {
        const hr_gcd row = D + static_cast<long long>(a) * nld;
        HrBest b; b.v = HC_INF; b.i = 0x7fffffff; b.tie = 0;
        for (int j0 = lane; j0 < n; j0 += 64 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; v[u] = row[j < n ? j : n - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 64 * u;
                if (j < n && j != a) {
                    const double x = method == 8 ? v[u] * v[u] : v[u];
                    if (x < b.v) { b.v = x; b.i = j; b.tie = 0; } else if (x == b.v) b.tie = 1;
                }
            }
        }
        b = hr_wave(b);
        if (lane == 0) { nn[a] = static_cast<uint16_t>(b.i < n ? b.i : 0); dnnA[a] = b.v; tie[a] = static_cast<unsigned char>(b.tie); }
    }
    __syncthreads();
    }   // fresh

    // where the results of a rebuilt row go: the LDS arrays (MODE 0) or the global image (MODE 2)
    auto outp = [&](auto *lds_ptr) { return (MODE == 2 && !GS) ? reinterpret_cast<decltype(lds_ptr)>(img + (reinterpret_cast<unsigned char *>(lds_ptr) - sm)) : lds_ptr; };
    int *wctl = outp(ctl);
    auto store_image = [&]() {
        __syncthreads();
        if (GS) return;
        uint4 *gi = reinterpret_cast<uint4 *>(img);
        const uint4 *li = reinterpret_cast<const uint4 *>(sm);
        for (int q = tid; q < lds_bytes / 16; q += HR_THREADS) gi[q] = li[q];
    };
#ifdef HR_TIMING
    long long hr_acc_setup = 0, hr_acc_rebuild = 0, hr_acc_barrier = 0, hr_acc_entries = 0, hr_acc_wave_busy = 0;
    int hr_rounds = 0;
    const long long hr_start = __builtin_readcyclecounter();
#endif
    while (na > 1) {
        double *dnn = dnnA + cur * nal;
        uint16_t *cid = cidA + cur * nal, *csz = cszA + cur * nal;
        double *dnnN = outp(dnnA + (cur ^ 1) * nal);
        uint16_t *cidN = outp(cidA + (cur ^ 1) * nal), *cszN = outp(cszA + (cur ^ 1) * nal);
        uint16_t *nnW = outp(nn);
        unsigned char *tieW = outp(tie);
        int np = 0, nb = 0, ns = 0;
#ifdef HR_TIMING
        const long long hr_t0 = __builtin_readcyclecounter();
#endif
        if (MODE != 2) {
        // (1) reciprocal pairs
        for (int a = tid; a < na; a += HR_THREADS) {
            partner[a] = HR_NONE;
            if (tie[a]) ctl[1] = 1;
#ifdef HR_ROUNDS
            if (tie[a] && blockIdx.x == 0) printf("tie at row %d of %d (done %d): nn %d dnn %.17g\n", a, na, done, (int)nn[a], dnn[a]);
#endif
        }
        __syncthreads();
        for (int a = tid; a < na; a += HR_THREADS) {
            const int b = nn[a];
            if (b > a && nn[b] == a) {
                partner[a] = static_cast<uint16_t>(b); partner[b] = static_cast<uint16_t>(a);
                plist[atomicAdd(&ctl[0], 1)] = static_cast<uint16_t>(a);
            }
        }
        __syncthreads();
        np = ctl[0];
        if (ctl[1] || np == 0) {                                // tie or no pair: the sequential kernel takes this task
            if (tid == 0) {
                status[blockIdx.x] = 1;
                if (MODE == 1) { reinterpret_cast<int *>(img + (reinterpret_cast<unsigned char *>(ctl) - sm))[10] = 1; atomicSub(remaining, 1); }
            }
            return;
        }
        // (2) rank of each pair by (height, lower original index) = the sequential algorithm's order
        for (int q = tid; q < np; q += HR_THREADS) {
            const int a = plist[q];
            const double h = dnn[a];
            const int ida = cid[a] < cid[partner[a]] ? cid[a] : cid[partner[a]];
            int rank = 0;
            for (int q2 = 0; q2 < np; ++q2) {
                const int a2 = plist[q2];
                const double h2 = dnn[a2];
                const int id2 = cid[a2] < cid[partner[a2]] ? cid[a2] : cid[partner[a2]];
                rank += (h2 < h || (h2 == h && id2 < ida)) ? 1 : 0;
            }
            pseq[a] = static_cast<uint16_t>(rank); pseq[partner[a]] = static_cast<uint16_t>(rank);
            const int ib_ = cid[a] < cid[partner[a]] ? cid[partner[a]] : cid[a];
            ia[done + rank] = ida + 1; ib[done + rank] = ib_ + 1;
            crit[done + rank] = h;                               // squared for ward.D2 until the final pass
        }
        // (3) new indices: the unmerged clusters keep their relative order in [0, ns), the merged clusters follow in rank order in
        // [ns, nb).  Every new row is then written as two dense runs of stores.  (With the merged clusters left in place, each
        // plain row was stored with a hole per merged column, filled later by a scattered 8-byte store: partial-line writes that
        // cost the HBM 57 % more reads and 34 % more writes than the algorithm needs -- FETCH_SIZE / WRITE_SIZE)
        // Exact ties abandon the task, so the order of the columns decides nothing.
        {
            const int chunk = (na + HR_THREADS - 1) / HR_THREADS;
            const int lo = tid * chunk, hi = lo + chunk < na ? lo + chunk : na;
            int c = 0;
            for (int a = lo; a < hi; ++a) c += partner[a] == HR_NONE ? 1 : 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); incl += lane >= d ? t : 0; }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            if (tid == 0) { int run = 0; for (int w = 0; w < nwave; ++w) { const int t = wsum[w]; wsum[w] = run; run += t; } wsum[nwave] = run; }
            __syncthreads();
            int pos = wsum[wave] + incl - c;
            for (int a = lo; a < hi; ++a)
                if (partner[a] == HR_NONE) { newidx[a] = static_cast<uint16_t>(pos); colmap[a] = static_cast<uint16_t>(pos); oldidx[pos++] = static_cast<uint16_t>(a); }
            ns = wsum[nwave];                                   // rows of unmerged clusters (= na - 2 np)
            for (int q = tid; q < np; q += HR_THREADS) {        // bit 15 of oldidx: the survivor is a merged cluster
                const int a = plist[q], B = ns + pseq[a];
                newidx[a] = static_cast<uint16_t>(B);
                oldidx[B] = static_cast<uint16_t>(a | 0x8000);
                colmap[a] = static_cast<uint16_t>(0x8000 | (2 * pseq[a]));
                colmap[partner[a]] = static_cast<uint16_t>(0x8000 | (2 * pseq[a] + 1));
            }
        }
        nb = ns + np;
        if (tid == 0) { ctl[2] = 0; ctl[3] = 0; ctl[4] = ns; }
        __syncthreads();
        }   // MODE != 2
        if (MODE == 1) {                                        // hand the round over to the rebuild launch
            __syncthreads();
            if (tid == 0) { ctl[5] = cur; ctl[6] = na; ctl[7] = done; ctl[8] = src + 1; ctl[9] = nb; ctl[10] = 0; ctl[11] = 1; }
            store_image();
            return;
        }
        if (MODE == 2) { np = ctl[0]; nb = ctl[9]; ns = ctl[4]; }
#ifdef HR_TIMING
        const long long hr_t1 = __builtin_readcyclecounter();
        long long hr_dual = 0, hr_slow = 0;
#endif
        // (4) next matrix, one wave per new row; nearest neighbour of the new row on the fly.
        // Rows of unmerged clusters (~90 %) are a gathered copy of the old row (eight loads in flight per lane) plus one
        // Lance-Williams value per merged column; rows of merged clusters take the general path.
        const hr_gcd Dsrc = src < 0 ? D : (hr_gcd)Sb[src];
        const hr_gd Ddst = Sb[src < 0 ? 0 : (src ^ 1)];
        const bool sq = (src < 0 && method == 8);
        auto do_row = [&](int A) {
            const int a = oldidx[A] & 0x7fff;
            const int pa = partner[a];                          // NONE or j > a
            const bool am = pa != HR_NONE;
            const hr_gcd ra = Dsrc + static_cast<long long>(a) * nld;
            const hr_gd wr = Ddst + static_cast<long long>(A) * nld;
            const double na_ = csz[a];
            HrBest best; best.v = HC_INF; best.i = 0x7fffffff; best.tie = 0;
            auto consider = [&](double v, int B) {
                if (v < best.v || (v == best.v && B < best.i)) { best.tie = (v == best.v) ? 1 : 0; best.v = v; best.i = B; }
                else if (v == best.v && B != best.i) best.tie = 1;
            };
            if (!am) {
                for (int B0 = lane; B0 < ns; B0 += 64 * 8) {   // columns of unmerged clusters
                    int bb[8];
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int B = B0 + 64 * u;
                        bb[u] = oldidx[B < ns ? B : ns - 1];
                        x[u] = ra[bb[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int B = B0 + 64 * u;
                        if (B < ns) {
                            const double v = B == A ? HC_INF : (sq ? x[u] * x[u] : x[u]);   // scratch diagonals hold +inf: no test in later rounds
                            wr[B] = v;
                            if (B != A) consider(v, B);
                        }
                    }
                }
                for (int B = ns + lane; B < nb; B += 64) {      // merged columns: d(a, k u l) from d(a,k), d(a,l)
                    const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
                    double d1 = ra[k1], d2 = ra[l1];
                    if (sq) { d1 *= d1; d2 *= d2; }
                    const double v = lance_williams(method, d1, d2, dnn[k1], static_cast<double>(csz[k1]), static_cast<double>(csz[l1]), na_);
                    wr[B] = v;
                    consider(v, B);
                }
            } else {
                const hr_gcd rj = Dsrc + static_cast<long long>(pa) * nld;
                const double hP = dnn[a];
                const double nj_ = csz[pa];
                const int seqP = pseq[a];
                for (int B0 = lane; B0 < nb; B0 += 64 * 4) {
                    int bb[4], pbv[4];
                    double x00[4], x01[4], x10[4], x11[4];      // D[a][b], D[a][pb], D[j][b], D[j][pb]
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int B = B0 + 64 * u;
                        bb[u] = oldidx[B < nb ? B : nb - 1] & 0x7fff;
                        const int pb = partner[bb[u]];
                        pbv[u] = pb;
                        const int pbc = pb == HR_NONE ? bb[u] : pb;
                        x00[u] = ra[bb[u]]; x01[u] = ra[pbc]; x10[u] = rj[bb[u]]; x11[u] = rj[pbc];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int B = B0 + 64 * u;
                        if (B < nb) {
                            const int b = bb[u];
                            const bool bm = pbv[u] != HR_NONE;
                            double d00 = x00[u], d01 = x01[u], d10 = x10[u], d11 = x11[u];
                            if (sq) { d00 *= d00; d01 *= d01; d10 *= d10; d11 *= d11; }
                            double v;
                            if (B == A) v = HC_INF;
                            else if (!bm) v = lance_williams(method, d00, d10, hP, na_, nj_, static_cast<double>(csz[b]));
                            else {
                                const double nk_ = csz[b], nl_ = csz[pbv[u]], hQ = dnn[b];
                                if (seqP < static_cast<int>(pseq[b])) {   // (a, j) merges first, then (b, l) against the merged cluster
                                    const double t1 = lance_williams(method, d00, d10, hP, na_, nj_, nk_);
                                    const double t2 = lance_williams(method, d01, d11, hP, na_, nj_, nl_);
                                    v = lance_williams(method, t1, t2, hQ, nk_, nl_, na_ + nj_);
                                } else {
                                    const double t1 = lance_williams(method, d00, d01, hQ, nk_, nl_, na_);
                                    const double t2 = lance_williams(method, d10, d11, hQ, nk_, nl_, nj_);
                                    v = lance_williams(method, t1, t2, hP, na_, nj_, nk_ + nl_);
                                }
                            }
                            wr[B] = v;
                            if (B != A) consider(v, B);
                        }
                    }
                }
            }
            best = hr_wave(best);
            if (lane == 0) {
                // the merged cluster keeps the smaller original index as its name (R: i2 < j2)
                cidN[A] = am ? (cid[a] < cid[pa] ? cid[a] : cid[pa]) : cid[a];
                cszN[A] = static_cast<uint16_t>(csz[a] + (am ? csz[pa] : 0));
                dnnN[A] = best.v;
            }
            // nn / tie of the new round live in the single-buffered arrays: nothing reads the old ones in this phase
            if (lane == 1) { nnW[A] = static_cast<uint16_t>(best.i < nb ? best.i : 0); }
            if (lane == 2) { tieW[A] = static_cast<unsigned char>(nb > 2 ? best.tie : 0); }
        };
        // two unmerged rows at a time share the column map (one set of LDS reads) and keep 16 loads in flight per lane
        auto finish_row = [&](int A, int a, HrBest best) {
            best = hr_wave(best);
            if (lane == 0) { cidN[A] = cid[a]; cszN[A] = csz[a]; dnnN[A] = best.v; }
            if (lane == 1) { nnW[A] = static_cast<uint16_t>(best.i < nb ? best.i : 0); }
            if (lane == 2) { tieW[A] = static_cast<unsigned char>(nb > 2 ? best.tie : 0); }
        };
#ifdef HR_NO_FIRST_STAGE
        const int stage_rows = src < 0 ? 0 : (np == 0 ? 2 : std::min(2, 2 * stage_pairs / np));
#else
        const int stage_rows = np == 0 ? 2 : std::min(2, 2 * stage_pairs / np);   // rows per wave whose pair entries fit the stage
#endif
        // Plain rows, staged form.  The old rows are read ONCE, contiguously (the gathered forms further down skip the pair members'
        // entries and come back for them after the sweep, by which time the lines have left the L2: the L2's request-size counters
        // showed every row fetched twice, 12.1 n^2 entries per task instead of 6.05).  An entry of an unmerged column goes straight
        // to its new column (a dense run of stores per instruction); an entry of a pair member is parked in this wave's LDS stage,
        // from where the Lance-Williams loop takes it.  NR = 2 rows per wave when the stage holds the round's pairs twice, else 1.
        // FIRST: the source is the pristine matrix (real diagonal; squared on the fly for ward.D2).
        auto staged = [&](auto NR_, auto FIRST_, int A0) {
            constexpr int NR = decltype(NR_)::value;
            constexpr bool FIRST = decltype(FIRST_)::value;
            hr_gcd r[NR];
            hr_gd w[NR];
            double *stg[NR];
            double mn[NR], sc[NR];
            int ix[NR], ao[NR];
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                ao[t] = __builtin_amdgcn_readfirstlane(oldidx[A0 + t] & 0x7fff);
                r[t] = Dsrc + static_cast<long long>(ao[t]) * nld;
                w[t] = Ddst + static_cast<long long>(A0 + t) * nld;
                stg[t] = stage + (static_cast<size_t>(wave) * stage_rows + t) * 2 * np;   // a wave's region does not depend on NR (odd last row)
                mn[t] = HC_INF; sc[t] = HC_INF; ix[t] = 0x7fffffff;
            }
            auto upd = [](double &m_, double &s_, int &i_, double v, int B) {
                s_ = fmin(s_, fmax(m_, v));
                if (v < m_) { m_ = v; i_ = B; }
            };
            int j0 = lane;
            auto pass = [&](auto U_) {
                constexpr int U = decltype(U_)::value;
                for (; j0 + 64 * (U - 1) < na; j0 += 64 * U) {
                    unsigned cm[U];
                    double x[NR][U];
#pragma unroll
                    for (int u = 0; u < U; ++u) cm[u] = colmap[j0 + 64 * u];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int t = 0; t < NR; ++t) x[t][u] = r[t][j0 + 64 * u];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (cm[u] & 0x8000u) {
#pragma unroll
                            for (int t = 0; t < NR; ++t) stg[t][cm[u] & 0x7fffu] = (FIRST && sq) ? x[t][u] * x[t][u] : x[t][u];
                        } else {
                            const int B = static_cast<int>(cm[u]);
#pragma unroll
                            for (int t = 0; t < NR; ++t) {
                                double v = x[t][u];
                                if (FIRST) v = B == A0 + t ? HC_INF : (sq ? v * v : v);   // scratch diagonals hold +inf: no test in later rounds
                                w[t][B] = v;
                                upd(mn[t], sc[t], ix[t], v, B);
                            }
                        }
                    }
                    if (U == 1) break;
                }
            };
            if (NR == 1) pass(std::integral_constant<int, 16>());   // 16 entries per lane in flight either way (128 VGPRs, no scratch)
            pass(std::integral_constant<int, 8>());
            pass(std::integral_constant<int, 4>());
            pass(std::integral_constant<int, 2>());
            pass(std::integral_constant<int, 1>());
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double nr_[NR];
#pragma unroll
            for (int t = 0; t < NR; ++t) nr_[t] = csz[ao[t]];
            for (int B = ns + lane; B < nb; B += 64) {          // merged columns: d(a, k u l) from d(a,k), d(a,l)
                const int rk = B - ns;
                const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
                const double nk_ = csz[k1], nl_ = csz[l1], hQ = dnn[k1];
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const double v = lance_williams(method, stg[t][2 * rk], stg[t][2 * rk + 1], hQ, nk_, nl_, nr_[t]);
                    w[t][B] = v;
                    sc[t] = fmin(sc[t], fmax(mn[t], v));
                    if (v < mn[t] || (v == mn[t] && B < ix[t])) { mn[t] = v; ix[t] = B; }
                }
            }
            __builtin_amdgcn_wave_barrier();                     // the stage is reused by this wave's next rows
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                HrBest g;
                g.v = mn[t]; g.i = ix[t]; g.tie = 0;
                g = hr_wave(g);
                g.tie |= __ballot(sc[t] == g.v) != 0ull ? 1 : 0; // a lane saw the minimum twice
                if (lane == 0) { cidN[A0 + t] = cid[ao[t]]; cszN[A0 + t] = csz[ao[t]]; dnnN[A0 + t] = g.v; }
                if (lane == 1) { nnW[A0 + t] = static_cast<uint16_t>(g.i < nb ? g.i : 0); }
                if (lane == 2) { tieW[A0 + t] = static_cast<unsigned char>(nb > 2 ? g.tie : 0); }
            }
        };
        // Rows of merged clusters, staged form (needs the two-row stage): the two old rows a and j of the pair are read once,
        // contiguously; an unmerged column gets its Lance-Williams value at once, the four entries of a merged column (a, j) x (k, l)
        // wait in the stage.  (The gathered form in do_row fetches d(a,l), d(j,l) from wherever column l lies: one more 128-byte
        // line per entry, 2.6 x the row's own bytes.)
        auto staged_merged = [&](int A) {
            const int a = __builtin_amdgcn_readfirstlane(oldidx[A] & 0x7fff);
            const int pa = __builtin_amdgcn_readfirstlane(partner[a]);
            const hr_gcd ra = Dsrc + static_cast<long long>(a) * nld, rj = Dsrc + static_cast<long long>(pa) * nld;
            const hr_gd wr = Ddst + static_cast<long long>(A) * nld;
            const double na_ = csz[a], nj_ = csz[pa], hP = dnn[a];
            const int seqP = pseq[a];
            double *sa = stage + static_cast<size_t>(wave) * 4 * np, *sj = sa + 2 * np;
            double mn = HC_INF, sc = HC_INF;
            int ix = 0x7fffffff;
            int j0 = lane;
            auto pass = [&](auto U_) {
                constexpr int U = decltype(U_)::value;
                for (; j0 + 64 * (U - 1) < na; j0 += 64 * U) {
                    unsigned cm[U];
                    double xa[U], xj[U], nc[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { cm[u] = colmap[j0 + 64 * u]; nc[u] = csz[j0 + 64 * u]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) { xa[u] = ra[j0 + 64 * u]; xj[u] = rj[j0 + 64 * u]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (sq) { xa[u] *= xa[u]; xj[u] *= xj[u]; }
                        if (cm[u] & 0x8000u) { sa[cm[u] & 0x7fffu] = xa[u]; sj[cm[u] & 0x7fffu] = xj[u]; }
                        else {
                            const int B = static_cast<int>(cm[u]);
                            const double v = lance_williams(method, xa[u], xj[u], hP, na_, nj_, nc[u]);
                            wr[B] = v;
                            sc = fmin(sc, fmax(mn, v));
                            if (v < mn) { mn = v; ix = B; }
                        }
                    }
                    if (U == 1) break;
                }
            };
            pass(std::integral_constant<int, 8>());
            pass(std::integral_constant<int, 4>());
            pass(std::integral_constant<int, 2>());
            pass(std::integral_constant<int, 1>());
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int B = ns + lane; B < nb; B += 64) {
                const int rk = B - ns;
                double v = HC_INF;                              // own column: the scratch diagonal
                if (B != A) {
                    const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
                    const double d00 = sa[2 * rk], d01 = sa[2 * rk + 1], d10 = sj[2 * rk], d11 = sj[2 * rk + 1];
                    const double nk_ = csz[k1], nl_ = csz[l1], hQ = dnn[k1];
                    if (seqP < static_cast<int>(pseq[k1])) {    // (a, j) merges first, then (k, l) against the merged cluster
                        const double t1 = lance_williams(method, d00, d10, hP, na_, nj_, nk_);
                        const double t2 = lance_williams(method, d01, d11, hP, na_, nj_, nl_);
                        v = lance_williams(method, t1, t2, hQ, nk_, nl_, na_ + nj_);
                    } else {
                        const double t1 = lance_williams(method, d00, d01, hQ, nk_, nl_, na_);
                        const double t2 = lance_williams(method, d10, d11, hQ, nk_, nl_, nj_);
                        v = lance_williams(method, t1, t2, hP, na_, nj_, nk_ + nl_);
                    }
                    sc = fmin(sc, fmax(mn, v));
                    if (v < mn || (v == mn && B < ix)) { mn = v; ix = B; }
                }
                wr[B] = v;
            }
            __builtin_amdgcn_wave_barrier();
            HrBest g;
            g.v = mn; g.i = ix; g.tie = 0;
            g = hr_wave(g);
            g.tie |= __ballot(sc == g.v) != 0ull ? 1 : 0;
            if (lane == 0) { cidN[A] = cid[a] < cid[pa] ? cid[a] : cid[pa]; cszN[A] = static_cast<uint16_t>(csz[a] + csz[pa]); dnnN[A] = g.v; }
            if (lane == 1) { nnW[A] = static_cast<uint16_t>(g.i < nb ? g.i : 0); }
            if (lane == 2) { tieW[A] = static_cast<unsigned char>(nb > 2 ? g.tie : 0); }
        };
        // work is handed out dynamically (the rows of merged clusters cost about twice a pair of plain rows, and a static
        // split left a quarter of the phase waiting at the barrier): merged rows first, then plain rows two at a time
        for (;;) {
            int q = 0;
            if (lane == 0) q = atomicAdd(wctl + 3, 1);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q >= np) break;
#ifdef HR_TIMING
            const long long q0 = __builtin_readcyclecounter();
#endif
            if (stage_rows == 2) staged_merged(newidx[plist[q]]); else do_row(newidx[plist[q]]);
#ifdef HR_TIMING
            hr_slow += __builtin_readcyclecounter() - q0;
#endif
        }
        if (stage_rows > 0) {
            const int step = stage_rows;
            for (;;) {
                int q = 0;
                if (lane == 0) q = atomicAdd(wctl + 2, step);
                q = __builtin_amdgcn_readfirstlane(q);
                if (q >= ns) break;
                const bool two = step == 2 && q + 1 < ns;
#ifdef HR_TIMING
                const long long q2 = __builtin_readcyclecounter();
#endif
                if (src < 0) { if (two) staged(std::integral_constant<int, 2>(), std::true_type(), q); else staged(std::integral_constant<int, 1>(), std::true_type(), q); }
                else         { if (two) staged(std::integral_constant<int, 2>(), std::false_type(), q); else staged(std::integral_constant<int, 1>(), std::false_type(), q); }
#ifdef HR_TIMING
                hr_dual += __builtin_readcyclecounter() - q2;
#endif
            }
        }
        // the gathered forms: rounds whose pairs do not fit the stage even one row at a time (large tasks with little LDS to spare)
        for (;;) {
            if (stage_rows > 0) break;
            int q = 0;
            if (lane == 0) q = atomicAdd(wctl + 2, 2);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q >= ns) break;
            const int A = q;
            if (q + 1 >= ns) { do_row(A); break; }
            const int A2 = q + 1;
            const int a1 = __builtin_amdgcn_readfirstlane(oldidx[A] & 0x7fff), a2 = __builtin_amdgcn_readfirstlane(oldidx[A2] & 0x7fff);
#ifdef HR_TIMING
            const long long q1 = __builtin_readcyclecounter();
#endif
            const hr_gcd r1 = Dsrc + static_cast<long long>(a1) * nld, r2 = Dsrc + static_cast<long long>(a2) * nld;
            const hr_gd w1 = Ddst + static_cast<long long>(A) * nld, w2 = Ddst + static_cast<long long>(A2) * nld;
            if (src >= 0) {
                // Later rounds (the bulk of the work): the source is a scratch matrix whose diagonal holds +inf, so a plain
                // gathered copy needs no diagonal test; per element: one LDS read (old column | merged flag), two loads, two
                // stores and a six-instruction running (min, second min, arg min) per row -- the kernel is bound by the vector
                // ALU (53 instructions per element before this path: SQ_INSTS_VALU, tools/pmc_hclust.sh), not by memory.
                double m1 = HC_INF, s1 = HC_INF, m2 = HC_INF, s2 = HC_INF;
                int i1 = 0x7fffffff, i2 = 0x7fffffff;
                auto upd = [](double &mn, double &sc, int &ix, double v, int B) {
                    sc = fmin(sc, fmax(mn, v));
                    if (v < mn) { mn = v; ix = B; }
                };
                int B0 = lane;
                // passes of 8, 4, 2, 1 columns per lane, none with bounds tests: the waves spend most of their time parked
                // on these loads (SQ_WAIT_ANY 64 % of the wave cycles), so as many as the registers allow go out together
                auto pass = [&](auto U_) {
                    constexpr int U = decltype(U_)::value;
                    for (; B0 + 64 * (U - 1) < ns; B0 += 64 * U) {
                        unsigned mm[U];
                        double x1[U], x2[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) mm[u] = oldidx[B0 + 64 * u];
#pragma unroll
                        for (int u = 0; u < U; ++u) { x1[u] = r1[mm[u]]; x2[u] = r2[mm[u]]; }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int B = B0 + 64 * u;
                            w1[B] = x1[u]; w2[B] = x2[u];
                            upd(m1, s1, i1, x1[u], B); upd(m2, s2, i2, x2[u], B);
                        }
                        if (U == 1) break;
                    }
                };
                pass(std::integral_constant<int, 8>());
                pass(std::integral_constant<int, 4>());
                pass(std::integral_constant<int, 2>());
                pass(std::integral_constant<int, 1>());
                const double n1 = csz[a1], n2 = csz[a2];
                for (int B = ns + lane; B < nb; B += 64) {      // merged columns: d(a, k u l) from d(a,k), d(a,l)
                    const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
                    const double nk_ = csz[k1], nl_ = csz[l1], hQ = dnn[k1];
                    const double v1 = lance_williams(method, r1[k1], r1[l1], hQ, nk_, nl_, n1);
                    const double v2 = lance_williams(method, r2[k1], r2[l1], hQ, nk_, nl_, n2);
                    w1[B] = v1; w2[B] = v2;
                    // equal values: the lower column wins, like the ascending sweep above
                    s1 = fmin(s1, fmax(m1, v1)); if (v1 < m1 || (v1 == m1 && B < i1)) { m1 = v1; i1 = B; }
                    s2 = fmin(s2, fmax(m2, v2)); if (v2 < m2 || (v2 == m2 && B < i2)) { m2 = v2; i2 = B; }
                }
                HrBest g1, g2;
                g1.v = m1; g1.i = i1; g1.tie = 0; g2.v = m2; g2.i = i2; g2.tie = 0;
                g1 = hr_wave(g1); g2 = hr_wave(g2);
                g1.tie |= __ballot(s1 == g1.v) != 0ull ? 1 : 0;     // a lane saw the minimum twice
                g2.tie |= __ballot(s2 == g2.v) != 0ull ? 1 : 0;
                if (lane == 0) { cidN[A] = cid[a1]; cszN[A] = csz[a1]; dnnN[A] = g1.v; cidN[A2] = cid[a2]; cszN[A2] = csz[a2]; dnnN[A2] = g2.v; }
                if (lane == 1) { nnW[A] = static_cast<uint16_t>(g1.i < nb ? g1.i : 0); nnW[A2] = static_cast<uint16_t>(g2.i < nb ? g2.i : 0); }
                if (lane == 2) { tieW[A] = static_cast<unsigned char>(nb > 2 ? g1.tie : 0); tieW[A2] = static_cast<unsigned char>(nb > 2 ? g2.tie : 0); }
#ifdef HR_TIMING
                hr_dual += __builtin_readcyclecounter() - q1;
#endif
                continue;
            }
            HrBest b1, b2;
            b1.v = b2.v = HC_INF; b1.i = b2.i = 0x7fffffff; b1.tie = b2.tie = 0;
            for (int B0 = lane; B0 < ns; B0 += 64 * 8) {
                int bb[8];
                double x1[8], x2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int B = B0 + 64 * u;
                    bb[u] = oldidx[B < ns ? B : ns - 1];
                    x1[u] = r1[bb[u]]; x2[u] = r2[bb[u]];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int B = B0 + 64 * u;
                    if (B < ns) {
                        const double v1 = B == A ? HC_INF : (sq ? x1[u] * x1[u] : x1[u]);
                        const double v2 = B == A2 ? HC_INF : (sq ? x2[u] * x2[u] : x2[u]);
                        w1[B] = v1; w2[B] = v2;
                        if (B != A) { if (v1 < b1.v) { b1.v = v1; b1.i = B; b1.tie = 0; } else if (v1 == b1.v) b1.tie = 1; }
                        if (B != A2) { if (v2 < b2.v) { b2.v = v2; b2.i = B; b2.tie = 0; } else if (v2 == b2.v) b2.tie = 1; }
                    }
                }
            }
            const double n1 = csz[a1], n2 = csz[a2];
            for (int B = ns + lane; B < nb; B += 64) {          // merged columns: d(a, k u l) from d(a,k), d(a,l)
                const int k1 = oldidx[B] & 0x7fff, l1 = partner[k1];
                double d1 = r1[k1], d2 = r1[l1], e1 = r2[k1], e2 = r2[l1];
                if (sq) { d1 *= d1; d2 *= d2; e1 *= e1; e2 *= e2; }
                const double nk_ = csz[k1], nl_ = csz[l1], hQ = dnn[k1];
                const double v1 = lance_williams(method, d1, d2, hQ, nk_, nl_, n1);
                const double v2 = lance_williams(method, e1, e2, hQ, nk_, nl_, n2);
                w1[B] = v1; w2[B] = v2;
                if (v1 < b1.v || (v1 == b1.v && B < b1.i)) { b1.tie = (v1 == b1.v) ? 1 : 0; b1.v = v1; b1.i = B; } else if (v1 == b1.v && B != b1.i) b1.tie = 1;
                if (v2 < b2.v || (v2 == b2.v && B < b2.i)) { b2.tie = (v2 == b2.v) ? 1 : 0; b2.v = v2; b2.i = B; } else if (v2 == b2.v && B != b2.i) b2.tie = 1;
            }
            finish_row(A, a1, b1);
            finish_row(A2, a2, b2);
#ifdef HR_TIMING
            hr_dual += __builtin_readcyclecounter() - q1;
#endif
        }
#ifdef HR_TIMING
        const long long hr_t2 = __builtin_readcyclecounter();
#endif
        if (MODE == 2) return;                                  // the next MODE 1 launch applies the transition
        __syncthreads();
        if (tid == 0) { ctl[0] = 0; }
#ifdef HR_TIMING
        hr_acc_setup += hr_t1 - hr_t0; hr_acc_rebuild += hr_t2 - hr_t1; hr_acc_barrier += (long long)__builtin_readcyclecounter() - hr_t2;
        hr_acc_entries += static_cast<long long>(na) * na + static_cast<long long>(nb) * nb; ++hr_rounds;
        hr_acc_wave_busy += hr_dual + hr_slow;
        if (blockIdx.x == 0 && tid == 0 && (done == 0 || (na < 1200 && na > 1100) || (na < 600 && na > 560) || (na < 300 && na > 280) || (na < 100 && na > 90)))
            printf("round na=%d np=%d nb=%d: setup %lld  rebuild %lld (wave0: plain rows %lld merged rows %lld)  tail-barrier %lld cycles\n", na, np, nb,
                   hr_t1 - hr_t0, hr_t2 - hr_t1, hr_dual, hr_slow, (long long)__builtin_readcyclecounter() - hr_t2);
#endif
#ifdef HR_ROUNDS
        if (blockIdx.x == 0 && tid == 0) printf("R %d %d %d\n", na, np, nb);      // round sizes of task 0 (traffic model)
#endif
        done += np; na = nb; cur ^= 1; src = src < 0 ? 0 : (src ^ 1);
        __syncthreads();
    }
#ifdef HR_TIMING
    if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0)
        printf("task %d: %d rounds, total %lld cycles: setup %lld  rebuild(wave 0 view) %lld (in row work %lld)  end-of-round barrier wait %lld; entries read+written %lld (%.3f cycles per entry)\n",
               (int)blockIdx.x, hr_rounds, (long long)__builtin_readcyclecounter() - hr_start, hr_acc_setup, hr_acc_rebuild, hr_acc_wave_busy, hr_acc_barrier, hr_acc_entries,
               (double)((long long)__builtin_readcyclecounter() - hr_start) / (double)hr_acc_entries);
#endif
    // (5) the sequential algorithm's order: ascending height, lowest index first; ward.D2 reports sqrt
    __syncthreads();
    {
        int npow2 = 1; while (npow2 < n - 1) npow2 <<= 1;
        double *kh = reinterpret_cast<double *>(sm);            // the state is dead: reuse LDS (16 B per entry <= state size)
        int *ki = reinterpret_cast<int *>(kh + npow2);
        int *kj = ki + npow2;
        for (int q = tid; q < npow2; q += HR_THREADS) {
            if (q < n - 1) { kh[q] = crit[q]; ki[q] = ia[q]; kj[q] = ib[q]; } else { kh[q] = HC_INF; ki[q] = 0x7fffffff; kj[q] = 0; }
        }
        __syncthreads();
        for (int size = 2; size <= npow2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (npow2 >> 1); t += HR_THREADS) {
                    const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const double x = kh[lo], y = kh[hi];
                    const bool gt = x > y || (x == y && ki[lo] > ki[hi]);
                    if (gt == up) {
                        kh[lo] = y; kh[hi] = x;
                        const int t1 = ki[lo]; ki[lo] = ki[hi]; ki[hi] = t1;
                        const int t2 = kj[lo]; kj[lo] = kj[hi]; kj[hi] = t2;
                    }
                }
                __syncthreads();
            }
        }
        for (int q = tid; q < n - 1; q += HR_THREADS) { crit[q] = method == 8 ? sqrt(kh[q]) : kh[q]; ia[q] = ki[q]; ib[q] = kj[q]; }
    }
    if (tid == 0) {
        status[blockIdx.x] = 0;
        if (MODE == 1) { reinterpret_cast<int *>(img + (reinterpret_cast<unsigned char *>(ctl) - sm))[10] = 2; atomicSub(remaining, 1); }
    }
}

#ifdef SHARP_LAB       // lab builds only (make LAB=1 -> sharp_amd/variants/libsharp_hip_lab.so; LAB_NOTES.md): the agglomeration forms that were
#include "../../tools/lab/hclust_tri.inc"      // measured and not adopted -- on the upper triangle, append-only first rounds, lazy rows
#include "../../tools/lab/hclust_front.inc"
#include "../../tools/lab/hclust_lazy.inc"
#endif

// ---------------------------------------------------------------------------------------------
// a5a: cutree for every level k = kmin..kmax (level index L = k - kmin), ids by first appearance.
// j2 is absorbed by i2 < j2 at its merge step, so a cluster's representative is its smallest member
// and "first appearance" order is the order of the representatives.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HC_THREADS) void cutree_kernel(const HcMeta *__restrict__ metas, const int *__restrict__ ia_all,
                                                            const int *__restrict__ ib_all, int *__restrict__ lab_all) {
    const HcMeta M = metas[blockIdx.x];
    const int n = M.n;
    const int *ia = ia_all + M.oM, *ib = ib_all + M.oM;
    int *lab = lab_all + M.oLab;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    int *absorbed = reinterpret_cast<int *>(sm);   // merge step at which i stops being a representative
    int *wsum = absorbed + n;                      // [HC_THREADS/64 + 1]
    uint16_t *parent = reinterpret_cast<uint16_t *>(wsum + HC_THREADS / 64 + 1);   // (n <= kHcMaxN < 65536)
    uint16_t *rank = parent + n;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += HC_THREADS) { absorbed[i] = 0x7fffffff; parent[i] = static_cast<uint16_t>(i); }
    __syncthreads();
    for (int s = tid; s < n - 1; s += HC_THREADS) { absorbed[ib[s] - 1] = s; parent[ib[s] - 1] = static_cast<uint16_t>(ia[s] - 1); }
    __syncthreads();
    const int chunk = (n + HC_THREADS - 1) / HC_THREADS;
    for (int L = 0; L < M.nk; ++L) {
        const int k = M.kmin + L;
        const int nm = n - k;                      // merges applied
        // exclusive prefix count of representatives -> 1-based id of each representative
        const int b0 = tid * chunk, b1 = min(n, b0 + chunk);
        int local = 0;
        for (int i = b0; i < b1; ++i) local += (absorbed[i] >= nm);
        int inc = local;
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        if (tid == 0) { int run = 0; for (int q = 0; q < HC_THREADS / 64; ++q) { const int t = wsum[q]; wsum[q] = run; run += t; } }
        __syncthreads();
        int run = wsum[w] + inc - local;
        for (int i = b0; i < b1; ++i) { if (absorbed[i] >= nm) rank[i] = static_cast<uint16_t>(++run); }
        __syncthreads();
        for (int i = tid; i < n; i += HC_THREADS) {
            int r = i;
            while (absorbed[r] < nm) r = parent[r];
            lab[static_cast<long long>(L) * n + i] = rank[r];
        }
        __syncthreads();
    }
}

// one-hot membership of the finest level (k = kmax): H[i][c] = (label_i == c + 1)
__global__ void onehot_kernel(const HcMeta *__restrict__ metas, const int *__restrict__ lab_all, double *__restrict__ H_all) {
    const HcMeta M = metas[blockIdx.y];
    const int *lab = lab_all + M.oLab + static_cast<long long>(M.nk - 1) * M.n;
    double *H = H_all + M.oH;
    const long long tot = static_cast<long long>(M.n) * M.kpad;
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < tot;
         q += static_cast<long long>(gridDim.x) * blockDim.x) {
        const int i = static_cast<int>(q / M.kpad), c = static_cast<int>(q % M.kpad);
        H[q] = (lab[i] == c + 1) ? 1.0 : 0.0;
    }
}

// The finest level's cluster sums WITHOUT the one-hot matrix and its skinny GEMM (48 clusters wide: 75 % of a 64-wide MFMA tile, a K
// loop of 2000 cells; 0.61 ms per chunk of 188 tasks, 0.43 as below, and the 0.04 ms one-hot pass goes too):
//   cluster_sums_kernel     CSt[j][c] = sum over the cells i of finest cluster c of Cr[i][j]   (p x kpad), one workgroup per (64
//                           columns, task): a wave walks every SS_WAVES-th row, adds its 64 entries to the cluster's row of the
//                           wave's LDS table (the label is wave-uniform), the tables are added in wave order at the end.
// It sums in a fixed order (rows ascending per wave, waves in order): the same bits every run.  (The rows' products with the sums,
// G = CS C^T, stay on the MFMA: one thread per cell with the 48 sums of a row j through the scalar cache took 0.77 ms against 0.50.)
constexpr int SS_WAVES = 2, SS_KMAX = 144;            // LDS: SS_WAVES * kpad * 512 B
__global__ __launch_bounds__(64 * SS_WAVES) void cluster_sums_kernel(const HcMeta *__restrict__ metas, const int *__restrict__ lab_all,
                                                                     const double *__restrict__ Cr_all, double *__restrict__ CSt_all) {
    const HcMeta M = metas[blockIdx.y];
    const int j0 = blockIdx.x * 64;
    if (j0 >= M.p) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_sm[];
    double *acc = reinterpret_cast<double *>(ss_sm);                       // [SS_WAVES][kpad][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kpad = M.kpad, n = M.n, p = M.p;
    for (int q = tid; q < SS_WAVES * kpad * 64; q += 64 * SS_WAVES) acc[q] = 0.0;
    __syncthreads();
    const int *lab = lab_all + M.oLab + static_cast<long long>(M.nk - 1) * n;
    const double *Cr = Cr_all + M.oCr;
    const int j = j0 + lane;
    const bool live = j < p;
    double *mine = acc + static_cast<size_t>(wave) * kpad * 64 + lane;
    int i = wave;
    for (; i + 3 * SS_WAVES < n; i += 4 * SS_WAVES) {                     // four rows' loads in flight
        double x[4];
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[u] = __builtin_amdgcn_readfirstlane(lab[i + u * SS_WAVES]) - 1;
            x[u] = live ? Cr[static_cast<long long>(i + u * SS_WAVES) * p + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mine[c[u] * 64] += x[u];
    }
    for (; i < n; i += SS_WAVES) {
        const int c = __builtin_amdgcn_readfirstlane(lab[i]) - 1;
        mine[c * 64] += live ? Cr[static_cast<long long>(i) * p + j] : 0.0;
    }
    __syncthreads();
    double *CSt = CSt_all + M.oCSt;
    for (int q = tid; q < kpad * 64; q += 64 * SS_WAVES) {
        const int c = q >> 6, l = q & 63;
        if (j0 + l >= p) continue;
        double v = acc[c * 64 + l];
#pragma unroll
        for (int w = 1; w < SS_WAVES; ++w) v += acc[(w * kpad + c) * 64 + l];
        CSt[static_cast<long long>(j0 + l) * kpad + c] = v;
    }
}

// copy the pristine distances of symmetric tasks (hclust updates D in place)
__global__ void copy_d_kernel(const HcMeta *__restrict__ metas, const double *__restrict__ Dall, double *__restrict__ D0all) {
    const HcMeta M = metas[blockIdx.y];
    if (!M.symmetric) return;
    const long long tot = static_cast<long long>(M.n) * M.nld;
    const double *D = Dall + M.oD;
    double *D0 = D0all + M.oD0;
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < tot;
         q += static_cast<long long>(gridDim.x) * blockDim.x)
        D0[q] = D[q];
}

// ---------------------------------------------------------------------------------------------
// a5b: per (task, level): median silhouette (cluster::silhouette semantics) and CH ("1-corr").
// Everything is derived from finest-level quantities computed by MFMA GEMMs:
//   T[i][f] = sum_{j in f} d(i,j)   (symmetric tasks: D0 * H;  feature tasks: cnt_f - G[i][f])
//   G[i][f] = c_i . sum_{j in f} c_j,   Q[f][f'] = (sum_f c) . (sum_f' c)
// A level-k cluster is a union of finest clusters; its sums add the finest columns in ascending order.
// ---------------------------------------------------------------------------------------------
constexpr int ST_THREADS = 512;
constexpr size_t ST_LDS_MAX = 160 * 1024;
// LDS of one stats workgroup: sil[npow2] (median by bitonic sort), part[ST_THREADS], and seven per-cluster arrays of kcap entries
inline size_t stats_lds_bytes(int max_n, int kcap) {
    int npow2 = 1; while (npow2 < max_n) npow2 <<= 1;
    return static_cast<size_t>(npow2) * 8 + ST_THREADS * 8 + 2 * static_cast<size_t>(kcap) * 8 + (5 * static_cast<size_t>(kcap) + 8) * 4;
}

__global__ __launch_bounds__(ST_THREADS) void stats_kernel(const HcMeta *__restrict__ metas, const int *__restrict__ lab_all,
                                                           const double *__restrict__ T_all, const double *__restrict__ G_all,
                                                           const double *__restrict__ Q_all, const double *__restrict__ nrm_all,
                                                           double *__restrict__ out_all, int count, int max_nk, int kcap) {
    // One workgroup per (task, level).  The levels of a task all read the task's G (n x kpad): the linear workgroup id is
    // dealt so that the eight tasks of a group sit on the eight XCDs (workgroups go round-robin to XCDs) and G is
    // fetched into one L2 once instead of once per level (34 GB -> 0.3 GB of HBM reads per step).
    const long long B = blockIdx.x;
    const long long per_group = 8LL * max_nk;
    const int zt = static_cast<int>(B / per_group) * 8 + static_cast<int>(B % 8);
    if (zt >= count) return;
    const HcMeta M = metas[zt];
    const int L = static_cast<int>((B % per_group) / 8);
    if (L >= M.nk) return;
    const int n = M.n, k = M.kmin + L, kf = M.kmax, kpad = M.kpad;
    const int *lab = lab_all + M.oLab + static_cast<long long>(L) * n;
    const int *labF = lab_all + M.oLab + static_cast<long long>(M.nk - 1) * n;
    const double *T = T_all + M.oT, *G = G_all + M.oG, *Q = Q_all + M.oQ, *nrm = nrm_all + M.oNrm;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    int npow2 = 1; while (npow2 < n) npow2 <<= 1;
    double *sil = reinterpret_cast<double *>(sm);            // npow2
    double *part = sil + npow2;                              // ST_THREADS
    double *cn2 = part + ST_THREADS;                         // k  : |sum_c|^2
    double *ctot = cn2 + kcap;                               // k  : sum_c . total
    int *cnt = reinterpret_cast<int *>(ctot + kcap);         // k
    int *cntF = cnt + kcap;                                  // kf
    int *fm = cntF + kcap;                                   // kf : level cluster (0-based) of finest cluster f
    int *start = fm + kcap;                                  // k + 1
    int *order = start + kcap + 1;                           // kf : finest clusters grouped by level cluster
    const int tid = threadIdx.x;
    for (int c = tid; c < kcap; c += ST_THREADS) { cnt[c] = 0; cntF[c] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += ST_THREADS) {
        atomicAdd(&cnt[lab[i] - 1], 1);
        atomicAdd(&cntF[labF[i] - 1], 1);
        fm[labF[i] - 1] = lab[i] - 1;
    }
    __syncthreads();
    // finest clusters grouped by level cluster (ascending inside a group): counts, a prefix over k <= kf entries, one thread per group
    for (int c = tid; c <= k; c += ST_THREADS) start[c] = 0;
    __syncthreads();
    for (int f = tid; f < kf; f += ST_THREADS) atomicAdd(&start[fm[f] + 1], 1);
    __syncthreads();
    if (tid == 0) for (int c = 0; c < k; ++c) start[c + 1] += start[c];
    // the kf x kf Gram matrix of the finest clusters' sums goes through LDS (the space of sil[], written later): the sums below
    // re-read it up to (group size) x (group size + kf) times per group, which took 3 of the kernel's 4.5 ms as dependent global loads
    const bool q_lds = kf * kf <= npow2;
    if (q_lds) for (int e = tid; e < kf * kf; e += ST_THREADS) sil[e] = Q[static_cast<long long>(e / kf) * kpad + e % kf];
    __syncthreads();
    for (int c = tid; c < k; c += ST_THREADS) {
        int pos = start[c];
        for (int f = 0; f < kf; ++f) if (fm[f] == c) order[pos++] = f;
    }
    __syncthreads();
    double tot2 = 0.0;
    for (int c = tid; c < k; c += ST_THREADS) {
        double a = 0.0, b = 0.0;
        for (int q = start[c]; q < start[c + 1]; ++q) {
            if (q_lds) {
                const double *qr = sil + order[q] * kf;
                for (int q2 = start[c]; q2 < start[c + 1]; ++q2) a += qr[order[q2]];
                for (int f = 0; f < kf; ++f) b += qr[f];
            } else {
                const double *qr = Q + static_cast<long long>(order[q]) * kpad;
                for (int q2 = start[c]; q2 < start[c + 1]; ++q2) a += qr[order[q2]];
                for (int f = 0; f < kf; ++f) b += qr[f];
            }
        }
        cn2[c] = a; ctot[c] = b;
    }
    __syncthreads();
    for (int c = 0; c < k; ++c) tot2 += ctot[c];             // |total|^2 (every thread, same order)
    const bool tfromG = !M.symmetric;
    const int nld = M.nld;
    double wpart = 0.0;
    for (int i = tid; i < n; i += ST_THREADS) {
        const int own = lab[i] - 1;
        // T and G are stored transposed (kpad x nld): for a fixed finest cluster f the lanes read consecutive cells.
        // The kf finest clusters are walked in the order that groups them by level cluster, eight loads at a time.
        const double *Ti = T + i, *Gi = G + i;
        double a = 0.0, bmin = 0.0, gown = 0.0;
        bool have_b = false;
        int c = 0;
        while (c < k && start[c + 1] == start[c]) ++c;           // (levels never have empty clusters; defensive)
        double sc = 0.0, gc = 0.0;
        for (int q0 = 0; q0 < kf; q0 += 8) {
            double gv[8], tv[8];
            int fv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u < kf ? q0 + u : kf - 1;
                fv[u] = order[q];
                gv[u] = Gi[static_cast<long long>(fv[u]) * nld];
                tv[u] = tfromG ? 0.0 : Ti[static_cast<long long>(fv[u]) * nld];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u;
                if (q < kf) {
                    sc += tfromG ? (static_cast<double>(cntF[fv[u]]) - gv[u]) : tv[u];
                    gc += gv[u];
                    if (q + 1 == start[c + 1]) {                   // cluster c complete
                        if (c == own) { a = sc / static_cast<double>(cnt[c] - 1); gown = gc; }
                        else { const double bb = sc / static_cast<double>(cnt[c]); if (!have_b || bmin > bb) { bmin = bb; have_b = true; } }
                        sc = 0.0; gc = 0.0;
                        ++c;
                        while (c < k && start[c + 1] == start[c]) ++c;
                    }
                }
            }
        }
        double s = 0.0;
        if (cnt[own] > 1 && bmin != a) s = (bmin - a) / fmax(a, bmin);
        sil[i] = s;
        double r = gown / (nrm[i] * sqrt(cn2[own]));
        r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
        wpart += (1.0 - r) * (1.0 - r);
    }
    for (int i = n + tid; i < npow2; i += ST_THREADS) sil[i] = HC_INF;
    part[tid] = wpart;
    __syncthreads();
    // bitonic sort of sil[0..npow2)
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (npow2 >> 1); t += ST_THREADS) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double x = sil[lo], y = sil[hi];
                if ((x > y) == up) { sil[lo] = y; sil[hi] = x; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        double W = 0.0;
        for (int q = 0; q < ST_THREADS; ++q) W += part[q];
        double B = 0.0;
        for (int c = 0; c < k; ++c) {
            double r = ctot[c] / (sqrt(cn2[c]) * sqrt(tot2));
            r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
            B += static_cast<double>(cnt[c]) * (1.0 - r) * (1.0 - r);
        }
        const double ch = (B / static_cast<double>(k - 1)) / (W / static_cast<double>(n - k));
        const double med = (n & 1) ? sil[n / 2] : (sil[n / 2 - 1] + sil[n / 2]) / 2;
        double *out = out_all + M.oOut;
        out[L] = med;
        out[M.nk + L] = ch;
    }
}

#include "hclust_stats.inc"

// ---------------------------------------------------------------------------------------------
// a5b for MANY candidate levels (the cross-block sMetaC of a run of >= 1e6 cells tries k = n/50000 .. n/5000: 1801 levels at 1e7
// cells, R/sMetaC.R:110-119).  stats_kernel recomputes every level from the finest-level quantities, O(n kf) per level and a
// per-cluster O(size^2) Gram sum by one thread: 0.4 - 0.75 s for 2200 - 8000 rows.  Consecutive levels differ by ONE merge, so here the
// per-level cluster sums are carried from the finest level down:
//   ml_prep_kernel   (one workgroup per task, levels in sequence): |sum_c|^2 of the merged cluster (Gram matrix of the cluster sums
//                    updated in place), sum_c . total, the between-cluster term of CH;
//   ml_cells_kernel  (one wave per cell, all levels): the cell's sums of distances / products per cluster live in the wave's LDS and
//                    follow the merges; per level the silhouette width (own mean, minimum over the other clusters' means) and the
//                    within term of CH -> s[i][L], w[i][L];
//   ml_level_kernel  (one workgroup per level): median of s[.][L] (bitonic sort in LDS), sum of w[.][L] in a fixed order, CH.
// The merges (r1 <- r2 in finest-cluster ids, r1 < r2) come from the host (a replay of the merge list).  Sums of a merged cluster are
// (sum of r1) + (sum of r2): a different association than stats_kernel's from-scratch sums, equal to rounding.
// ---------------------------------------------------------------------------------------------
constexpr int ML_WAVES = 3;          // cells in flight per workgroup: 44 B of LDS per finest cluster and wave
struct MlMeta {
    long long oS;                    // n * nk doubles: s[i][L]; w follows at oS + n * nk
    long long oMerge;                // nk entries of r1 / r2 / cn2 of the merged cluster / B of the level
    long long oFin;                  // kf entries of cntF (int) / cn2F / ctotF
};

__global__ __launch_bounds__(1024) void ml_prep_kernel(const HcMeta *__restrict__ metas, const MlMeta *__restrict__ mls, const int *__restrict__ lab_all,
                                                       double *__restrict__ Q_all, const int *__restrict__ r1_all, const int *__restrict__ r2_all,
                                                       double *__restrict__ cn2m_all, double *__restrict__ B_all, int *__restrict__ cntF_all,
                                                       double *__restrict__ cn2F_all, double *__restrict__ tot2_all) {
    const HcMeta M = metas[blockIdx.x];
    const MlMeta X = mls[blockIdx.x];
    const int n = M.n, kf = M.kmax, kpad = M.kpad, nk = M.nk;
    const int *labF = lab_all + M.oLab + static_cast<long long>(nk - 1) * n;
    double *Q = Q_all + M.oQ;
    const int *r1s = r1_all + X.oMerge, *r2s = r2_all + X.oMerge;
    double *cn2m = cn2m_all + X.oMerge, *Bl = B_all + X.oMerge;
    int *cntF = cntF_all + X.oFin;
    double *cn2F = cn2F_all + X.oFin;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    double *cn2 = reinterpret_cast<double *>(sm);        // kf
    double *ctot = cn2 + kf;                             // kf
    double *part = ctot + kf;                            // 1024
    int *cnt = reinterpret_cast<int *>(part + 1024);     // kf
    const int tid = threadIdx.x;
    for (int f = tid; f < kf; f += 1024) cnt[f] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&cnt[labF[i] - 1], 1);
    // |S_f|^2 and S_f . total (row sums of the Gram matrix of the finest clusters' sums, ascending)
    for (int f = tid; f < kf; f += 1024) {
        const double *qr = Q + static_cast<long long>(f) * kpad;
        double b = 0.0;
        for (int g = 0; g < kf; ++g) b += qr[g];
        ctot[f] = b;
        cn2[f] = qr[f];
    }
    __syncthreads();
    for (int f = tid; f < kf; f += 1024) { cntF[f] = cnt[f]; cn2F[f] = cn2[f]; }
    double tot2 = 0.0;
    for (int f = 0; f < kf; ++f) tot2 += ctot[f];        // |total|^2 (every thread, same order)
    if (tid == 0) tot2_all[blockIdx.x] = tot2;
    auto bterm = [&](int r) {
        double rc = ctot[r] / (sqrt(cn2[r]) * sqrt(tot2));
        rc = rc > 1.0 ? 1.0 : (rc < -1.0 ? -1.0 : rc);
        return static_cast<double>(cnt[r]) * (1.0 - rc) * (1.0 - rc);
    };
    // between-cluster term at the finest level: fixed assignment of clusters to threads, partial sums added in thread order
    {
        double b = 0.0;
        for (int f = tid; f < kf; f += 1024) b += bterm(f);
        part[tid] = b;
        __syncthreads();
        if (tid == 0) { double B = 0.0; for (int q = 0; q < 1024; ++q) B += part[q]; Bl[nk - 1] = B; part[0] = B; }
        __syncthreads();
    }
    double B = part[0];
    __syncthreads();
    for (int L = nk - 2; L >= 0; --L) {
        const int r1 = r1s[L], r2 = r2s[L];
        const double cross = Q[static_cast<long long>(r1) * kpad + r2];          // S_r1 . S_r2
        // Gram matrix of the cluster sums: row and column r1 take r2's (entries of dead clusters are never read again)
        for (int x = tid; x < kf; x += 1024) {
            if (x != r1 && x != r2) {
                const double v = Q[static_cast<long long>(r1) * kpad + x] + Q[static_cast<long long>(r2) * kpad + x];
                Q[static_cast<long long>(r1) * kpad + x] = v;
                Q[static_cast<long long>(x) * kpad + r1] = v;
            }
        }
        if (tid == 0) {
            const double t_old = bterm(r1) + bterm(r2);
            const double c2 = (cn2[r1] + cn2[r2]) + 2.0 * cross;
            cn2[r1] = c2; ctot[r1] = ctot[r1] + ctot[r2]; cnt[r1] = cnt[r1] + cnt[r2];
            Q[static_cast<long long>(r1) * kpad + r1] = c2;
            B = (B - t_old) + bterm(r1);
            cn2m[L] = c2; Bl[L] = B;
        }
        __syncthreads();
    }
}

// One wave per cell.  G / T: n x kpad ROW-major here (a cell's finest-level products / distance sums are one contiguous row).
__global__ __launch_bounds__(64 * ML_WAVES) void ml_cells_kernel(const HcMeta *__restrict__ metas, const MlMeta *__restrict__ mls, int task,
                                                                 const int *__restrict__ lab_all, const double *__restrict__ T_all,
                                                                 const double *__restrict__ G_all, const double *__restrict__ nrm_all,
                                                                 const int *__restrict__ r1_all, const int *__restrict__ r2_all,
                                                                 const double *__restrict__ cn2m_all, const int *__restrict__ cntF_all,
                                                                 const double *__restrict__ cn2F_all, double *__restrict__ S_all) {
    const HcMeta M = metas[task];
    const MlMeta X = mls[task];
    const int n = M.n, kf = M.kmax, kpad = M.kpad, nk = M.nk;
    const int *labF = lab_all + M.oLab + static_cast<long long>(nk - 1) * n;
    const double *T = T_all + M.oT, *G = G_all + M.oG, *nrm = nrm_all + M.oNrm;
    const int *r1s = r1_all + X.oMerge, *r2s = r2_all + X.oMerge;
    const double *cn2m = cn2m_all + X.oMerge;
    const int *cntF = cntF_all + X.oFin;
    const double *cn2F = cn2F_all + X.oFin;
    double *S = S_all + X.oS, *Wt = S + static_cast<long long>(n) * nk;
    const bool tfromG = !M.symmetric;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = static_cast<size_t>(kf) * (8 + 8 + 2 + 2 + 2);
    unsigned char *base = sm + ((per_wave + 15) & ~static_cast<size_t>(15)) * wave;
    double *st = reinterpret_cast<double *>(base);        // sum of distances to the members of cluster r (r = its smallest finest id)
    double *sg = st + kf;                                 // c_i . (sum of the members' centred rows)
    uint16_t *cnt = reinterpret_cast<uint16_t *>(sg + kf);
    uint16_t *live = cnt + kf;                            // the clusters of the current level, any order
    uint16_t *pos = live + kf;                            // position of r in live[]
    for (long long i = static_cast<long long>(blockIdx.x) * ML_WAVES + wave; i < n; i += static_cast<long long>(gridDim.x) * ML_WAVES) {
        const double *Gi = G + i * kpad, *Ti = T + i * kpad;
        for (int f = lane; f < kf; f += 64) {
            const double g = Gi[f];
            sg[f] = g;
            st[f] = tfromG ? (static_cast<double>(cntF[f]) - g) : Ti[f];
            cnt[f] = static_cast<uint16_t>(cntF[f]);
            live[f] = static_cast<uint16_t>(f); pos[f] = static_cast<uint16_t>(f);
        }
        int own = labF[i] - 1, k = kf;
        double cn2own = cn2F[own];
        const double nr = nrm[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int L = nk - 1; L >= 0; --L) {
            if (L < nk - 1) {                             // the merge that leads from level L + 1 to level L
                const int r1 = r1s[L], r2 = r2s[L];
                if (lane == 0) {
                    st[r1] = st[r1] + st[r2]; sg[r1] = sg[r1] + sg[r2];
                    cnt[r1] = static_cast<uint16_t>(cnt[r1] + cnt[r2]);
                    const int p2 = pos[r2], last = live[k - 1];
                    live[p2] = static_cast<uint16_t>(last); pos[last] = static_cast<uint16_t>(p2);
                }
                --k;
                if (own == r2 || own == r1) { own = r1; cn2own = cn2m[L]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // b = the smallest mean distance to another cluster (a minimum: any order)
            double bmin = HC_INF;
            for (int q = lane; q < k; q += 64) {
                const int r = live[q];
                if (r != own) { const double bb = st[r] / static_cast<double>(cnt[r]); bmin = bb < bmin ? bb : bmin; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(bmin, o); bmin = y < bmin ? y : bmin; }
            if (lane == 0) {
                const int co = cnt[own];
                const double a = st[own] / static_cast<double>(co - 1);
                double s = 0.0;
                if (co > 1 && bmin != a) s = (bmin - a) / fmax(a, bmin);
                double r = sg[own] / (nr * sqrt(cn2own));
                r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
                S[i * nk + L] = s;
                Wt[i * nk + L] = (1.0 - r) * (1.0 - r);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(ST_THREADS) void ml_level_kernel(const HcMeta *__restrict__ metas, const MlMeta *__restrict__ mls, int task,
                                                              const double *__restrict__ S_all, const double *__restrict__ B_all,
                                                              double *__restrict__ out_all) {
    const HcMeta M = metas[task];
    const MlMeta X = mls[task];
    const int n = M.n, nk = M.nk, L = blockIdx.x, k = M.kmin + L;
    const double *S = S_all + X.oS, *Wt = S + static_cast<long long>(n) * nk;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    int npow2 = 1; while (npow2 < n) npow2 <<= 1;
    double *sil = reinterpret_cast<double *>(sm);            // npow2
    double *part = sil + npow2;                              // ST_THREADS
    const int tid = threadIdx.x;
    double wpart = 0.0;
    for (int i = tid; i < n; i += ST_THREADS) {              // (the same assignment of cells to threads as stats_kernel)
        sil[i] = S[static_cast<long long>(i) * nk + L];
        wpart += Wt[static_cast<long long>(i) * nk + L];
    }
    for (int i = n + tid; i < npow2; i += ST_THREADS) sil[i] = HC_INF;
    part[tid] = wpart;
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (npow2 >> 1); t += ST_THREADS) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double x = sil[lo], y = sil[hi];
                if ((x > y) == up) { sil[lo] = y; sil[hi] = x; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        double W = 0.0;
        for (int q = 0; q < ST_THREADS; ++q) W += part[q];
        const double B = B_all[X.oMerge + L];
        const double ch = (B / static_cast<double>(k - 1)) / (W / static_cast<double>(n - k));
        const double med = (n & 1) ? sil[n / 2] : (sil[n / 2 - 1] + sil[n / 2]) / 2;
        double *out = out_all + M.oOut;
        out[L] = med;
        out[M.nk + L] = ch;
    }
}

// gather the chosen label column of every task into one contiguous buffer
__global__ void pack_labels_kernel(const HcMeta *__restrict__ metas, const int *__restrict__ lab_all, const int *__restrict__ chosen,
                                   const long long *__restrict__ dst_off, int *__restrict__ dst) {
    const HcMeta M = metas[blockIdx.y];
    const int *src = lab_all + M.oLab + static_cast<long long>(chosen[blockIdx.y]) * M.n;
    int *d = dst + dst_off[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M.n; i += gridDim.x * blockDim.x) d[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {

// grow-only pinned host buffer: the per-chunk result downloads (heights 3 MB, labels 1.5 MB at cfg2) go through it at PCIe speed
// instead of through the driver's staging of pageable memory
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        SHARP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
};

struct Workspace {
    PinnedBuf<double> h_out, h_height;
    PinnedBuf<int> h_packed;
    DevBuf<double> D, D0, S0, S1, Cr, Ct, nrm, height, H, T, G, CSt, Q, out;
    DevBuf<int> ia, ib, lab, chosen, packed, status, remaining;
    DevBuf<unsigned char> img;          // LDS state images of the round-per-launch agglomeration
    DevBuf<double> nnp;                 // per-tile row minima of the distance matrices (HcMeta::nn)
    DevBuf<unsigned char> seqstate;     // nearest-neighbour state of the sequential kernel for tasks beyond kHcLdsMaxN observations
    // many-levels statistics (ml_*_kernel)
    DevBuf<MlMeta> mlmeta;
    DevBuf<double> mlS, mlcn2m, mlB, mlcn2F, mltot2;
    DevBuf<int> mlr1, mlr2, mlcntF;
    DevBuf<long long> packoff;
    DevBuf<HcMeta> meta;
    DevBuf<RowPrepTask> prep;
    DevBuf<GemmTask> gemm;
#ifdef SHARP_LAB
    DevBuf<DistI8Task> i8;              // the sliced-integer form of the distance GEMM (tools/lab/gemm_i8.hip): descriptors, digits, row scales
#endif
    DevBuf<signed char> sl;
    DevBuf<double> slscale;
};
// Two sets of buffers so that consecutive chunks of tasks can be in flight together (run_chunks); the scratch of the agglomeration
// itself (S0, S1, img, remaining) is only ever used by one chunk at a time and always comes from set 0.
// Slot 3: the third set of a batch of three chunks or more; slot 2: a batch started from another batch's progress callback;
// slots 4 and 5: batches whose distance matrices are built ahead of time for the NEXT block of a SHARP_unlimited run (hc_prefetch_*).
struct WorkspaceSets { Workspace w[6]; };
Workspace &ws(int slot = 0) { return per_slot<WorkspaceSets>().w[slot]; }
// The device buffers of every set of the calling thread's slot go back to the driver (sharp_trim for worker and helper slots: a process
// that ran several slots on ONE GPU -- the tests do -- otherwise keeps tens of GB per slot); the caller has synchronised the device.
static void release_workspaces_of_slot() {
    for (Workspace &W : per_slot<WorkspaceSets>().w) {
        for (DevBuf<double> *b : {&W.D, &W.D0, &W.S0, &W.S1, &W.Cr, &W.Ct, &W.nrm, &W.height, &W.H, &W.T, &W.G, &W.CSt, &W.Q, &W.out,
                                  &W.mlS, &W.mlcn2m, &W.mlB, &W.mlcn2F, &W.mltot2, &W.slscale}) b->release();
        for (DevBuf<int> *b : {&W.ia, &W.ib, &W.lab, &W.chosen, &W.packed, &W.status, &W.remaining, &W.mlr1, &W.mlr2, &W.mlcntF}) b->release();
        W.img.release(); W.seqstate.release(); W.sl.release(); W.nnp.release();
    }
}

inline long long rup(long long v, long long a) { return (v + a - 1) / a * a; }

// Batches of at most this many tasks run the round-per-launch agglomeration (a task spread over several workgroups); above, one
// workgroup per task in one launch.  Measured at 2000 observations per task: 75 tasks 8.7 vs 11.7 ms, 125 tasks 11.1 vs 12.1 ms,
// 150 tasks 14.0 vs 13.2 ms, 175 tasks 15.3 vs 14.4 ms (tools/bench_hc.py, SHARP_HC_SPLIT=1 / 0).
constexpr int kHcSplitMaxTasks = 136;
// more candidate levels than this in a chunk: the incremental per-level statistics (ml_*_kernel) instead of stats_kernel
constexpr int kMlMinLevels = 256;
static int ml_min_levels() { return knobs().ml_min_levels > 0 ? knobs().ml_min_levels : kMlMinLevels; }   // (SHARP_ML_MIN_LEVELS: tests)
constexpr int kHcSplitMinObs = 1000;

// model selection, R/get_opt_hclust.R:162-229
void select_level(const HcParams &prm, int n, int kmin, int nk, const double *msil, const double *CH, const double *height,
                  int &oind, int &branch, int &rc) {
    double mx = msil[0];
    for (int c = 1; c < nk; ++c) if (msil[c] > mx) mx = msil[c];
    std::vector<int> ties;
    for (int c = 0; c < nk; ++c) if (msil[c] == mx) ties.push_back(c);
    oind = ties.empty() ? 1 : ties[(ties.size() + 1) / 2 - 1] + 1;   // tmp[ceiling(length(tmp)/2)]
    branch = 0;
    if (mx <= prm.sil_thre) {
        branch = 1;
        int wm = 0;
        for (int c = 1; c < nk; ++c) if (CH[c] > CH[wm]) wm = c;     // which.max: first maximum
        oind = wm + 1;
        if (oind == 1) {
            const int nh = n - 1, t0 = nh > 10 ? nh - 10 : 0, tl = nh - t0;
            const double *tmp = height + t0;
            int pind = -1;
            for (int i = 0; i + 1 < tl; ++i)
                if (tmp[i + 1] - tmp[i] > (prm.height_Ntimes - 1) * tmp[i]) { pind = i; break; }
            if (pind >= 0) {
                branch = 2;
                const double opth = (tmp[pind] + tmp[pind + 1]) / 2;
                int idx = n;                                         // which.max(c(height, Inf) > opth)
                for (int i = 0; i < n - 1; ++i) if (height[i] > opth) { idx = i + 1; break; }
                const int kk = n + 1 - idx;
                oind = kk - 1;                                       // "for consistency": assumes kmin == 2
            }
        }
    }
    (void)kmin;
    if (oind < 1 || oind > nk) { rc |= SHARP_WARN_RANGE; oind = oind < 1 ? 1 : nk; }
}

}  // namespace

// ---- the decision log (hclust.hpp) -------------------------------------------------------------------------------------------------
namespace {
struct DecisionLog { std::mutex mu; bool on = false; std::vector<double> rows; };
DecisionLog &dlog() { static DecisionLog *L = new DecisionLog; return *L; }
inline double dnan() { return std::numeric_limits<double>::quiet_NaN(); }
}  // namespace
bool decision_log_on() { return dlog().on || knobs().decision_log; }
void decision_log_set(bool on) { DecisionLog &L = dlog(); std::lock_guard<std::mutex> lk(L.mu); L.on = on; L.rows.clear(); }
void decision_log_add(const double *row) { DecisionLog &L = dlog(); std::lock_guard<std::mutex> lk(L.mu); L.rows.insert(L.rows.end(), row, row + kDecisionCols); }
void decision_log_override(int level, int block, int k_taken) {
    DecisionLog &L = dlog();
    std::lock_guard<std::mutex> lk(L.mu);
    for (size_t r = L.rows.size() / kDecisionCols; r-- > 0;) {        // (the latest row of that call)
        double *row = L.rows.data() + r * kDecisionCols;
        if (static_cast<int>(row[0]) == level && static_cast<int>(row[1]) == block) { row[12] = k_taken; return; }
    }
}
int decision_log_fetch(double *rows, int cap_rows) {
    DecisionLog &L = dlog();
    std::lock_guard<std::mutex> lk(L.mu);
    const int nr = static_cast<int>(L.rows.size() / kDecisionCols);
    std::vector<int> ord(nr);
    for (int i = 0; i < nr; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
        const double *x = L.rows.data() + static_cast<size_t>(a) * kDecisionCols, *y = L.rows.data() + static_cast<size_t>(b) * kDecisionCols;
        for (int c = 0; c < 4; ++c) if (x[c] != y[c]) return x[c] < y[c];
        return false;
    });
    for (int i = 0; i < nr && i < cap_rows; ++i)
        std::copy(L.rows.data() + static_cast<size_t>(ord[i]) * kDecisionCols, L.rows.data() + static_cast<size_t>(ord[i] + 1) * kDecisionCols,
                  rows + static_cast<size_t>(i) * kDecisionCols);
    return nr;
}
// the row of one decision: the same arithmetic, line for line, as the CPU checker's decision_row (the two logs are compared entry by entry)
void decision_row(const HcParams &prm, int n, int kmin, int nk, const double *msil, const double *CH, const double *height, int oind,
                  int branch, double *row) {
    for (int c = 0; c < kDecisionCols; ++c) row[c] = dnan();
    row[0] = prm.dec_level; row[1] = prm.dec_block; row[2] = prm.dec_k; row[3] = prm.dec_fold; row[4] = n;
    row[5] = branch; row[6] = kmin + oind - 1; row[12] = 0; row[13] = nk;
    if (prm.N_cluster > 0) { row[5] = 3; row[6] = prm.N_cluster; row[7] = 1; row[8] = msil[0]; row[13] = 1; return; }
    double mx = msil[0];
    for (int c = 1; c < nk; ++c) if (msil[c] > mx) mx = msil[c];
    row[10] = mx - prm.sil_thre;
    const double *val = branch == 0 ? msil : CH;
    double best = branch == 0 ? mx : val[0];
    if (branch != 0) for (int c = 1; c < nk; ++c) if (val[c] > best) best = val[c];
    int ties = 0;
    double second = dnan();
    for (int c = 0; c < nk; ++c) {
        if (val[c] == best) ++ties;
        else if (val[c] < best && (!(second == second) || val[c] > second)) second = val[c];
    }
    row[7] = ties; row[8] = best; row[9] = second;
    if (branch >= 1 && (branch == 2 || CH[0] == best)) {                    // which.max(CHind) == 1: the height rule was consulted (:196-210)
        bool first = true;
        for (int c = 1; c < nk; ++c) if (CH[c] > CH[0]) first = false;
        if (first) {
            const int nh = n - 1, t0 = nh > 10 ? nh - 10 : 0, tl = nh - t0;
            const double *tmp = height + t0;
            double rmax = dnan();
            for (int i = 0; i + 1 < tl; ++i) {
                const double dif = tmp[i + 1] - tmp[i], den = (prm.height_Ntimes - 1) * tmp[i];
                const double r = den > 0 ? dif / den : (dif > 0 ? std::numeric_limits<double>::infinity() : 0.0);
                if (branch == 2) { if (dif > den) { rmax = r; break; } }
                else if (!(rmax == rmax) || r > rmax) rmax = r;
            }
            row[11] = rmax;
        }
    }
}

namespace {

// One chunk of tasks: enqueue_chunk() puts all its device work on streams, finish_chunk() fetches the statistics, selects the
// levels on the host and fetches the labels.  `pipe` = the chunk is one of several in flight (run on its slot's own stream, ordered
// against its neighbours by events, see get_opt_hclust_batch); otherwise everything is relative to the library's main stream.
struct ChunkJob {
    size_t i0 = 0, i1 = 0;
    int T = 0, slot = 0;
    bool pipe = false, first = true;
    hipEvent_t input_ready = nullptr;          // pipelined: what the chunk's stream waits for before it reads its tasks' inputs (nullptr: EV.in)
    bool i8 = false;          // the distance matrices through the sliced-integer GEMM
    std::vector<HcMeta> metas;
    std::vector<RowPrepTask> prep;
    std::vector<GemmTask> g;
    long long oOut = 0, oM = 0, oLab = 0;
    int max_n = 0, max_p = 0, max_nk = 0, max_kpad = 0, NS = 1;
    bool split = false;                        // round-per-launch agglomeration (few tasks)
    bool ml = false;                           // many candidate levels: the incremental statistics kernels
    std::vector<MlMeta> mlmetas;
    int ml_off = 0, ml_cnt = 0, mlt_off = 0, mlt_cnt = 0;   // GEMM descriptors of the row-major G and T
    bool seq_pending = false;                  // the sequential fallback kernel is still to be launched (with the statistics phase)
    bool has_next = false;                     // pipelined: another chunk follows (its distance GEMM is enqueued before this one's tail)
    bool one_range = false;                    // everything on the current stream (a batch prepared ahead of time, hc_prefetch_begin)
    int scratch_slot = 0;                      // whose S0 / S1 / img / remaining the agglomeration uses (set 0 unless the batch is nested)
    int prev_slot = 1, next_slot = 1;          // pipelined: the slots of the chunk before and after this one (two or three slots in rotation)
    int next2_slot = -1;                       // three slots: the chunk after the next, whose distance GEMM the statistics also let pass
    hipEvent_t mid_event = nullptr;            // recorded behind round `mid_round` of the round-per-launch agglomeration (if it gets that far)
    int mid_round = 8;
    bool mid_recorded = false;
    struct Range { int t0, t1; int off[5], cnt[5]; int off8 = 0; bool any_sym, any_feat; };
    std::vector<Range> ranges;
};
enum : int { PH_DIST = 1, PH_AGGLO = 2, PH_STATS = 4, PH_ALL = 7 };
struct PipeEvents {
    hipEvent_t in = nullptr, out[8] = {nullptr}, gemm[6] = {nullptr}, hc[6] = {nullptr}, done[6] = {nullptr}, prep[6] = {nullptr};   // (per workspace slot)
};
PipeEvents &pipe_events() {
    PipeEvents &e = per_slot<PipeEvents>();
    if (!e.in) {
        SHARP_HIP_CHECK(hipEventCreateWithFlags(&e.in, hipEventDisableTiming));
        for (auto &x : e.out) SHARP_HIP_CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        for (int q = 0; q < 6; ++q) {
            SHARP_HIP_CHECK(hipEventCreateWithFlags(&e.gemm[q], hipEventDisableTiming));
            SHARP_HIP_CHECK(hipEventCreateWithFlags(&e.hc[q], hipEventDisableTiming));
            SHARP_HIP_CHECK(hipEventCreateWithFlags(&e.done[q], hipEventDisableTiming));
            SHARP_HIP_CHECK(hipEventCreateWithFlags(&e.prep[q], hipEventDisableTiming));
        }
    }
    return e;
}

// Descriptors, workspace and uploads of a chunk (before its first phase).
void setup_chunk(const std::vector<HcTask> &tasks, ChunkJob &J) {
    Ctx &c = ctx();
    Workspace &W = ws(J.slot);
    Workspace &W0 = ws(J.scratch_slot);        // agglomeration scratch: shared, one chunk's agglomeration runs at a time
    PipeEvents &EV = pipe_events();
    const size_t i0 = J.i0;
    const int T = J.T = static_cast<int>(J.i1 - J.i0);
    std::vector<HcMeta> &metas = J.metas;
    metas.assign(T, HcMeta());
    // a pipelined chunk lives on its slot's stream from its first upload on (the slot's buffers are reused by the chunk after next,
    // which is on the same stream); its inputs come from the main stream (EV.in, recorded by the caller)
    hipStream_t chunk_stream = J.pipe ? c.aux_stream(J.slot) : c.stream;
    if (J.pipe) SHARP_HIP_CHECK(hipStreamWaitEvent(chunk_stream, J.input_ready ? J.input_ready : EV.in, 0));
    StreamScope chunk_scope(chunk_stream);
    long long oD = 0, oD0 = 0, oCr = 0, oCt = 0, oN = 0, oM = 0, oLab = 0, oK = 0, oCS = 0, oQ = 0, oOut = 0;
    int max_n = 0, max_p = 0, max_nk = 0, max_kpad = 0;
    bool any_sym = false, any_feat = false;
    for (int t = 0; t < T; ++t) {
        const HcTask &tk = tasks[i0 + t];
        HcMeta &M = metas[t];
        SHARP_REQUIRE(tk.n >= 3, "get_opt_hclust: need at least 3 observations");
        SHARP_REQUIRE(tk.n <= kHcMaxN, "get_opt_hclust: more than 16384 observations in one clustering task is not supported");
        SHARP_REQUIRE(tk.prm.hmethod >= 1 && tk.prm.hmethod <= 8, "get_opt_hclust: unknown agglomeration method");
        M.n = tk.n; M.p = tk.symmetric ? tk.n : tk.p; M.nld = static_cast<int>(rup(tk.n, 128));
        M.method = tk.prm.hmethod; M.symmetric = tk.symmetric ? 1 : 0; M.pad0 = 0;
        if (tk.prm.N_cluster > 0) {
            SHARP_REQUIRE(tk.prm.N_cluster >= 2, "The given N.cluster is less than 2, which is not suitable for clustering!");
            SHARP_REQUIRE(tk.prm.N_cluster <= tk.n - 1, "N.cluster must be smaller than the number of observations");
            M.kmin = M.kmax = tk.prm.N_cluster;
        } else {
            M.kmin = tk.prm.minN;
            M.kmax = std::min(tk.prm.maxN, tk.n - 1);
            SHARP_REQUIRE(M.kmin >= 2 && M.kmax >= M.kmin, "get_opt_hclust: empty range of cluster numbers (minN.cluster..maxN.cluster)");
        }
        M.nk = M.kmax - M.kmin + 1;
        M.kpad = static_cast<int>(rup(M.kmax, 16));
        M.oD = oD; oD += static_cast<long long>(M.nld) * M.nld;
        M.oD0 = oD0; if (M.symmetric) oD0 += static_cast<long long>(M.nld) * M.nld;
        M.oCr = oCr; oCr += static_cast<long long>(M.n) * M.p;
        M.oCt = oCt; oCt += rup(M.p, 16) * M.nld;
        M.oNrm = oN; oN += M.n;
        M.oM = oM; oM += M.n;
        M.oLab = oLab; oLab += static_cast<long long>(M.nk) * M.n;
        M.oH = M.oT = M.oG = oK; oK += static_cast<long long>(M.nld) * M.kpad;   // H: n x kpad; T, G: kpad x nld (transposed)
        M.oCSt = oCS; oCS += static_cast<long long>(M.p) * M.kpad;
        M.oQ = oQ; oQ += static_cast<long long>(M.kpad) * M.kpad;
        M.oOut = oOut; oOut += 2LL * M.nk;
        max_n = std::max(max_n, M.n); max_p = std::max(max_p, M.p); max_nk = std::max(max_nk, M.nk);
        max_kpad = std::max(max_kpad, M.kpad);
        any_sym |= tk.symmetric; any_feat |= !tk.symmetric;
    }
    J.oOut = oOut; J.oM = oM; J.oLab = oLab; J.max_n = max_n; J.max_p = max_p; J.max_nk = max_nk; J.max_kpad = max_kpad;
    SHARP_REQUIRE(max_nk > ml_min_levels() ||
                  stats_lds_bytes(max_n, std::max(max_kpad, 64)) <= ST_LDS_MAX,
                  "get_opt_hclust: this many observations with this many candidate cluster numbers does not fit the silhouette kernel "
                  "(LDS: 8 B per observation rounded up to a power of two + 36 B per candidate cluster)");
    { HostTimer ht("hc_workspace_alloc");
    W.D.ensure(oD); W0.S0.ensure(oD); W0.S1.ensure(oD); W.D0.ensure(std::max<long long>(oD0, 1)); W.Cr.ensure(oCr); W.Ct.ensure(oCt); W.nrm.ensure(oN);
    W.height.ensure(oM); W.ia.ensure(oM); W.ib.ensure(oM); W.lab.ensure(oLab);
    W.H.ensure(oK); W.T.ensure(oK); W.G.ensure(oK); W.CSt.ensure(oCS); W.Q.ensure(oQ); W.out.ensure(oOut); }
    { HostTimer ht("hc_workspace_alloc");
    W.meta.ensure(T); W.prep.ensure(T); W.gemm.ensure(5 * static_cast<size_t>(T)); W.status.ensure(T); }
    // feature tasks of at most 2048 observations: the distance GEMM also leaves every row's minimum per column tile (SHARP_HC_NN_GEMM=0: the
    // agglomeration's first round scans D)
    {
        bool parts = knobs().hc_nn_gemm;
#ifdef SHARP_LAB
        if (knobs().dist_i8 && max_p <= 8192) parts = false;   // (the sliced-integer GEMM of the lab build writes none)
#endif
        long long oNN = 0;
        if (parts) for (int t = 0; t < T; ++t) if (!metas[t].symmetric && metas[t].nld <= 2048) oNN += static_cast<long long>(metas[t].nld / 128) * metas[t].nld;
        if (oNN > 0) {
            W.nnp.ensure(oNN);
            oNN = 0;
            for (int t = 0; t < T; ++t) if (!metas[t].symmetric && metas[t].nld <= 2048) { metas[t].nn = W.nnp.p + oNN; oNN += static_cast<long long>(metas[t].nld / 128) * metas[t].nld; }
        }
    }
    W.meta.upload(metas.data(), T);

    // Every descriptor of the chunk goes up once; the device work is then enqueued per RANGE of tasks, each range on its
    // own stream: the agglomeration is a memory-latency/scatter-bound kernel and the correlation GEMM an MFMA-bound one,
    // so a range's GEMM, cutree and silhouette statistics run underneath the agglomeration of the other ranges.
    int NS = T >= 194 ? 2 : 1;  // two ranges of more than 96 tasks each (the one-launch agglomeration); 3 or more are slower than one
    {
        // the round-per-launch agglomeration synchronises with the host every few rounds: one range at a time
        // (small tasks -- the similarity matrices of wMetaC and of a per-block sMetaC, a few hundred meta-clusters -- have nothing to
        // spread over several workgroups: the per-round launches and the host's look every eight rounds only cost, 0.27 ms per SHARP() call)
        const bool split = knobs().hc_split >= 0 ? knobs().hc_split == 1 : (T <= kHcSplitMaxTasks && max_n >= kHcSplitMinObs);
        J.split = split;
        if (!knobs().hc_seq && ((!knobs().hc_mono && split) || max_n > HR_MAXN)) NS = 1;
    }
    if (knobs().hc_ranges > 0) NS = std::max(1, std::min(8, knobs().hc_ranges));
    if (J.pipe || J.one_range) NS = 1;          // the overlap comes from the neighbouring chunks / blocks
    if (J.max_nk > ml_min_levels()) NS = 1;   // many levels: one range
    NS = std::min(NS, T);
    std::vector<RowPrepTask> &prep = J.prep;
    prep.assign(T, RowPrepTask());
    for (int t = 0; t < T; ++t) {
        const HcTask &tk = tasks[i0 + t];
        const HcMeta &M = metas[t];
        prep[t] = RowPrepTask{tk.d_mat, tk.ld, M.n, M.p, M.nld, static_cast<int>(rup(M.p, 16)), M.symmetric, W.Cr.p + M.oCr,
                              W.Ct.p + M.oCt, W.nrm.p + M.oNrm, W.D.p + M.oD};
    }
    W.prep.upload(prep.data(), T);
    typedef ChunkJob::Range Range;
    J.NS = NS;
    std::vector<Range> &ranges = J.ranges;
    ranges.assign(NS, Range());
    std::vector<GemmTask> &g = J.g;
    g.clear();
    g.reserve(5 * static_cast<size_t>(T));
    for (int s = 0; s < NS; ++s) {
        Range &R = ranges[s];
        R.t0 = static_cast<int>(static_cast<long long>(T) * s / NS);
        R.t1 = static_cast<int>(static_cast<long long>(T) * (s + 1) / NS);
        R.any_sym = R.any_feat = false;
        for (int kind = 0; kind < 5; ++kind) {
            R.off[kind] = static_cast<int>(g.size());
            for (int t = R.t0; t < R.t1; ++t) {
                const HcMeta &M = metas[t];
                R.any_sym |= M.symmetric != 0; R.any_feat |= M.symmetric == 0;
                switch (kind) {
                    case 0:   // D = 1 - U U^T (feature tasks)
                        if (!M.symmetric) g.push_back(GemmTask{W.Ct.p + M.oCt, W.Ct.p + M.oCt, W.D.p + M.oD, M.n, M.n, M.p, M.nld, M.nld, M.nld, 1, 1, 1,
                                                               const_cast<double *>(M.nn), M.method == 8});
                        break;
                    case 1:   // finest-level sums on the MFMA:  CSt = Cr^T H
                        g.push_back(GemmTask{W.Cr.p + M.oCr, W.H.p + M.oH, W.CSt.p + M.oCSt, M.p, M.kpad, M.n, M.p, M.kpad, M.kpad, 0, 0, 0});
                        break;
                    case 2:   // G^T = CS C^T  (kpad x nld: the statistics kernel reads it with lanes along the cells)
                        g.push_back(GemmTask{W.CSt.p + M.oCSt, W.Ct.p + M.oCt, W.G.p + M.oG, M.kpad, M.n, M.p, M.kpad, M.nld, M.nld, 0, 0, 0});
                        break;
                    case 3:   // Q = CS CS^T
                        g.push_back(GemmTask{W.CSt.p + M.oCSt, W.CSt.p + M.oCSt, W.Q.p + M.oQ, M.kpad, M.kpad, M.p, M.kpad, M.kpad, M.kpad, 0, 0, 0});
                        break;
                    default:  // (symmetric) T^T = H^T D0  (kpad x nld)
                        if (M.symmetric) g.push_back(GemmTask{W.H.p + M.oH, W.D0.p + M.oD0, W.T.p + M.oT, M.kpad, M.n, M.n, M.kpad, M.nld, M.nld, 0, 0, 0});
                        break;
                }
            }
            R.cnt[kind] = static_cast<int>(g.size()) - R.off[kind];
        }
    }
#ifdef SHARP_LAB
    J.i8 = knobs().dist_i8 && max_p <= 8192;
    if (J.i8) {
        std::vector<DistI8Task> d8;
        size_t oSl = 0, oSc = 0;
        for (int t = 0; t < T; ++t) if (!metas[t].symmetric) { oSl += dist_i8_slice_bytes(metas[t].nld, metas[t].p); oSc += metas[t].nld; }
        W.sl.ensure(std::max<size_t>(oSl, 1)); W.slscale.ensure(std::max<size_t>(oSc, 1));
        oSl = 0; oSc = 0;
        for (int s = 0; s < NS; ++s) {
            ranges[s].off8 = static_cast<int>(d8.size());
            for (int t = ranges[s].t0; t < ranges[s].t1; ++t) {
                const HcMeta &M = metas[t];
                if (M.symmetric) continue;
                d8.push_back(DistI8Task{W.Cr.p + M.oCr, W.D.p + M.oD, W.sl.p + oSl, W.slscale.p + oSc, M.n, M.p, M.nld, (M.p + 31) / 32});
                oSl += dist_i8_slice_bytes(M.nld, M.p); oSc += M.nld;
            }
        }
        W.i8.ensure(std::max<size_t>(d8.size(), 1));
        if (!d8.empty()) W.i8.upload(d8.data(), d8.size());
    }
#endif
    // many candidate levels (> kMlMinLevels; SHARP_ML_MIN_LEVELS for tests): G and T of the whole chunk row-major (n x kpad)
    {
        J.ml = max_nk > ml_min_levels();
    }
    if (J.ml) {
        SHARP_REQUIRE(static_cast<size_t>(max_kpad) * 22 + 64 <= HR_LDS_CU, "get_opt_hclust: too many candidate cluster numbers (more than ~7400)");
        J.ml_off = static_cast<int>(g.size());
        for (int t = 0; t < T; ++t) {                      // G = C CS^T  (n x kpad)
            const HcMeta &M = metas[t];
            g.push_back(GemmTask{W.Ct.p + M.oCt, W.CSt.p + M.oCSt, W.G.p + M.oG, M.n, M.kpad, M.p, M.nld, M.kpad, M.kpad, 0, 0, 0});
        }
        J.ml_cnt = T;
        J.mlt_off = static_cast<int>(g.size());
        for (int t = 0; t < T; ++t) {                      // (symmetric) T = D0 H  (n x kpad)
            const HcMeta &M = metas[t];
            if (M.symmetric) g.push_back(GemmTask{W.D0.p + M.oD0, W.H.p + M.oH, W.T.p + M.oT, M.n, M.kpad, M.n, M.nld, M.kpad, M.kpad, 0, 0, 0});
        }
        J.mlt_cnt = static_cast<int>(g.size()) - J.mlt_off;
        W.gemm.ensure(g.size());
        J.mlmetas.assign(T, MlMeta());
        long long oS = 0, oMg = 0, oFin = 0;
        for (int t = 0; t < T; ++t) {
            const HcMeta &M = metas[t];
            J.mlmetas[t].oS = oS; oS += 2LL * M.n * M.nk;
            J.mlmetas[t].oMerge = oMg; oMg += M.nk;
            J.mlmetas[t].oFin = oFin; oFin += M.kmax;
        }
        W.mlmeta.ensure(T); W.mlS.ensure(oS); W.mlcn2m.ensure(oMg); W.mlB.ensure(oMg); W.mlr1.ensure(oMg); W.mlr2.ensure(oMg);
        W.mlcntF.ensure(oFin); W.mlcn2F.ensure(oFin); W.mltot2.ensure(T);
        W.mlmeta.upload(J.mlmetas.data(), T);
    }
    W.gemm.upload(g.data(), g.size());
    (void)any_sym; (void)any_feat;
    if (NS > 1) SHARP_HIP_CHECK(hipEventRecord(EV.in, chunk_stream));     // inputs and descriptors are ready
}

// Device work of a chunk, in three phases (`phases`: any contiguous set, in order): PH_DIST rows -> distance matrix, PH_AGGLO the
// agglomeration, PH_STATS cutree and the per-level statistics.  A pipelined chunk has one range on its slot's stream.
void enqueue_chunk(ChunkJob &J, int phases) {
    Ctx &c = ctx();
    Workspace &W = ws(J.slot);
    Workspace &W0 = ws(J.scratch_slot);
    PipeEvents &EV = pipe_events();
    const int NS = J.NS, max_n = J.max_n, max_p = J.max_p, max_nk = J.max_nk, max_kpad = J.max_kpad;
    hipStream_t main_stream = c.stream;
    hipStream_t chunk_stream = J.pipe ? c.aux_stream(J.slot) : main_stream;
    hipEvent_t ev_in = EV.in, *ev_out = EV.out;

    for (int s = 0; s < NS; ++s) {
        const ChunkJob::Range &R = J.ranges[s];
        const int Ts = R.t1 - R.t0;
        hipStream_t st = NS > 1 ? c.aux_stream(s) : chunk_stream;
        StreamScope scope(st);
        const HcMeta *dmeta = W.meta.p + R.t0;
        auto launch_sequential = [&](bool fallback_only) {
            KernelTimer tm("hclust_sequential");
            const int nal = (max_n + 1) & ~1;
            const size_t lds = (static_cast<size_t>(nal) * 8 + 32 * 8 + static_cast<size_t>(nal) * 4 * 3 + 32 * 4 + 8 + static_cast<size_t>(max_n) + 16 + 32 * 12 + 16 + 15) / 16 * 16;
            DevBuf<long long> dbg;
#ifdef SHARP_LAB                                                // (lab build, tools/build_variant.sh: phase ablation and per-phase cycle counts)
            const char *abl = lab_env("SHARP_HC_ABLATE");
            const char *tim = lab_env("SHARP_HC_TIMING");
            if (tim) { dbg.alloc(static_cast<size_t>(Ts) * 6); dbg.zero(); }
#else
            const char *abl = nullptr;
#endif
            if (max_n <= kHcLdsMaxN) {
                SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(hclust_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    static_cast<int>(lds)));
                hipLaunchKernelGGL(hclust_kernel<false>, dim3(Ts), dim3(HC_THREADS), lds, st, dmeta, W.D.p, W.ia.p, W.ib.p, W.height.p,
                                   abl ? atoi(abl) : 0, dbg.p, fallback_only ? W.status.p + R.t0 : nullptr, nullptr, 0LL);
            } else {                                            // state in global memory (W.seqstate is per slot, like D)
                W.seqstate.ensure(static_cast<size_t>(J.T) * lds);
                hipLaunchKernelGGL(hclust_kernel<true>, dim3(Ts), dim3(HC_THREADS), 0, st, dmeta, W.D.p, W.ia.p, W.ib.p, W.height.p,
                                   abl ? atoi(abl) : 0, dbg.p, fallback_only ? W.status.p + R.t0 : nullptr,
                                   W.seqstate.p + static_cast<size_t>(R.t0) * lds, static_cast<long long>(lds));
            }
            launch_check("hclust_kernel");
#ifdef SHARP_LAB
            if (tim) {
                std::vector<long long> h(static_cast<size_t>(Ts) * 6);
                dbg.download(h.data(), h.size());
                double acc[6] = {0, 0, 0, 0, 0, 0};
                for (int t = 0; t < Ts; ++t) for (int q = 0; q < 6; ++q) acc[q] += static_cast<double>(h[static_cast<size_t>(t) * 6 + q]);
                fprintf(stderr, "hclust phases T=%d n=%d, mean shader cycles per task: argmin %.0f | loads+LW+stores %.0f | nb reduce %.0f | "
                                "list+barrier %.0f | rescans %.0f | end barrier %.0f\n", Ts, max_n, acc[0] / Ts, acc[1] / Ts, acc[2] / Ts,
                        acc[3] / Ts, acc[4] / Ts, acc[5] / Ts);
            }
#endif
        };
        if (phases & PH_DIST) {
        if (NS > 1) SHARP_HIP_CHECK(hipStreamWaitEvent(st, ev_in, 0));
        // pipelined: this chunk's row preparation (HBM-bound, 0.75 ms alone) follows the previous chunk's at once, i.e. it runs beside the
        // previous chunk's distance GEMM (MFMA-bound) -- behind that GEMM it ran beside the previous chunk's agglomeration, took 2.5 ms
        // there and delayed this chunk's GEMM, which the NEXT agglomeration waits for, by as much (r03 timeline: the second
        // agglomeration of a cfg2 step started 1.9 ms after the first had ended);
        if (J.pipe && !J.first) SHARP_HIP_CHECK(hipStreamWaitEvent(st, knobs().hc_prep_early ? EV.prep[J.prev_slot] : EV.gemm[J.prev_slot], 0));
        // a3: rows -> centred/normalised (+ 1 - S for similarity input), then D = 1 - U U^T
        row_prep_batched(W.prep.p + R.t0, Ts, max_n, max_p, !R.any_sym);
        if (J.pipe) SHARP_HIP_CHECK(hipEventRecord(EV.prep[J.slot], st));
        // this chunk's distance GEMM starts when the previous chunk's has finished, i.e. together with the previous chunk's
        // agglomeration, and fills the CUs that one leaves free (it holds a whole CU per task)
        if (J.pipe && !J.first) SHARP_HIP_CHECK(hipStreamWaitEvent(st, EV.gemm[J.prev_slot], 0));
#ifdef SHARP_LAB
        if (R.cnt[0] && J.i8) dist_i8_batched(W.i8.p + R.off8, R.cnt[0], max_n);
        else
#endif
        if (R.cnt[0]) gemm_tn_f64_batched(W.gemm.p + R.off[0], R.cnt[0], max_n, max_n, "corr_dist_gemm", true, true);
        if (J.pipe) SHARP_HIP_CHECK(hipEventRecord(EV.gemm[J.slot], st));
        if (R.any_sym) {
            KernelTimer tm("copy_d");
            hipLaunchKernelGGL(copy_d_kernel, dim3(64, Ts), dim3(256), 0, st, dmeta, W.D.p, W.D0.p);
            launch_check("copy_d_kernel");
        }
        }
        if (phases & PH_AGGLO) {
        if (J.pipe && !J.first) SHARP_HIP_CHECK(hipStreamWaitEvent(st, EV.hc[J.prev_slot], 0));   // one agglomeration at a time (S0 / S1)
        // a4: agglomeration.  Reducible methods go through the bulk-synchronous kernel (streams whole rows between two scratch
        // matrices, D stays pristine); whatever it abandons (exact ties, centroid/median, n > 4096) is done by the
        // sequential NN-list kernel, which skips the tasks whose status is 0 -- no host round trip in between.
        {
            const bool use_rnn = !knobs().hc_seq;                 // SHARP_HC_SEQ=1 (cross-check): the sequential kernel only
            const bool gs = max_n > HR_MAXN;                    // state arrays in global memory (always round per launch)
            KernelTimer tm("hclust");
            if (use_rnn) {
                const int nal = (max_n + 3) & ~3;
                int npow2 = 1; while (npow2 < max_n - 1) npow2 <<= 1;
                const size_t state = (static_cast<size_t>(nal) * (16 + 4 + 4 + 2 * 7 + 1) + 16 * 4 + (1024 / 64 + 1) * 4 + 64 + 15) / 16 * 16;
                const size_t lds = std::max(state, static_cast<size_t>(npow2) * 16);
                const bool mono = knobs().hc_mono;                 // SHARP_HC_MONO=1 (cross-check): the whole agglomeration in one launch
                // Few tasks (one projection, the wMetaC / sMetaC similarity tasks, a cross-block sMetaC of thousands of meta-clusters):
                // one round per pair of launches, every task spread over several workgroups -- 25 tasks of 2000: 4.7 ms against
                // 10.3 ms in one launch, 50 tasks 6.5 against 11.3.  Many tasks (kHcSplitMaxTasks): one launch is faster (the chip is
                // then at its memory limit either way and the per-round launches only add their gaps).  SHARP_HC_SPLIT = 1 / 0 forces
                // the choice; it is made for the whole chunk (a range of a larger chunk stays one launch).
                const bool split = J.split || gs;
                if ((!mono || gs) && split) {
                    // workgroups per task in the rebuild launches: eight when there are tens of tasks (measured, 25 - 136 tasks of 2000);
                    // a lone big task (a per-block or cross-block sMetaC of thousands of clusters) gets up to a quarter of the chip
                    int wpt = std::max(1, std::min(8, (5 * c.num_cu / 2 + Ts - 1) / Ts));
                    if (Ts <= 8) wpt = std::max(wpt, std::min(64, c.num_cu / (4 * Ts)));
                    if (knobs().hc_wpt > 0) wpt = knobs().hc_wpt;
                    W0.img.ensure(static_cast<size_t>(Ts) * lds);
                    W0.remaining.ensure(1);
                    const int rem0 = Ts;
                    W0.remaining.upload(&rem0, 1);
                    auto ka = gs ? hclust_rnn_kernel<1024, 1, true> : hclust_rnn_kernel<1024, 1, false>;
                    auto kb = gs ? hclust_rnn_kernel<1024, 2, true> : hclust_rnn_kernel<1024, 2, false>;
                    // the rebuild launches stage the pair members' entries like MODE 0: whatever the CU has beyond the state (all of it
                    // when the state is global)
                    const size_t ldsa = gs ? 0 : lds;
                    const size_t ldsl = gs ? HR_LDS_CU : std::max(lds, HR_LDS_CU);
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsa)));
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsl)));
                    const int max_rounds = max_n + 8;               // every round merges at least one pair
                    // after `finish_at` rounds (about a quarter of the clusters left at the usual 10 % per round) the rest runs in ONE launch
                    const int finish_at = knobs().hc_finish_at;
                    auto kc = gs ? hclust_rnn_kernel<1024, 3, true> : hclust_rnn_kernel<1024, 3, false>;
                    if (finish_at >= 0) SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsl)));
                    for (int r = 0; r < max_rounds; ++r) {
                        if (J.mid_event && r == J.mid_round) { SHARP_HIP_CHECK(hipEventRecord(J.mid_event, st)); J.mid_recorded = true; }
                        if (finish_at >= 0 && r == finish_at) {
                            hipLaunchKernelGGL(kc, dim3(Ts), dim3(1024), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p, W.height.p,
                                               W.status.p + R.t0, W0.img.p, static_cast<long long>(lds), static_cast<int>(lds), r, W0.remaining.p, static_cast<int>(ldsl));
                            break;
                        }
                        hipLaunchKernelGGL(ka, dim3(Ts), dim3(1024), ldsa, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p, W.height.p,
                                           W.status.p + R.t0, W0.img.p, static_cast<long long>(lds), static_cast<int>(lds), r, W0.remaining.p, static_cast<int>(ldsa));
                        hipLaunchKernelGGL(kb, dim3(Ts, wpt), dim3(1024), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p, W.height.p,
                                           W.status.p + R.t0, W0.img.p, static_cast<long long>(lds), static_cast<int>(lds), r, W0.remaining.p, static_cast<int>(ldsl));
                        if ((r & 7) == 7) {                         // a finished task costs two empty workgroups per round: look now and then
                            int rem = 0;
                            W0.remaining.download(&rem, 1);
                            if (rem <= 0) break;
                        }
                    }
#ifdef SHARP_LAB
                } else if (Ts <= c.num_cu && max_n <= HL_MAXN && lab_env("SHARP_HC_LAZY") && lab_env("SHARP_HC_LAZY")[0] == '1') {
                    // SHARP_HC_LAZY=1 (an experiment kept for reference): one workgroup per task, rows rewritten only
                    // when their cluster merges (hclust_lazy.inc) -- half the bytes of hclust_rnn_kernel, same merges, but at four waves
                    // per CU (a 16 KB LDS row buffer each) it runs at a quarter of the bandwidth: 62 ms against 30 ms at cfg2
                    const size_t ldsz = hclust_lazy_lds(max_n);
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(hclust_lazy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(ldsz)));
                    int theta = 50;
                    if (const char *e = lab_env("SHARP_HC_LAZY_THETA")) theta = std::max(10, std::min(95, atoi(e)));
                    hipLaunchKernelGGL(hclust_lazy_kernel, dim3(Ts), dim3(HL_THREADS), ldsz, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p,
                                       W.height.p, W.status.p + R.t0, theta);
                } else if (Ts <= c.num_cu && max_n <= HT_MAXN && knobs().hc_tri) {
                    // one workgroup per CU on the upper triangle of the matrix (hclust_tri.inc): half the bytes of hclust_rnn_kernel
                    const size_t tstate = (static_cast<size_t>(nal) * (16 + 8 + 4 + 4 + 2 * 7 + 1) + 16 * 4 + (1024 / 64 + 1) * 4 + 64 + 15) / 16 * 16;
                    const size_t ldsl = std::max(std::max(tstate, static_cast<size_t>(npow2) * 16), HR_LDS_CU);
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(hclust_tri_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(ldsl)));
                    hipLaunchKernelGGL(hclust_tri_kernel, dim3(Ts), dim3(HT_THREADS), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p,
                                       W.height.p, W.status.p + R.t0, static_cast<int>(ldsl));
                } else if (Ts <= c.num_cu && knobs().hc_front > 0 && max_n <= 2400) {
                    // SHARP_HC_FRONT=c (an experiment, hclust_front.inc): the first c rounds without rewriting the matrix -- new rows and the
                    // survivors' tails appended beside the pristine D -- then one compaction into S0 and hclust_rnn_kernel's MODE 3 for the rest
                    const size_t ldsf = (static_cast<size_t>(max_n) * 3 / 2 + 8) * 34 + 96;
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(hclust_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(ldsf)));
                    W0.img.ensure(static_cast<size_t>(Ts) * lds);
                    W0.remaining.ensure(1);
                    hipLaunchKernelGGL(hclust_front_kernel, dim3(Ts), dim3(HF_THREADS), ldsf, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p, W.height.p,
                                       W.status.p + R.t0, W0.img.p, static_cast<long long>(lds), knobs().hc_front);
                    launch_check("hclust_front_kernel");
                    auto kc = hclust_rnn_kernel<1024, 3, false>;
                    const size_t ldsl = std::max(lds, HR_LDS_CU);
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsl)));
                    hipLaunchKernelGGL(kc, dim3(Ts), dim3(1024), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p, W.height.p,
                                       W.status.p + R.t0, W0.img.p, static_cast<long long>(lds), static_cast<int>(lds), 1, W0.remaining.p, static_cast<int>(ldsl));
#endif
                } else if (Ts <= c.num_cu && !knobs().hc_half) {
                    auto k0 = hclust_rnn_kernel<1024, 0>;
                    // one workgroup per CU: everything the CU has beyond the state stages the pair members' entries
                    const size_t ldsl = std::max(lds, HR_LDS_CU);
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(ldsl)));
                    hipLaunchKernelGGL(k0, dim3(Ts), dim3(1024), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p,
                                       W.height.p, W.status.p + R.t0, nullptr, 0LL, static_cast<int>(lds), 0, nullptr, static_cast<int>(ldsl));
                } else {
                    // (also SHARP_HC_HALF=1 with at most one task per CU: the eight-wave form then leaves half of every CU's registers and LDS to a
                    // workgroup of the next chunk's distance GEMM -- an experiment)
                    auto k0 = hclust_rnn_kernel<512, 0>;
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(lds)));
                    const size_t ldsl = std::max(lds, HR_LDS_CU / 2);      // two workgroups per CU
                    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(ldsl)));
                    hipLaunchKernelGGL(k0, dim3(Ts), dim3(512), ldsl, st, dmeta, W.D.p, W0.S0.p, W0.S1.p, W.ia.p, W.ib.p,
                                       W.height.p, W.status.p + R.t0, nullptr, 0LL, static_cast<int>(lds), 0, nullptr, static_cast<int>(ldsl));
                }
                launch_check("hclust_rnn_kernel");
            }
            // whatever the bulk-synchronous kernel abandoned (status != 0): the sequential kernel.  When it is only the fallback, a
            // pipelined chunk with a successor launches it with its statistics phase: its (normally idle) workgroups would otherwise
            // take CU slots from the successor's distance GEMM, which is on the critical path.
            if (!use_rnn || !(J.pipe && J.has_next)) launch_sequential(use_rnn);
            else J.seq_pending = true;
        }
        if (J.pipe) SHARP_HIP_CHECK(hipEventRecord(EV.hc[J.slot], st));
        if (J.pipe && knobs().step_marks)
            SHARP_HIP_CHECK(hipLaunchHostFunc(st, [](void *) { step_mark("    (device) an agglomeration ends"); }, nullptr));
        }
        if (!(phases & PH_STATS)) continue;
        // pipelined: the next chunk's distance GEMM (already enqueued) is on the critical path -- its agglomeration cannot start
        // before it -- and this tail is not: it waits for that GEMM and then runs beside the next agglomeration, whose stream has
        // the higher priority or is served first, on the CUs that one leaves free
        if (J.pipe && J.has_next) SHARP_HIP_CHECK(hipStreamWaitEvent(st, EV.gemm[J.next_slot], 0));
        if (J.pipe && J.next2_slot >= 0) SHARP_HIP_CHECK(hipStreamWaitEvent(st, EV.gemm[J.next2_slot], 0));
        if (J.seq_pending) { launch_sequential(true); J.seq_pending = false; }
        // a5a: labels for every candidate k
        {
            const size_t lds = static_cast<size_t>(max_n) * 8 + (HC_THREADS / 64 + 1) * 4 + 16;
            SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(cutree_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                static_cast<int>(lds)));
            KernelTimer tm("cutree");
            hipLaunchKernelGGL(cutree_kernel, dim3(Ts), dim3(HC_THREADS), lds, st, dmeta, W.ia.p, W.ib.p, W.lab.p);
            launch_check("cutree_kernel");
        }
        // the finest level's cluster sums: a dedicated kernel (SHARP_STATS_SUMS=0: the one-hot matrix and a skinny GEMM); the many-levels
        // form and clusterings of more than SS_KMAX clusters keep the GEMM
        const bool sums = knobs().stats_sums && !J.ml && max_kpad <= SS_KMAX;
        if (!sums || R.any_sym) {
            KernelTimer tm("onehot");
            hipLaunchKernelGGL(onehot_kernel, dim3(64, Ts), dim3(256), 0, st, dmeta, W.lab.p, W.H.p);
            launch_check("onehot_kernel");
        }
        if (sums) {
            KernelTimer tm("cluster_sums_gemm");
            const size_t lds = static_cast<size_t>(SS_WAVES) * max_kpad * 64 * 8;
            SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(cluster_sums_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            hipLaunchKernelGGL(cluster_sums_kernel, dim3((max_p + 63) / 64, Ts), dim3(64 * SS_WAVES), lds, st, dmeta, W.lab.p, W.Cr.p, W.CSt.p);
            launch_check("cluster_sums_kernel");
        } else if (R.cnt[1]) gemm_tn_f64_batched(W.gemm.p + R.off[1], R.cnt[1], max_p, max_kpad, "cluster_sums_gemm");
        if (J.ml) {
            // a5b, many levels.  (The chunk is one range here: NS = 1 whenever J.ml, see setup_chunk.)
            if (R.cnt[3]) gemm_tn_f64_batched(W.gemm.p + R.off[3], R.cnt[3], max_kpad, max_kpad, "cluster_gram_gemm");
            gemm_tn_f64_batched(W.gemm.p + J.ml_off, J.ml_cnt, max_n, max_kpad, "row_cluster_dot_gemm");
            if (J.mlt_cnt) gemm_tn_f64_batched(W.gemm.p + J.mlt_off, J.mlt_cnt, max_n, max_kpad, "dist_cluster_sums_gemm");
            // the merge that leads from level L + 1 to level L, in finest-cluster ids: a replay of the merge list on the host
            std::vector<int> h_ia(J.oM), h_ib(J.oM);
            W.ia.download(h_ia.data(), J.oM);
            W.ib.download(h_ib.data(), J.oM);
            long long tot_levels = 0;
            for (int t = 0; t < J.T; ++t) tot_levels += J.metas[t].nk;
            std::vector<int> h_r1(tot_levels, 0), h_r2(tot_levels, 0);
            for (int t = 0; t < J.T; ++t) {
                const HcMeta &M = J.metas[t];
                const int *ia = h_ia.data() + M.oM, *ib = h_ib.data() + M.oM;
                std::vector<int> fin(M.n, 0);                    // cell -> finest-cluster id if the cell is a representative at k = kmax
                std::vector<char> absorbed(M.n, 0);
                for (int q = 0; q < M.n - M.kmax; ++q) absorbed[ib[q] - 1] = 1;
                int f = 0;
                for (int i = 0; i < M.n; ++i) if (!absorbed[i]) fin[i] = f++;   // ids by first appearance = ascending representative
                for (int L = M.nk - 2; L >= 0; --L) {
                    const int q = M.n - 1 - (M.kmin + L);         // level k has the merges 0 .. n - k - 1 applied
                    h_r1[J.mlmetas[t].oMerge + L] = fin[ia[q] - 1];
                    h_r2[J.mlmetas[t].oMerge + L] = fin[ib[q] - 1];
                }
            }
            W.mlr1.upload(h_r1.data(), tot_levels);
            W.mlr2.upload(h_r2.data(), tot_levels);
            {
                KernelTimer tm("sil_ch_stats");
                const size_t lds_p = static_cast<size_t>(max_kpad) * 20 + 1024 * 8 + 64;
                SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ml_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_p)));
                hipLaunchKernelGGL(ml_prep_kernel, dim3(Ts), dim3(1024), lds_p, st, dmeta, W.mlmeta.p + R.t0, W.lab.p, W.Q.p, W.mlr1.p, W.mlr2.p,
                                   W.mlcn2m.p, W.mlB.p, W.mlcntF.p, W.mlcn2F.p, W.mltot2.p + R.t0);
                launch_check("ml_prep_kernel");
                const size_t per_wave = (static_cast<size_t>(max_kpad) * 22 + 15) & ~static_cast<size_t>(15);
                const size_t lds_c = per_wave * ML_WAVES;
                SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ml_cells_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_c)));
                int npow2 = 1; while (npow2 < max_n) npow2 <<= 1;
                const size_t lds_l = static_cast<size_t>(npow2) * 8 + ST_THREADS * 8;
                SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ml_level_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_l)));
                for (int t = R.t0; t < R.t1; ++t) {
                    const HcMeta &M = J.metas[t];
                    const int blocks = std::min((M.n + ML_WAVES - 1) / ML_WAVES, c.num_cu * std::max(1, static_cast<int>(HR_LDS_CU / std::max<size_t>(lds_c, 1))));
                    hipLaunchKernelGGL(ml_cells_kernel, dim3(blocks), dim3(64 * ML_WAVES), lds_c, st, W.meta.p, W.mlmeta.p, t, W.lab.p, W.T.p, W.G.p, W.nrm.p,
                                       W.mlr1.p, W.mlr2.p, W.mlcn2m.p, W.mlcntF.p, W.mlcn2F.p, W.mlS.p);
                    hipLaunchKernelGGL(ml_level_kernel, dim3(M.nk), dim3(ST_THREADS), lds_l, st, W.meta.p, W.mlmeta.p, t, W.mlS.p, W.mlB.p, W.out.p);
                }
                launch_check("ml_cells_kernel");
            }
        } else {
        if (R.cnt[2]) gemm_tn_f64_batched(W.gemm.p + R.off[2], R.cnt[2], max_kpad, max_n, "row_cluster_dot_gemm");
        if (R.cnt[3]) gemm_tn_f64_batched(W.gemm.p + R.off[3], R.cnt[3], max_kpad, max_kpad, "cluster_gram_gemm");
        if (R.cnt[4]) gemm_tn_f64_batched(W.gemm.p + R.off[4], R.cnt[4], max_kpad, max_n, "dist_cluster_sums_gemm");
        // a5b: silhouette medians + CH per level
        {
            const int kcap = std::max(J.max_kpad, 64);
            // at most 64 finest clusters (every call of the reference's defaults: maxN = 40): the walk over them fits one register per lane
            const bool lane_form = knobs().stats_lane && J.max_kpad <= 64;
            const size_t lds = lane_form ? stats_lane_lds_bytes(max_n, kcap) : stats_lds_bytes(max_n, kcap);
            const auto kern = lane_form ? stats_lane_kernel : stats_kernel;
            SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                static_cast<int>(lds)));
            KernelTimer tm("sil_ch_stats");
            const long long blocks = static_cast<long long>((Ts + 7) / 8) * 8 * max_nk;
            hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(ST_THREADS), lds, st, dmeta, W.lab.p, W.T.p,
                               W.G.p, W.Q.p, W.nrm.p, W.out.p, Ts, max_nk, kcap);
            launch_check("stats_kernel");
        }
        }   // !J.ml
        if (NS > 1) {
            SHARP_HIP_CHECK(hipEventRecord(ev_out[s], st));
            SHARP_HIP_CHECK(hipStreamWaitEvent(main_stream, ev_out[s], 0));
        }
    }
    if (J.pipe && (phases & PH_STATS)) SHARP_HIP_CHECK(hipEventRecord(EV.done[J.slot], chunk_stream));
    if (J.pipe && (phases & PH_STATS) && knobs().step_marks)
        SHARP_HIP_CHECK(hipLaunchHostFunc(chunk_stream, [](void *) { step_mark("    (device) a chunk's statistics end"); }, nullptr));
}

void finish_chunk(const std::vector<HcTask> &tasks, ChunkJob &J, bool want_v, std::vector<HcResult> &out) {
    Ctx &c = ctx();
    Workspace &W = ws(J.slot);
    const size_t i0 = J.i0;
    const int T = J.T;
    const long long oOut = J.oOut, oM = J.oM, oLab = J.oLab;
    const std::vector<HcMeta> &metas = J.metas;
    if (J.pipe) SHARP_HIP_CHECK(hipStreamWaitEvent(c.stream, pipe_events().done[J.slot], 0));
    W.h_out.ensure(std::max<long long>(oOut, 1)); W.h_height.ensure(std::max<long long>(oM, 1));
    double *h_out = W.h_out.p, *h_height = W.h_height.p;
    if (c.profiling) {      // which agglomeration kernel did the work (tests assert on it)
        int fallback = T;
        if (!knobs().hc_seq) {
            std::vector<int> st(T);
            W.status.download(st.data(), T);
            fallback = 0;
            for (int v : st) fallback += v != 0;
        }
        c.stats["host:hclust_tasks_bulk_synchronous"].launches += T - fallback;
        c.stats["host:hclust_tasks_sequential"].launches += fallback;
    }
    HostTimer ht_tail("hc_download_select");
    W.out.download(h_out, oOut);
    W.height.download(h_height, oM);
    // model selection on the host (a few dozen numbers per task)
    std::vector<int> chosen(T);
    std::vector<long long> poff(T);
    long long ptot = 0;
    for (int t = 0; t < T; ++t) { poff[t] = ptot; ptot += metas[t].n; }
    host_parallel_for(T, T >= 16 ? 16 : 1, [&](int t) {     // (each task writes its own result only)
        const HcTask &tk = tasks[i0 + t];
        const HcMeta &M = metas[t];
        HcResult &R = out[i0 + t];
        R.rc = 0; R.nk = M.nk;
        R.msil.assign(h_out + M.oOut, h_out + M.oOut + M.nk);
        R.CHind.assign(h_out + M.oOut + M.nk, h_out + M.oOut + 2 * M.nk);
        R.height.assign(h_height + M.oM, h_height + M.oM + M.n - 1);
        int oind = 1;
        if (tk.prm.N_cluster > 0) {
            R.branch = 0; R.maxsil = R.msil[0];
            R.CHind[0] = std::numeric_limits<double>::quiet_NaN();   // intCriteria value: filled by the single-task wrapper
        } else {
            select_level(tk.prm, M.n, M.kmin, M.nk, R.msil.data(), R.CHind.data(), R.height.data(), oind, R.branch, R.rc);
            R.maxsil = *std::max_element(R.msil.begin(), R.msil.end());
        }
        if (decision_log_on()) {
            double row[kDecisionCols];
            decision_row(tk.prm, M.n, M.kmin, M.nk, R.msil.data(), R.CHind.data(), R.height.data(), oind, R.branch, row);
            decision_log_add(row);
        }
        chosen[t] = oind - 1;
    });
    if (c.profiling) {      // which rule of R/get_opt_hclust.R:162-229 decided each task's level (bench.py reports the split per data set)
        static const char *const rule[3] = {"silhouette", "CH", "height"};
        for (int t = 0; t < T; ++t) {
            if (tasks[i0 + t].prm.N_cluster > 0) continue;
            const std::string key = std::string(tasks[i0 + t].symmetric ? "host:level_meta_by_" : "host:level_base_by_") + rule[std::min(2, std::max(0, out[i0 + t].branch))];
            c.stats[key].launches += 1;
        }
    }
    W.chosen.ensure(T); W.packoff.ensure(T); W.packed.ensure(ptot);
    W.chosen.upload(chosen.data(), T);
    W.packoff.upload(poff.data(), T);
    hipLaunchKernelGGL(pack_labels_kernel, dim3(8, T), dim3(256), 0, c.stream, W.meta.p, W.lab.p, W.chosen.p, W.packoff.p, W.packed.p);
    launch_check("pack_labels_kernel");
    W.h_packed.ensure(std::max<long long>(ptot, 1));
    int *h_packed = W.h_packed.p;
    W.packed.download(h_packed, ptot);
    std::vector<int> h_lab;
    if (want_v) { h_lab.resize(oLab); W.lab.download(h_lab.data(), oLab); }
    host_parallel_for(T, T >= 16 ? 16 : 1, [&](int t) {
        const HcMeta &M = metas[t];
        HcResult &R = out[i0 + t];
        R.f.assign(h_packed + poff[t], h_packed + poff[t] + M.n);
        R.optN = *std::max_element(R.f.begin(), R.f.end());
        if (want_v) R.v.assign(h_lab.begin() + M.oLab, h_lab.begin() + M.oLab + static_cast<long long>(M.nk) * M.n);
    });
}

}  // namespace

namespace { thread_local int g_batch_depth = 0; }         // > 0 while a batch's progress callback runs: a batch started from there is nested
namespace {
struct AfterAgglo { std::function<void(hipEvent_t)> fn; bool fired = false; };
AfterAgglo &after_agglo() { return per_slot<AfterAgglo>(); }
}  // namespace
void hc_release_workspaces() { release_workspaces_of_slot(); }
void hc_set_after_last_agglomeration(std::function<void(hipEvent_t)> fn) { after_agglo().fn = std::move(fn); after_agglo().fired = false; }
bool hc_after_last_agglomeration_fired() { return after_agglo().fired; }

void get_opt_hclust_batch(const std::vector<HcTask> &tasks, bool want_v, std::vector<HcResult> &out,
                          const std::function<void(size_t)> *progress, const std::function<hipEvent_t(size_t)> *prepare) {
    out.assign(tasks.size(), HcResult());
    if (tasks.empty()) return;
    ctx();
    if (g_batch_depth > 0) {
        // A batch started from another batch's progress callback (the per-fold wMetaC and the sMetaC of a finished block while later
        // chunks of the outer batch are in flight): one chunk on the current stream, with the third set of buffers and its OWN
        // agglomeration scratch -- the outer batch's chunks own sets 0 and 1 and share set 0's scratch.
        SHARP_REQUIRE(!progress && tasks.size() <= static_cast<size_t>(ctx().num_cu) * 4, "get_opt_hclust: nested batch too large");
        ChunkJob J;
        J.i0 = 0; J.i1 = tasks.size();
        J.slot = 2; J.scratch_slot = 2; J.one_range = true;
        setup_chunk(tasks, J);
        enqueue_chunk(J, PH_ALL);
        finish_chunk(tasks, J, want_v, out);
        return;
    }
    size_t free_b = 0, total_b = 0;
    { HostTimer ht("hc_mem_info"); SHARP_HIP_CHECK(hipMemGetInfo(&free_b, &total_b)); }
    const double budget = std::max(0.5 * static_cast<double>(free_b), 2.0e9);
    // More tasks than CUs: equal chunks of at most one task per CU.  The agglomeration kernel then runs its 1024-thread form
    // (one task per CU, twice the loads in flight) chunk after chunk -- as fast as all tasks at once with two per CU -- and the
    // workspaces (three distance-matrix-sized buffers per task) are sized for a chunk: 18 GB instead of 36 GB at cfg2, which
    // is what the first call pays in hipMalloc time.
    size_t max_tasks = tasks.size();
    {
        const size_t ncu = static_cast<size_t>(ctx().num_cu);
        if (knobs().hc_chunk > 0) max_tasks = static_cast<size_t>(knobs().hc_chunk);
        else if (tasks.size() > ncu) {
            size_t nch = (tasks.size() + ncu - 1) / ncu;
            // Many chunks (the blocks of a SHARP_unlimited call as one batch): what counts is the steady state, where chunk j + 1's distance
            // GEMM has only the CUs chunk j's agglomeration leaves free -- with 250 tasks per chunk that is 6 CUs and the two run one after
            // the other (36 ms per chunk); at most 3/4 of the CUs per chunk leaves the GEMM a quarter of the chip (18 ms per chunk of 179).
            // Two chunks (cfg2, cfg4's share) are better off as large as they can be: an agglomeration of 137 tasks takes as long as one of 188.
            if (nch >= 3) nch = (tasks.size() + ncu * 3 / 4 - 1) / (ncu * 3 / 4);
            max_tasks = (tasks.size() + nch - 1) / nch;
        }
    }
    std::vector<std::pair<size_t, size_t>> bounds;
    size_t i0 = 0;
    // SHARP_HC_FIRST_CHUNK=n: a first chunk of n tasks (its distance GEMM has the chip to itself and nothing to run beside: a short one
    // starts the first agglomeration earlier), the rest in equal chunks as above
    // (cfg3, 1250 tasks: 140 + 7 x 159 against 7 x 179: 162.2 against 164.4 ms per call, medians of 7 and 8 interleaved runs; a first chunk
    // of at most kHcSplitMaxTasks tasks takes the round-per-launch agglomeration, which wants the whole chip: 173 ms.)  Default with three
    // chunks or more: four fifths of a chunk, above that threshold; SHARP_HC_FIRST_CHUNK=-1: equal chunks.
    size_t first_tasks = 0;
    int first_knob = knobs().hc_first_chunk;
    if (first_knob == 0 && knobs().hc_chunk <= 0 && tasks.size() > 2 * max_tasks)
        first_knob = std::max(kHcSplitMaxTasks + 4, static_cast<int>(max_tasks) * 4 / 5);
    if (first_knob > 0 && static_cast<size_t>(first_knob) < max_tasks && tasks.size() > max_tasks) {
        first_tasks = static_cast<size_t>(first_knob);
        const size_t rest = tasks.size() - first_tasks, nch = (rest + max_tasks - 1) / max_tasks;
        max_tasks = (rest + nch - 1) / nch;
    }
    while (i0 < tasks.size()) {
        double bytes = 0;
        size_t i1 = i0;
        while (i1 < tasks.size() && i1 - i0 < (i0 == 0 && first_tasks ? first_tasks : max_tasks)) {
            const HcTask &t = tasks[i1];
            const double nld = static_cast<double>(rup(t.n, 128));
            const double p = t.symmetric ? t.n : t.p;
            const double b = 8.0 * (nld * nld * (t.symmetric ? 4 : 3) + 2 * nld * p + 4.0 * 64 * t.n) + 4.0 * 64 * t.n;   // D, two scratch matrices (+ D0)
            if (i1 > i0 && bytes + b > budget) break;
            bytes += b;
            ++i1;
        }
        bounds.push_back({i0, i1});
        i0 = i1;
    }
    // Several chunks: two in flight.  Chunk j + 1's row preparation and distance GEMM (MFMA-bound) run beside chunk j's
    // agglomeration (HBM-bound, one workgroup per task holding a whole CU: 188 of 256 CUs at cfg2) on the CUs that one leaves free,
    // chunk j's cutree / cluster sums / silhouette statistics beside chunk j + 1's agglomeration, and the host's level selection and
    // label download of chunk j under chunk j + 1's device work.  SHARP_HC_PIPE=0: one chunk at a time.
    const bool pipe = bounds.size() > 1 && knobs().hc_pipe;
    if (!pipe) {
        if (prepare) (void)(*prepare)(tasks.size());                         // (everything on the caller's stream: its order is the dependency)
        for (const auto &b : bounds) {
            ChunkJob J;
            J.i0 = b.first; J.i1 = b.second;
            { HostTimer ht("hc_single_setup"); setup_chunk(tasks, J); }
            { HostTimer ht("hc_single_enqueue"); enqueue_chunk(J, PH_ALL); }
            { HostTimer ht("hc_single_finish"); finish_chunk(tasks, J, want_v, out); }
        }
        return;
    }
    SHARP_HIP_CHECK(hipEventRecord(pipe_events().in, ctx().stream));        // the tasks' inputs are ready
    // Two chunks: two sets of buffers, host order  dist(0) agglo(0) | dist(1) stats(0) agglo(1) | fetch(0) fetch(1).
    // Three chunks or more: THREE sets in rotation (0, 1, 3), host order  ... | dist(j) stats(j-1) agglo(j) fetch(j-2) | ... : with two sets
    // chunk j's distance GEMM could only be enqueued once chunk j - 2's statistics -- which run beside chunk j - 1's agglomeration -- had
    // been fetched, i.e. when that agglomeration was over: GEMM and agglomeration then took turns (27.5 ms per chunk of 179 tasks).
    auto call_progress = [&](size_t done) {
        if (!progress) return;
        ++g_batch_depth;
        try { (*progress)(done); } catch (...) { --g_batch_depth; throw; }
        --g_batch_depth;
    };
    const size_t nb = bounds.size();
    const size_t R = nb >= 3 ? 3 : 2;
    static const int slot_of[3] = {0, 1, 3};
    ChunkJob jobs[3];
    size_t fetched = 0;                                                     // chunks 0 .. fetched - 1 are finished
    auto fetch_upto = [&](size_t upto) {                                    // finish chunks in order, report
        for (; fetched < upto; ++fetched) { finish_chunk(tasks, jobs[fetched % R], want_v, out); step_mark("chunk fetched", static_cast<int>(fetched)); }
    };
    auto start_chunk = [&](size_t j) {                                      // descriptors, uploads, rows and distance matrices of chunk j
        ChunkJob &J = jobs[j % R];
        J = ChunkJob();
        J.i0 = bounds[j].first; J.i1 = bounds[j].second;
        J.slot = slot_of[j % R]; J.prev_slot = slot_of[(j + R - 1) % R]; J.next_slot = slot_of[(j + 1) % R];
        J.pipe = true; J.first = j == 0; J.has_next = j + 1 < nb;
        if (prepare) J.input_ready = (*prepare)(J.i1);                    // (whatever prepare_ahead has not asked for already)
        setup_chunk(tasks, J);
        enqueue_chunk(J, PH_DIST);
    };
    const bool stats_last = R == 3;
    try {
    start_chunk(0);
    for (size_t j = 0; j < nb; ++j) {
        if (j >= 1 && !stats_last) enqueue_chunk(jobs[(j - 1) % R], PH_STATS);
        enqueue_chunk(jobs[j % R], PH_AGGLO);
        step_mark("agglomeration enqueued, chunk", static_cast<int>(j));
        // the inputs of the chunk after the next are asked for NOW, behind this agglomeration's launch: start_chunk(j + 1) below has to wait
        // for chunk j - 2's statistics first (its buffers), and inputs produced only then sat on the critical path in front of that chunk's GEMM
        if (prepare && j + 2 < nb) (void)(*prepare)(bounds[j + 2].second);
        if (j + 1 == nb && after_agglo().fn && g_batch_depth == 0) {        // the caller's side work behind the last agglomeration
            std::function<void(hipEvent_t)> fn = std::move(after_agglo().fn);
            after_agglo().fn = nullptr;
            fn(pipe_events().hc[jobs[j % R].slot]);
            after_agglo().fired = true;
        }
        // chunk j - 2's statistics ran beside chunk j - 1's agglomeration: fetched now, which also frees its set for chunk j + 1, whose
        // distance matrices are enqueued at once (they wait, on the device, for chunk j's GEMM); then the caller's work on finished
        // tasks, with chunk j's agglomeration and chunk j + 1's GEMM for the device to chew on
        if (R == 3 && j >= 2) fetch_upto(j - 1);
        if (j + 1 < nb) start_chunk(j + 1);
        // three sets: chunk j - 1's statistics go behind chunk j + 1's distance GEMM as well -- that GEMM decides when the next agglomeration
        // can start, the statistics only when a finished block's tail can; beside one agglomeration both crawled
        if (j >= 1 && stats_last) {
            ChunkJob &P = jobs[(j - 1) % R];
            P.next2_slot = j + 1 < nb ? jobs[(j + 1) % R].slot : -1;
            enqueue_chunk(P, PH_STATS);
        }
        if (fetched > 0) call_progress(bounds[fetched - 1].second);
    }
    enqueue_chunk(jobs[(nb - 1) % R], PH_STATS);
    if (nb >= 2) { fetch_upto(nb - 1); call_progress(bounds[nb - 2].second); }
    fetch_upto(nb);
    } catch (...) {
        (void)hipDeviceSynchronize();                                       // chunks are in flight on their own streams: nothing of this batch
        throw;                                                              // may still be running when the caller sees the error
    }
}

// ---- a batch whose distance matrices are built ahead of time (SHARP_unlimited: the next block's front under the current block's tail)
struct HcPrefetch {
    std::vector<HcTask> tasks;
    ChunkJob J;
    hipEvent_t ready = nullptr, agglo_done = nullptr;
    bool agglo_enqueued = false;
    ~HcPrefetch() { if (ready) (void)hipEventDestroy(ready); if (agglo_done) (void)hipEventDestroy(agglo_done); }
};

bool hc_prefetch_possible(const std::vector<HcTask> &tasks) {
    if (tasks.empty() || tasks.size() > static_cast<size_t>(ctx().num_cu) || knobs().hc_chunk > 0) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    double bytes = 0;
    for (const HcTask &t : tasks) {
        const double nld = static_cast<double>(rup(t.n, 128));
        bytes += 8.0 * (nld * nld * 4 + 2 * nld * (t.symmetric ? t.n : t.p));
    }
    return bytes < 0.25 * static_cast<double>(free_b);          // (always against the memory that is free NOW: the GPU may be shared)
}

std::shared_ptr<HcPrefetch> hc_prefetch_begin(std::vector<HcTask> tasks, int slot) {
    auto P = std::make_shared<HcPrefetch>();
    P->tasks = std::move(tasks);
    P->J.i0 = 0; P->J.i1 = P->tasks.size();
    P->J.slot = 4 + (slot & 1);
    P->J.one_range = true;
    setup_chunk(P->tasks, P->J);
    enqueue_chunk(P->J, PH_DIST);                           // on the stream that is current here (the caller's prefetch stream)
    SHARP_HIP_CHECK(hipEventCreateWithFlags(&P->ready, hipEventDisableTiming));
    SHARP_HIP_CHECK(hipEventRecord(P->ready, ctx().stream));
    return P;
}

hipEvent_t hc_prefetch_agglomerate(HcPrefetch &P) {
    SHARP_HIP_CHECK(hipStreamWaitEvent(ctx().stream, P.ready, 0));
    if (!P.agglo_done) SHARP_HIP_CHECK(hipEventCreateWithFlags(&P.agglo_done, hipEventDisableTiming));
    // round-per-launch form: the event sits behind the large rounds (the first eight move 3/4 of the bytes); the remaining small
    // rounds leave most of the chip idle, and the next block's front may as well start there
    P.J.mid_event = P.agglo_done;
    enqueue_chunk(P.J, PH_AGGLO);
    P.J.mid_event = nullptr;
    if (!P.J.mid_recorded) SHARP_HIP_CHECK(hipEventRecord(P.agglo_done, ctx().stream));
    P.agglo_enqueued = true;
    return P.agglo_done;
}

void hc_prefetch_stamp_block(HcPrefetch &P, int block) { for (HcTask &t : P.tasks) t.prm.dec_block = block; }   // (decision log: the call that uses the batch names its block)
void hc_prefetch_finish(HcPrefetch &P, bool want_v, std::vector<HcResult> &out) {
    out.assign(P.tasks.size(), HcResult());
    if (!P.agglo_enqueued) hc_prefetch_agglomerate(P);
    enqueue_chunk(P.J, PH_STATS);
    finish_chunk(P.tasks, P.J, want_v, out);
}

}  // namespace sharp

using namespace sharp;

// ---------------------------------------------------------------------------------------------
// C ABI: single-problem wrappers (the reference's exported functions take one matrix at a time)
// ---------------------------------------------------------------------------------------------
namespace {

bool host_is_symmetric(const double *mat, int n, int p) {   // isSymmetric(): square + all.equal(m, t(m), 100*eps)
    if (n != p) return false;
    long double num = 0, den = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            num += fabsl(static_cast<long double>(mat[static_cast<size_t>(i) * n + j]) - mat[static_cast<size_t>(j) * n + i]);
            den += fabsl(static_cast<long double>(mat[static_cast<size_t>(i) * n + j]));
        }
    const double tol = 100 * 2.220446049250313e-16;
    long double xy = num;
    if (den > 0 && den / (static_cast<long double>(n) * n) > tol) xy = num / den;
    return xy < tol;
}

// clusterCrit::intCriteria(., "Calinski_Harabasz") for the N.cluster-given branch (R/get_opt_hclust.R:105)
double host_ch_euclid(const double *y, int n, int p, const int *cl, int g) {
    std::vector<double> cen(static_cast<size_t>(g) * p, 0.0), all(p, 0.0);
    std::vector<int> cnt(g, 0);
    for (int i = 0; i < n; ++i) {
        const int c = cl[i] - 1; cnt[c]++;
        for (int k = 0; k < p; ++k) { cen[static_cast<size_t>(c) * p + k] += y[static_cast<size_t>(i) * p + k]; all[k] += y[static_cast<size_t>(i) * p + k]; }
    }
    for (int c = 0; c < g; ++c) for (int k = 0; k < p; ++k) cen[static_cast<size_t>(c) * p + k] /= cnt[c];
    for (int k = 0; k < p; ++k) all[k] /= n;
    double B = 0, W = 0;
    for (int c = 0; c < g; ++c) for (int k = 0; k < p; ++k) { const double d = cen[static_cast<size_t>(c) * p + k] - all[k]; B += cnt[c] * d * d; }
    for (int i = 0; i < n; ++i) for (int k = 0; k < p; ++k) { const double d = y[static_cast<size_t>(i) * p + k] - cen[static_cast<size_t>(cl[i] - 1) * p + k]; W += d * d; }
    return (B / (g - 1)) / (W / (n - g));
}

}  // namespace

extern "C" {

int sharp_get_opt_hclust(const double *mat, int n, int p, int hmethod, int N_cluster, int minN, int maxN, double sil_thre,
                         double height_Ntimes, int *f, int *v, double *msil, double *CHind, double *maxsil, double *height,
                         int *optN, int *nk, int *branch) {
    int warn = 0;
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(mat && f, "sharp_get_opt_hclust: null argument");
    SHARP_REQUIRE(n >= 3 && p >= 1, "sharp_get_opt_hclust: need n >= 3 and p >= 1");
    if (N_cluster != 0) {
        SHARP_REQUIRE(N_cluster >= 2, "The given N.cluster is less than 2, which is not suitable for clustering!");
    }
    HcTask t;
    t.n = n; t.p = p; t.ld = p;
    t.symmetric = host_is_symmetric(mat, n, p);
    t.prm.hmethod = hmethod > 0 ? hmethod : 1;
    t.prm.N_cluster = N_cluster;
    t.prm.minN = minN > 0 ? minN : 2;
    t.prm.maxN = maxN > 0 ? maxN : 40;
    t.prm.sil_thre = sil_thre;
    t.prm.height_Ntimes = height_Ntimes > 0 ? height_Ntimes : 2.0;
    DevBuf<double> dm(static_cast<size_t>(n) * p);
    dm.upload(mat, static_cast<size_t>(n) * p);
    t.d_mat = dm.p;
    std::vector<HcTask> tasks{t};
    std::vector<HcResult> res;
    get_opt_hclust_batch(tasks, v != nullptr, res);
    HcResult &R = res[0];
    warn = R.rc;
    std::copy(R.f.begin(), R.f.end(), f);
    if (v) std::copy(R.v.begin(), R.v.end(), v);
    if (N_cluster > 0) {
        // CH of the N.cluster branch is clusterCrit's Euclidean index on the (scaled) matrix
        std::vector<double> y(mat, mat + static_cast<size_t>(n) * p);
        if (!t.symmetric) {
            for (int i = 0; i < n; ++i) {
                double *r = y.data() + static_cast<size_t>(i) * p;
                long double s = 0; for (int k = 0; k < p; ++k) s += r[k];
                const double mean = static_cast<double>(s / p);
                long double ss = 0; for (int k = 0; k < p; ++k) { r[k] -= mean; ss += static_cast<long double>(r[k] * r[k]); }
                const double sd = std::sqrt(static_cast<double>(ss) / std::max(1, p - 1));
                for (int k = 0; k < p; ++k) r[k] /= sd;
            }
        }
        R.CHind[0] = host_ch_euclid(y.data(), n, p, R.f.data(), R.optN);
        R.optN = N_cluster;
    }
    if (msil) std::copy(R.msil.begin(), R.msil.end(), msil);
    if (CHind) std::copy(R.CHind.begin(), R.CHind.end(), CHind);
    if (height) std::copy(R.height.begin(), R.height.end(), height);
    if (maxsil) *maxsil = R.maxsil;
    if (optN) *optN = R.optN;
    if (nk) *nk = R.nk;
    if (branch) *branch = R.branch;
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return warn;
}

/* the decision log (hclust.hpp says what a row holds) */
int sharp_decision_log(int enable) {
    SHARP_API_BEGIN
    decision_log_set(enable != 0);
    SHARP_API_END
}
int sharp_last_decisions(double *rows, int cap_rows, int *n_rows) {
    SHARP_API_BEGIN
    SHARP_REQUIRE(n_rows && (rows || cap_rows == 0) && cap_rows >= 0, "sharp_last_decisions: null argument");
    *n_rows = decision_log_fetch(rows, cap_rows);
    SHARP_API_END
}

int sharp_getrowColor(const double *E, int n, int p, int hmethod, int indN_cluster, int minN, int maxN, double sil_thre,
                      double height_Ntimes, int *rowColor, double *maxsil) {
    std::vector<int> f(static_cast<size_t>(n > 0 ? n : 1));
    const int rc = sharp_get_opt_hclust(E, n, p, hmethod, indN_cluster, minN, maxN, sil_thre, height_Ntimes > 0 ? height_Ntimes : 1.0,
                                        f.data(), nullptr, nullptr, nullptr, maxsil, nullptr, nullptr, nullptr, nullptr);
    if (rc != SHARP_OK && rc != SHARP_WARN_RANGE) return rc;
    // colorL has 40 names; cluster j > 40 wraps onto colour ((j-1) %% 40) + 1 (R/getrowColor.R:59-68)
    for (int i = 0; i < n; ++i) rowColor[i] = f[i] > 40 ? ((f[i] - 1) % 40) + 1 : f[i];
    return rc;
}

}  // extern "C"
