// linalg.hip -- batched fp64 kernels: row preparation (scale/centre/normalise), tiled transpose and a
// TN GEMM on v_mfma_f64_16x16x4_f64.  Used for the correlation distance of get_opt_hclust
// (R/get_opt_hclust.R:71-72: n_t x n_t x p contraction per task) and for the per-cluster sums behind
// silhouette / CH (hclust.hip).
#include "linalg.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace sharp {

// Pointers read out of a descriptor in memory are generic (flat) to the compiler: flat loads count on lgkmcnt as well as
// vmcnt, so every LDS wait would also wait for the global prefetch.  Casting to the global address space gives global_load.
typedef __attribute__((address_space(1))) const double *gcdp;   // global const double *
typedef __attribute__((address_space(1))) double *gdp;          // global double *


typedef double v4f64 __attribute__((ext_vector_type(4)));

// Workgroup barrier for the LDS tiles: waits for this wave's LDS operations only (lgkmcnt(0)) -- a __syncthreads() would also wait
// (vmcnt(0)) for the global prefetch of the next k tile -- and orders memory accesses for the COMPILER too: the plain
// __builtin_amdgcn_s_barrier() is not a memory operation to LLVM, which may then move LDS loads and stores across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// GEMM: 64x64 tile per 256-thread workgroup, each wave a 32x32 quadrant = 2x2 MFMA 16x16 tiles.
// A/B operand lane map (f64 16x16x4): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// result register r of lane l is D[row = (l>>4) + 4r][col = l&15].
// LDS tiles are k-major [16][80]: a 32-lane half reads two k rows, 160 dwords apart = 32 banks apart,
// so ds_read_b64 is conflict-free.
// ---------------------------------------------------------------------------------------------
constexpr int GT = 64, GK = 16, GLD = 80;

__global__ __launch_bounds__(256) void gemm_tn_f64_kernel(const GemmTask *__restrict__ tasks) {
    const GemmTask t = tasks[blockIdx.z];
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    if (m0 >= t.M || n0 >= t.N) return;
    if (t.symmetric && n0 < m0) return;   // mirrored from the (n0, m0) tile
    __shared__ double As[GK][GLD];
    __shared__ double Bs[GK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int lrow = tid >> 4;          // 0..15  (k within the tile)
    const int lcol = (tid & 15) * 4;    // 0..60  (4 consecutive m / n)
    // the next k tile is fetched into registers while the MFMAs of the current one run (the same products in the same order as
    // without it: only the global-load latency, which every k tile used to pay in full, moves under the arithmetic)
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int k = k0 + lrow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mm = m0 + lcol + q, nn = n0 + lcol + q;
            ra[q] = (k < t.K && mm < t.M) ? ((gcdp)t.At)[static_cast<long long>(k) * t.lda + mm] : 0.0;
            rb[q] = (k < t.K && nn < t.N) ? ((gcdp)t.Bt)[static_cast<long long>(k) * t.ldb + nn] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < t.K; k0 += GK) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { As[lrow][lcol + q] = ra[q]; Bs[lrow][lcol + q] = rb[q]; }
        lds_barrier();   // lgkmcnt(0): the tile is in LDS
        if (k0 + GK < t.K) fetch(k0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[kr][wr + i * 16 + (lane & 15)];
                b[i] = Bs[kr][wc + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // raw barrier: a __syncthreads() here would also wait (vmcnt(0)) for the prefetch of the next tile
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr + i * 16 + (lane >> 4) + 4 * r;
                const int col = n0 + wc + j * 16 + (lane & 15);
                if (row < t.M && col < t.N) {
                    double v = acc[i][j][r];
                    if (t.epilogue == 1) {          // correlation distance
                        v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
                        v = 1.0 - v;
                        if (row == col) v = 0.0;
                    } else if (t.epilogue == 2) {   // correlation similarity, unit diagonal
                        v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
                        if (row == col) v = 1.0;
                    }
                    ((gdp)t.C)[static_cast<long long>(row) * t.ldc + col] = v;
                    if (t.symmetric && n0 > m0) ((gdp)t.C)[static_cast<long long>(col) * t.ldc + row] = v;
                }
            }
}

// Fast path for padded operands (the correlation-distance GEMM).  Measured: with 64x64 tiles the kernel is bound by the
// L2 -> CU operand traffic (8 flop per byte staged), not by the MFMA pipe, so this path uses a 128x128 tile per
// 512-thread workgroup (16 flop/byte): 8 waves as 2 (M) x 4 (N), each a 64x32 block = 4x2 MFMA tiles (64 accumulator
// VGPRs); no bounds checks in the K loop, 16-byte loads, next K tile fetched into registers while the MFMAs run.
constexpr int FT = 128, FLD = 144;   // 144 doubles per k row: two k rows of a half-wave land 32 banks apart

constexpr double NN_INF = 1.0e300;   // (hclust.hip: HC_INF)
template <int CTRL>
__device__ __forceinline__ double min_dpp_step(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return fmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false)));
}
// minimum over the 16 lanes of a DPP row (quad swaps, then the two mirrors): every lane of the row ends with it
__device__ __forceinline__ double min_row16(double x) {
    x = min_dpp_step<0xB1>(x);
    x = min_dpp_step<0x4E>(x);
    x = min_dpp_step<0x141>(x);
    x = min_dpp_step<0x140>(x);
    return x;
}

// Block -> (task, tile).  Only live tiles are launched (a workgroup that exits at once still costs a full dispatch: with
// the lower-triangle tiles of a symmetric product in the grid the kernel ran at 43 TF/s instead of 76, see
// tools/micro/mfma_f64_loop.hip), and the linear id is dealt so that the eight tasks of a group sit on the eight XCDs
// (workgroups go round-robin to XCDs): all tiles of a task then share one L2.
__global__ __launch_bounds__(512) void gemm_tn_f64_fast_kernel(const GemmTask *__restrict__ tasks, int count, int tiles_max, long long block0) {
    const long long B = block0 + blockIdx.x;          // (block0: the launch is one slice of the batch, see gemm_tn_f64_batched)
    const long long per_group = 8LL * tiles_max;
    const int zt = static_cast<int>(B / per_group) * 8 + static_cast<int>(B % 8);
    if (zt >= count) return;
    int L = static_cast<int>((B % per_group) / 8);
    const GemmTask t = tasks[zt];
    const int ntm = (t.M + FT - 1) / FT, ntn = (t.N + FT - 1) / FT;
    int ti, tj;
    if (t.symmetric) {            // upper triangle incl. diagonal, row by row
        ti = 0;
        while (ti < ntn && L >= ntn - ti) { L -= ntn - ti; ++ti; }
        if (ti >= ntn) return;
        tj = ti + L;
    } else {
        if (L >= ntm * ntn) return;
        ti = L / ntn; tj = L % ntn;
    }
    const int m0 = ti * FT, n0 = tj * FT;
    __shared__ double As[2][GK][FLD];     // two k tiles: the next one is stored while the current one is read (one barrier per tile)
    __shared__ double Bs[2][GK][FLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Wave -> block of the tile.  The second wave row takes its column blocks in REVERSE order: waves w and w + 4 share a SIMD, and in the
    // tiles where some waves have little or nothing to compute (below) the idle column blocks of the two rows then fall on different SIMDs.
    const int wm = wave >> 2, wn = wm ? 3 - (wave & 3) : (wave & 3);
    const int wr = wm * 64, wc = wn * 32;
    // Live sub-tiles of this wave's 4 x 2: those inside the matrix (2000 = 15 x 128 + 80: in the last tile row / column 3 of 8 sub-tile rows /
    // columns are padding) and, in a diagonal tile of a symmetric product, those not wholly below the diagonal (the mirrored stores of the
    // sub-tiles above it fill them).  Kept as a rectangle ni x nj from the block's corner; a wave with none only loads, stores and waits.
    // At cfg2 31 of a task's 136 tiles are such tiles and the busiest SIMD in them does 12 sub-tile products per k step instead of 16.
    int ni = (t.M - (m0 + wr) + 15) / 16, nj = (t.N - (n0 + wc) + 15) / 16;
    ni = ni < 0 ? 0 : (ni > 4 ? 4 : ni); nj = nj < 0 ? 0 : (nj > 2 ? 2 : nj);
    if (t.symmetric && ti == tj) { const int live = wn * 2 + nj - wm * 4; ni = live < ni ? (live < 0 ? 0 : live) : ni; }
    if (ni == 0 || nj == 0) { ni = 0; nj = 0; }
    v4f64 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int lrow = tid >> 5, lcol = (tid & 31) * 4;
    const int Kp = (t.K + GK - 1) / GK * GK;
    typedef double d2 __attribute__((ext_vector_type(2)));
    gcdp ap = (gcdp)t.At + static_cast<long long>(lrow) * t.lda + m0 + lcol;
    gcdp bp = (gcdp)t.Bt + static_cast<long long>(lrow) * t.ldb + n0 + lcol;
    typedef __attribute__((address_space(1))) const d2 *gd2p;
    d2 ra0 = *(gd2p)(ap), ra1 = *(gd2p)(ap + 2);
    d2 rb0 = *(gd2p)(bp), rb1 = *(gd2p)(bp + 2);
    As[0][lrow][lcol] = ra0.x; As[0][lrow][lcol + 1] = ra0.y; As[0][lrow][lcol + 2] = ra1.x; As[0][lrow][lcol + 3] = ra1.y;
    Bs[0][lrow][lcol] = rb0.x; Bs[0][lrow][lcol + 1] = rb0.y; Bs[0][lrow][lcol + 2] = rb1.x; Bs[0][lrow][lcol + 3] = rb1.y;
    lds_barrier();   // lgkmcnt(0): the first tile is in LDS
    // The second-dispatched half of the workgroup loses instruction arbitration to the older half at every k tile (priority, then age);
    // one static priority for it evens that out: 12.00 -> 11.77 ms at cfg2.  (s_setprio 1 / 0 around every MFMA block: 12.6 ms; all
    // 24 LDS reads of a k tile issued before its 32 MFMAs: 12.15; both: 13.7.)  The guard must be wave-uniform: s_setprio ignores EXEC.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // the k loop, one copy per shape of the wave's live block
    auto kloop = [&](auto NI_, auto NJ_) {
        constexpr int NI = decltype(NI_)::value, NJ = decltype(NJ_)::value;
        int buf = 0;
        for (int k0 = 0; k0 < Kp; k0 += GK) {
            // next tile into registers (the last iteration re-reads the final tile: unconditional loads keep the waits counted) ...
            const int kn = k0 + GK < Kp ? k0 + GK : k0;
            gcdp an = ap + static_cast<long long>(kn) * t.lda;
            gcdp bn = bp + static_cast<long long>(kn) * t.ldb;
            ra0 = *(gd2p)(an); ra1 = *(gd2p)(an + 2);
            rb0 = *(gd2p)(bn); rb1 = *(gd2p)(bn + 2);
            // ... while the MFMAs run on the current one
            if (NI > 0) {
#pragma unroll
                for (int kk = 0; kk < GK; kk += 4) {
                    const int kr = kk + (lane >> 4);
                    double a[4], b[2];
#pragma unroll
                    for (int i = 0; i < NI; ++i) a[i] = As[buf][kr][wr + i * 16 + (lane & 15)];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[j] = Bs[buf][kr][wc + j * 16 + (lane & 15)];
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
            // the other LDS tile was last read before the previous barrier: store the next tile there, one barrier per k tile
            buf ^= 1;
            As[buf][lrow][lcol] = ra0.x; As[buf][lrow][lcol + 1] = ra0.y; As[buf][lrow][lcol + 2] = ra1.x; As[buf][lrow][lcol + 3] = ra1.y;
            Bs[buf][lrow][lcol] = rb0.x; Bs[buf][lrow][lcol + 1] = rb0.y; Bs[buf][lrow][lcol + 2] = rb1.x; Bs[buf][lrow][lcol + 3] = rb1.y;
            lds_barrier();   // lgkmcnt(0); a raw barrier: __syncthreads() would add waits of its own
        }
    };
#define SHARP_GEMM_ROW(I) \
    if (nj == 2) kloop(std::integral_constant<int, I>(), std::integral_constant<int, 2>()); \
    else kloop(std::integral_constant<int, I>(), std::integral_constant<int, 1>());
    switch (ni) {
        case 0: kloop(std::integral_constant<int, 0>(), std::integral_constant<int, 0>()); break;
        case 1: SHARP_GEMM_ROW(1) break;
        case 2: SHARP_GEMM_ROW(2) break;
        case 3: SHARP_GEMM_ROW(3) break;
        default: SHARP_GEMM_ROW(4) break;
    }
#undef SHARP_GEMM_ROW
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (i >= ni || j >= nj) continue;
            // mirrored as well: every sub-tile of an off-diagonal tile, and in a diagonal tile those above the diagonal (their mirror images were not computed)
            const bool mirror = t.symmetric && (n0 > m0 || wn * 2 + j > wm * 4 + i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr + i * 16 + (lane >> 4) + 4 * r;
                const int col = n0 + wc + j * 16 + (lane & 15);
                if (row < t.M && col < t.N) {
                    double v = acc[i][j][r];
                    if (t.epilogue == 1) {
                        v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
                        v = 1.0 - v;
                        if (row == col) v = 0.0;
                    } else if (t.epilogue == 2) {
                        v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
                        if (row == col) v = 1.0;
                    }
                    ((gdp)t.C)[static_cast<long long>(row) * t.ldc + col] = v;
                    if (mirror) ((gdp)t.C)[static_cast<long long>(col) * t.ldc + row] = v;
                    acc[i][j][r] = v;
                }
            }
        }
    if (t.nn == nullptr) return;
    // Row minima of the tile (GemmTask::nn): every row's smallest off-diagonal entry over the tile's columns and, for the mirrored entries, every
    // column's smallest entry over the tile's rows.  The first round of the agglomeration takes a row's nearest neighbour from the one tile that
    // holds the row's minimum (1 KB) instead of scanning the row (16 KB): one of the ~10.7 passes over n^2 that hclust_rnn_kernel makes.  Values
    // only -- a (value, index, tie) reduction across lanes cost the GEMM 18 % (the K loop is only 30 tiles long), this one 3 %.
    double *rows_s = &As[0][0][0];          // [4 column blocks][128 rows]   (the k loop ended on a barrier: the tiles are free)
    double *cols_s = rows_s + 4 * FT;       // [2 row blocks][128 columns]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double b = NN_INF;
            const int row = m0 + wr + i * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wc + j * 16 + (lane & 15);
                if (i < ni && j < nj && row < t.M && col < t.N && row != col) {
                    const double v = acc[i][j][r];
                    b = fmin(b, t.nn_square ? v * v : v);
                }
            }
            b = min_row16(b);
            if ((lane & 15) == 0) rows_s[wn * FT + wr + i * 16 + (lane >> 4) + 4 * r] = b;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double b = NN_INF;
        const int col = n0 + wc + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool mirror = t.symmetric && (n0 > m0 || wn * 2 + j > wm * 4 + i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr + i * 16 + (lane >> 4) + 4 * r;
                if (i < ni && j < nj && mirror && row < t.M && col < t.N) {
                    const double v = acc[i][j][r];
                    b = fmin(b, t.nn_square ? v * v : v);
                }
            }
        }
        b = fmin(b, __shfl_xor(b, 16));
        b = fmin(b, __shfl_xor(b, 32));
        if (lane < 16) cols_s[wm * FT + wc + j * 16 + lane] = b;
    }
    __syncthreads();
    if (tid < FT) {
        double b = fmin(fmin(rows_s[tid], rows_s[FT + tid]), fmin(rows_s[2 * FT + tid], rows_s[3 * FT + tid]));
        if (ti == tj) b = fmin(b, fmin(cols_s[tid], cols_s[FT + tid]));
        if (m0 + tid < t.M) ((gdp)t.nn)[static_cast<long long>(tj) * t.ldc + m0 + tid] = b;
    } else if (tid < 2 * FT && ti != tj) {
        const int cc = tid - FT;
        if (n0 + cc < t.N) ((gdp)t.nn)[static_cast<long long>(ti) * t.ldc + n0 + cc] = fmin(cols_s[cc], cols_s[FT + cc]);
    }
}

void gemm_tn_f64_batched(const GemmTask *d_tasks, int count, int max_M, int max_N, const char *timer_name, bool fast, bool symmetric) {
    if (count <= 0 || max_M <= 0 || max_N <= 0) return;
    Ctx &c = ctx();
    KernelTimer tm(timer_name);
    for (int z0 = 0; z0 < count; z0 += 65535) {
        const int nz = std::min(65535, count - z0);
        if (fast) {
            const int ntm = (max_M + FT - 1) / FT, ntn = (max_N + FT - 1) / FT;
            const int tiles_max = symmetric ? ntn * (ntn + 1) / 2 : ntm * ntn;
            const long long blocks = static_cast<long long>((nz + 7) / 8) * 8 * tiles_max;
            // A block prepared ahead of time under another block's tail (Ctx::polite, SHARP_unlimited block after block): the batch goes out in slices
            // of a few rounds of workgroups: a freed half of a CU always fits the next 512-thread workgroup of this kernel, never the
            // 1024-thread workgroups of the tail's agglomeration kernels, which then waited -- whatever the stream priorities -- until
            // the whole batch had been dispatched (2.4 ms instead of 0.2 for the per-fold wMetaC trees).  Between slices the queue drains.
            long long slice = blocks;
            if (c.polite) {
                const int per_cu = knobs().gemm_slice;                     // SHARP_GEMM_SLICE: workgroups per CU in a slice (0: no slices)
                if (per_cu > 0) slice = static_cast<long long>(c.num_cu) * per_cu;
            }
            for (long long b0 = 0; b0 < blocks; b0 += slice)
                hipLaunchKernelGGL(gemm_tn_f64_fast_kernel, dim3(static_cast<unsigned>(std::min(slice, blocks - b0))), dim3(512), 0, c.stream,
                                   d_tasks + z0, nz, tiles_max, b0);
        }
        else
            hipLaunchKernelGGL(gemm_tn_f64_kernel, dim3((max_N + GT - 1) / GT, (max_M + GT - 1) / GT, nz), dim3(256), 0, c.stream,
                               d_tasks + z0);
        launch_check("gemm_tn_f64_kernel");
    }
}

// ---------------------------------------------------------------------------------------------
// Row preparation: one wave per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the row held as RP_MAXV values per lane (entries beyond p are zero and masked by the caller)
constexpr int RP_MAXV = 8;

// feature rows short enough to stay in registers: prepared AND transposed by row_prep_t_kernel
__device__ __forceinline__ bool fused_rows(const RowPrepTask &t) { return t.mode == 0 && t.p <= 64 * RP_MAXV; }

__global__ __launch_bounds__(256) void row_prep_kernel(const RowPrepTask *__restrict__ tasks, int skip_fused) {
    const RowPrepTask t = tasks[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= t.n) return;
    gcdp x = (gcdp)t.src + static_cast<long long>(row) * t.lds;
    gdp cr = (gdp)t.Cr + static_cast<long long>(row) * t.p;
    const int p = t.p;
    if (fused_rows(t) && skip_fused) return;              // row_prep_t_kernel's
    if (t.mode == 0 && p <= 64 * RP_MAXV) {
        // feature rows of a base-clustering task (p = reduced dimension): the row is read once and kept in registers;
        // the arithmetic and its order are those of the general path below
        double xv[RP_MAXV];
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) { const int q = lane + 64 * u; xv[u] = x[q < p ? q : 0]; }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) s += xv[u];
        const double mean = wave_sum(s) / p;
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { const double c = xv[u] - mean; ss += c * c; }
        const double sd = sqrt(wave_sum(ss) / static_cast<double>(p - 1 > 1 ? p - 1 : 1));
        double s2 = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { xv[u] = (xv[u] - mean) / sd; s2 += xv[u]; }
        const double mean2 = wave_sum(s2) / p;
        double n2 = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { xv[u] -= mean2; n2 += xv[u] * xv[u]; }
        const double nr = sqrt(wave_sum(n2));
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) cr[lane + 64 * u] = xv[u] / nr;
        if (lane == 0) ((gdp)t.nrm)[row] = 1.0;
        return;
    }
    double s = 0.0;
    for (int q = lane; q < p; q += 64) s += x[q];
    const double mean = wave_sum(s) / p;
    if (t.mode == 0) {
        // t(scale(t(mat))): centre, divide by sd with p-1  (R/get_opt_hclust.R:71)
        double ss = 0.0;
        for (int q = lane; q < p; q += 64) { const double c = x[q] - mean; ss += c * c; }
        const double sd = sqrt(wave_sum(ss) / static_cast<double>(p - 1 > 1 ? p - 1 : 1));
        // cor(t(mat)) centres again and normalises by the root sum of squares (R/get_opt_hclust.R:72)
        double s2 = 0.0;
        for (int q = lane; q < p; q += 64) s2 += (x[q] - mean) / sd;
        const double mean2 = wave_sum(s2) / p;
        double n2 = 0.0;
        for (int q = lane; q < p; q += 64) { const double c = (x[q] - mean) / sd - mean2; n2 += c * c; }
        const double nr = sqrt(wave_sum(n2));
        for (int q = lane; q < p; q += 64) cr[q] = ((x[q] - mean) / sd - mean2) / nr;
        if (lane == 0) ((gdp)t.nrm)[row] = 1.0;
    } else {
        // symmetric similarity: rows of S are the feature vectors of get_CH (centred, not scaled),
        // and d = as.dist(1 - S) takes the lower triangle (R/get_opt_hclust.R:66-69)
        double n2 = 0.0;
        for (int q = lane; q < p; q += 64) { const double c = x[q] - mean; n2 += c * c; cr[q] = c; }
        const double nr = sqrt(wave_sum(n2));
        if (lane == 0) ((gdp)t.nrm)[row] = nr;
        gdp drow = (gdp)t.D + static_cast<long long>(row) * t.nld;
        for (int q = lane; q < p; q += 64) {
            double d;
            if (q == row) d = 0.0;
            else if (q < row) d = 1.0 - x[q];                                         // S[row][q], row > q
            else d = 1.0 - ((gcdp)t.src)[static_cast<long long>(q) * t.lds + row];   // S[q][row], q > row
            drow[q] = d;
        }
    }
}

// Cr (n x p) -> Ct (p x nld), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const RowPrepTask *__restrict__ tasks, int skip_fused) {
    const RowPrepTask t = tasks[blockIdx.z];
    if (fused_rows(t) && skip_fused) return;
    __shared__ double tile[32][33];
    const int r0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
    if (r0 >= t.nld || q0 >= t.p_pad) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, q = q0 + tx;
        tile[j][tx] = (r < t.n && q < t.p) ? ((gcdp)t.Cr)[static_cast<long long>(r) * t.p + q] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int q = q0 + j, r = r0 + tx;
        if (q < t.p_pad && r < t.nld) ((gdp)t.Ct)[static_cast<long long>(q) * t.nld + r] = (r < t.n && q < t.p) ? tile[tx][j] : 0.0;
    }
}

// Row preparation and transposition in one pass for feature rows of at most 64 * RP_MAXV values (every base-clustering task): sixteen
// rows per workgroup, a row per wave at a time with the register path's arithmetic (same operations in the same order as
// row_prep_kernel), the normalised rows written to Cr and kept in LDS, from where Ct's columns go out as 128-byte runs.  One read of
// the source instead of a write and a re-read of Cr in between -- beside an HBM-bound agglomeration (pipelined chunks) the separate
// transpose crawled (8 ms for 0.55 ms of work).
constexpr int RT_ROWS = 16;
__global__ __launch_bounds__(256) void row_prep_t_kernel(const RowPrepTask *__restrict__ tasks) {
    const RowPrepTask t = tasks[blockIdx.y];
    const int r0 = blockIdx.x * RT_ROWS;
    if (!fused_rows(t) || r0 >= t.nld) return;
    extern __shared__ double rt_tile[];                   // [RT_ROWS][ldt]
    const int p = t.p, ldt = t.p_pad + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < RT_ROWS; i += 4) {
        const int row = r0 + i;
        double *trow = rt_tile + i * ldt;
        if (row >= t.n) {                                 // padding rows of Ct are zero
            for (int q = lane; q < t.p_pad; q += 64) trow[q] = 0.0;
            continue;
        }
        gcdp x = (gcdp)t.src + static_cast<long long>(row) * t.lds;
        gdp cr = (gdp)t.Cr + static_cast<long long>(row) * t.p;
        double xv[RP_MAXV];
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) { const int q = lane + 64 * u; xv[u] = x[q < p ? q : 0]; }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) s += xv[u];
        const double mean = wave_sum(s) / p;
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { const double c = xv[u] - mean; ss += c * c; }
        const double sd = sqrt(wave_sum(ss) / static_cast<double>(p - 1 > 1 ? p - 1 : 1));
        double s2 = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { xv[u] = (xv[u] - mean) / sd; s2 += xv[u]; }
        const double mean2 = wave_sum(s2) / p;
        double n2 = 0.0;
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) if (lane + 64 * u < p) { xv[u] -= mean2; n2 += xv[u] * xv[u]; }
        const double nr = sqrt(wave_sum(n2));
#pragma unroll
        for (int u = 0; u < RP_MAXV; ++u) {
            const int q = lane + 64 * u;
            if (q < p) { const double v = xv[u] / nr; cr[q] = v; trow[q] = v; }
            else if (q < t.p_pad) trow[q] = 0.0;          // padding rows of Ct
        }
        if (lane == 0) ((gdp)t.nrm)[row] = 1.0;
    }
    __syncthreads();
    const int i = threadIdx.x & (RT_ROWS - 1);
    for (int q = threadIdx.x / RT_ROWS; q < t.p_pad; q += 256 / RT_ROWS)
        ((gdp)t.Ct)[static_cast<long long>(q) * t.nld + r0 + i] = rt_tile[i * ldt + q];
}

// all_feature_rows: every task of the batch has feature rows of at most 64 * RP_MAXV values (the caller knows; the old pair of kernels is
// then not launched at all)
void row_prep_batched(const RowPrepTask *d_tasks, int count, int max_n, int max_p, bool all_feature_rows) {
    if (count <= 0 || max_n <= 0) return;
    Ctx &c = ctx();
    KernelTimer tm("row_prep");
    const bool fused = max_p <= 64 * RP_MAXV;
    for (int z0 = 0; z0 < count; z0 += 65535) {
        const int nz = std::min(65535, count - z0);
        const int nld_max = (max_n + 127) / 128 * 128;
        if (fused) {
            const size_t lds = static_cast<size_t>(RT_ROWS) * ((max_p + 15) / 16 * 16 + 1) * sizeof(double);
            SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_prep_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                static_cast<int>(lds)));
            hipLaunchKernelGGL(row_prep_t_kernel, dim3(nld_max / RT_ROWS, nz), dim3(256), lds, c.stream, d_tasks + z0);
            launch_check("row_prep_t_kernel");
            if (all_feature_rows) continue;
        }
        hipLaunchKernelGGL(row_prep_kernel, dim3((max_n + 3) / 4, nz), dim3(256), 0, c.stream, d_tasks + z0, fused ? 1 : 0);
        launch_check("row_prep_kernel");
        hipLaunchKernelGGL(transpose_kernel, dim3((max_p + 15 + 31) / 32, (nld_max + 31) / 32, nz), dim3(256), 0, c.stream, d_tasks + z0, fused ? 1 : 0);
        launch_check("transpose_kernel");
    }
}

}  // namespace sharp

using namespace sharp;

extern "C" {

/* Test entry for the fp64 MFMA GEMM kernels (tests/test_linalg_gpu.py): C (M x N, row-major) = sum_k At[k][.] Bt[k][.] for host
 * operands At (K x M) and Bt (K x N), row-major.  fast = 1 runs the 128 x 128-tile kernel on zero-padded copies (what the
 * correlation-distance stage feeds it), fast = 0 the generic 64 x 64 kernel on the operands as they are. */
int sharp_gemm_tn_f64(const double *At, const double *Bt, double *C, int M, int N, int K, int epilogue, int symmetric, int fast) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(At && Bt && C && M > 0 && N > 0 && K > 0, "sharp_gemm_tn_f64: bad arguments");
    SHARP_REQUIRE(!symmetric || (M == N), "sharp_gemm_tn_f64: symmetric needs M == N");
    const long long lda = fast ? (M + 127) / 128 * 128 : M, ldb = fast ? (N + 127) / 128 * 128 : N;
    const int Kp = fast ? (K + 15) / 16 * 16 : K;
    std::vector<double> ha(static_cast<size_t>(Kp) * lda, 0.0), hb(static_cast<size_t>(Kp) * ldb, 0.0);
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < M; ++i) ha[static_cast<size_t>(k) * lda + i] = At[static_cast<size_t>(k) * M + i];
        for (int j = 0; j < N; ++j) hb[static_cast<size_t>(k) * ldb + j] = Bt[static_cast<size_t>(k) * N + j];
    }
    DevBuf<double> dA(ha.size()), dB(hb.size()), dC(static_cast<size_t>(M) * N);
    dA.upload(ha.data(), ha.size());
    dB.upload(hb.data(), hb.size());
    dC.zero();
    GemmTask t{dA.p, symmetric ? dA.p : dB.p, dC.p, M, N, K, lda, symmetric ? lda : ldb, N, epilogue, symmetric, fast};
    DevBuf<GemmTask> dt(1);
    dt.upload(&t, 1);
    gemm_tn_f64_batched(dt.p, 1, M, N, "test_gemm", fast != 0, symmetric != 0);
    dC.download(C, static_cast<size_t>(M) * N);
    SHARP_API_END
}

}  // extern "C"

