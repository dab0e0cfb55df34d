// gemm_i8.hip -- the correlation-distance matrix D = 1 - U U^T (R/get_opt_hclust.R:66-74) on the INTEGER matrix cores.
// An alternative to the fp64 MFMA kernel (linalg.hip gemm_tn_f64_fast_kernel), OFF by default (SHARP_DIST_I8=1): correct, tested,
// and at the end of round 4 slower -- 0.9 ms for the digits + 5.1 ms for the products per chunk of 188 tasks of 2000 x 391 against
// 5.1 ms for the fp64 kernel (tools/bench_i8.py); DESIGN.md 5 "Round 4: the distance GEMM on the integer matrix cores" has the account.
//
// The rows of U are centred unit vectors in fp64.  Each row is scaled by a power of two so that its entries lie in (-1, 1) and cut
// into NS = 7 signed 7-bit digits (u = 2^e * sum_s a_s 2^(-6 - 7 s), |a_s| <= 64: every step of the cut is exact in fp64, the
// remainder is below 2^-49 of the row's largest entry).  U U^T is then the sum over digit pairs (s, t) of 2^(-12 - 7 (s + t)) A_s A_t^T,
// and an int8 x int8 product summed over p <= 8192 terms is EXACT in int32, so v_mfma_i32_32x32x32_i8 computes every A_s A_t^T without
// any rounding; pairs with s + t >= NS are below the truncation of the digits and dropped (28 products).  The pairs of one level
// l = s + t share an int32 accumulator (|sum| <= 7 * 8192 * 4096 < 2^31), and the seven accumulators of an element are folded in fp64 at
// the end, smallest first: one or two roundings of an otherwise exact dot product of the TRUNCATED rows.  Measured against a long-double
// product (tools/test_i8.py): max |D - ref| = 0.9 - 2.0e-14 with seven digits, where the fp64 MFMA kernel has 0.2 - 2.1e-15; an eighth
// digit (36 products) would bring it below the fp64 kernel's.  Rate: the i8 MFMA does 2048 op/clk/SIMD against 32 flop/clk/SIMD of
// v_mfma_f64_16x16x4_f64, so 28 products cost 28/64 of the fp64 MFMA time at equal clocks (tools/micro/mfma_i8_loop.hip: 3.3 Pop/s
// sustained, 2.7 ms for a chunk) -- what the kernel loses is the staging: 49 KB of digits per 32-deep k step and CU.
//
// Digit layout in HBM, per task: [k step of 32][slice][half h = 0, 1][row 0 .. nld)[16 bytes]  -- the A and the B fragment of the MFMA
// are both "row r, k = 16 h + 0..15" (lane l: r = l & 31, h = l >> 5; checked with integer data in the micro-benchmark), so one layout
// serves both operands of U U^T, a workgroup's panel of a k step is 28 contiguous runs, and a wave reads a fragment as 64 consecutive
// 16-byte pieces of LDS.
#include "linalg.hpp"
#include <functional>
#include <map>
#include <vector>

namespace sharp {

namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const v4i *gv4p;
typedef __attribute__((address_space(1))) const double *gcdp;
typedef __attribute__((address_space(1))) double *gdp;
typedef __attribute__((address_space(1))) unsigned int *gu32p;

constexpr int NS = kDistI8Slices;
constexpr int BM = 64, BN = 128;                       // workgroup tile: 2 x 4 waves, a 32 x 32 block each
constexpr int PIECES_A = NS * 2 * BM, PIECES_B = NS * 2 * BN, PIECES = PIECES_A + PIECES_B;   // 16-byte pieces of a k step's panel
constexpr int DI_THREADS = 512;
constexpr int RLOADS = (PIECES + 255) / 256;             // requests per panel of each of the four requesting waves
constexpr int PAD = RLOADS * 256;                       // pieces of a panel in LDS (the last round's spare slots included)

// ---- rows -> digits ---------------------------------------------------------------------------------------------------------------------
// One workgroup per 32 rows of a task: 8 threads per row, each 4 consecutive k of every k step.
__global__ __launch_bounds__(256) void slice_rows_kernel(const DistI8Task *__restrict__ tasks) {
    const DistI8Task t = tasks[blockIdx.y];
    const int r0 = blockIdx.x * 32;
    if (r0 >= t.nld) return;
    const int tid = threadIdx.x, row = r0 + (tid >> 3), j = tid & 7;
    const bool live = row < t.n;
    gcdp x = (gcdp)t.Cr + static_cast<long long>(live ? row : 0) * t.p;
    // the row's largest |u| (eight threads per row, then across them)
    double mx = 0.0;
    if (live) for (int k = j; k < t.p; k += 8) mx = fmax(mx, fabs(x[k]));
    mx = fmax(mx, __shfl_xor(mx, 1)); mx = fmax(mx, __shfl_xor(mx, 2)); mx = fmax(mx, __shfl_xor(mx, 4));
    int e = 0;
    if (mx > 0.0) { (void)frexp(mx, &e); }                // mx = f 2^e, f in [0.5, 1): |u| 2^-e < 1
    const double down = ldexp(1.0, -e);
    if (j == 0) ((gdp)t.scale)[row] = live && mx > 0.0 ? ldexp(1.0, e) : 0.0;
    const int h = j >> 2, off = (j & 3) * 4;
    for (int ks = 0; ks < t.ksteps; ++ks) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = ks * 32 + j * 4 + q;
            v[q] = (live && k < t.p) ? x[k] * down * 64.0 : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned int word = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a = rint(v[q]);
                v[q] = (v[q] - a) * 128.0;
                word |= (static_cast<unsigned int>(static_cast<int>(a)) & 0xffu) << (8 * q);
            }
            gu32p dst = (gu32p)(t.sl + ((static_cast<long long>(ks) * NS * 2 + s * 2 + h) * t.nld + row) * 16 + off);
            *dst = word;
        }
    }
}

// ---- digits -> D ------------------------------------------------------------------------------------------------------------------------
// Block -> (task, tile) as in gemm_tn_f64_fast_kernel: only tiles that reach the upper triangle are launched, the eight tasks of a
// group sit on the eight XCDs.
__global__ __launch_bounds__(DI_THREADS) void dist_i8_kernel(const DistI8Task *__restrict__ tasks, int count, const unsigned int *__restrict__ tile_list,
                                                              int tiles_max, long long block0) {
    const long long B = block0 + blockIdx.x;
    const long long per_group = 8LL * tiles_max;
    const int zt = static_cast<int>(B / per_group) * 8 + static_cast<int>(B % 8);
    if (zt >= count) return;
    const int L = static_cast<int>((B % per_group) / 8);
    const DistI8Task t = tasks[zt];
    const unsigned int tl = tile_list[L];                          // (row block, column block) of the L-th tile: super-tile order, see dist_i8_products
    const int m0 = static_cast<int>(tl >> 16) * BM, n0 = static_cast<int>(tl & 0xffffu) * BN;
    if (m0 >= t.n || n0 >= t.n) return;                            // (the list is made for the largest task of the launch)
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    v4i *panel = reinterpret_cast<v4i *>(sm);                     // three panels of PAD pieces
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int rowb = m0 + 32 * wm, colb = n0 + 32 * wn;
    // a wave's block is computed when it is not wholly below the diagonal and not wholly padding
    const bool live = colb + 31 >= rowb && rowb < t.n && colb < t.n;
    // Staging: the panel of a k step goes from HBM / L2 straight into LDS (global_load_lds_dwordx4: a wave-instruction fills 64 consecutive
    // 16-byte pieces, which is the panel's own order), three panels in rotation, the panel of step ks + 3 requested when step ks starts.
    // Piece q of the A part is (slice-half sh = q / BM, row q % BM), of the B part (q' / BN, q' % BN).  Only waves 0 .. 3 -- one per
    // SIMD -- make requests, RLOADS each per panel (the spare slots of the last round re-load the last piece into the panel's padding, so
    // that the waits count): the other wave of every SIMD starts its MFMAs right behind the barrier while the CU's one address unit works
    // through the 48 requests; with all eight waves requesting, no MFMA issued until they were through (2630 cycles per step against
    // 1830 for the MFMAs and the barrier alone).
    const bool requester = wave < 4;
    unsigned int src[RLOADS];                                     // (in 16-byte pieces: a task's digits are below 2^32 pieces)
#pragma unroll
    for (int u = 0; u < RLOADS; ++u) {
        int q = (tid & 255) + u * 256;
        if (q >= PIECES) q = PIECES - 1;
        unsigned int o;
        if (q < PIECES_A) o = static_cast<unsigned int>(q / BM) * t.nld + m0 + q % BM;
        else { const int q2 = q - PIECES_A; o = static_cast<unsigned int>(q2 / BN) * t.nld + n0 + q2 % BN; }
        src[u] = o;
    }
    const unsigned int kstride = static_cast<unsigned int>(NS) * 2 * t.nld;   // pieces per k step
    gv4p base = (gv4p)t.sl;
    typedef __attribute__((address_space(3))) v4i *lv4p;
    auto request = [&](int ks, int buf) __attribute__((always_inline)) {
        if (!requester) return;
#pragma unroll
        for (int u = 0; u < RLOADS; ++u)
            __builtin_amdgcn_global_load_lds(base + (src[u] + static_cast<unsigned int>(ks) * kstride), (lv4p)(panel + buf * PAD + u * 256 + wave * 64), 16, 0, 0);
    };
    auto landed = [&]() __attribute__((always_inline)) {          // all but the newest panel's requests of this wave have landed; its LDS reads too
        if (requester) __builtin_amdgcn_s_waitcnt(0x0070 | RLOADS); else __builtin_amdgcn_s_waitcnt(0x0070);
    };
    const int last = t.ksteps - 1;
    v16i acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[s][q] = 0;
    const int r = lane & 31, h = lane >> 5;
    const int offa = h * BM + 32 * wm + r, offb = PIECES_A + h * BN + 32 * wn + r;
    // The fragments of step ks + 1 are read from LDS while the MFMAs of step ks run (two register sets): behind one barrier per step all
    // eight waves read their fragments at the same time, 112 KB through a 128 B/clk LDS = 875 cycles in which no MFMA could start,
    // against 1792 cycles of MFMAs per step and SIMD.  Panel j is therefore read during step j - 1, its buffer is free again at
    // barrier(j), and three buffers carry the panel being read (ks + 1) and two in flight (ks + 2, ks + 3).
    v4i fa[2][NS], fb[2][NS];
    auto read_frags = [&](int set, int buf) __attribute__((always_inline)) {
        const v4i *pa = panel + buf * PAD + offa, *pb = panel + buf * PAD + offb;
#pragma unroll
        for (int s = 0; s < NS; ++s) { fa[set][s] = pa[s * 2 * BM]; fb[set][s] = pb[s * 2 * BN]; }
    };
    auto products = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int u = 0; u + s < NS; ++u) {
                acc[s + u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][s], fb[set][u], acc[s + u], 0, 0, 0);
            }
    };
    request(0, 0);
    request(last < 1 ? last : 1, 1);
    landed();                                                     // panel 0 is in LDS
    __builtin_amdgcn_s_barrier();
    request(last < 2 ? last : 2, 2);
    if (live) read_frags(0, 0);
    // one step: wait for panel ks + 1, ask for panel ks + 3 into the buffer panel ks has left, read the fragments of ks + 1, multiply ks
    // one step: wait for panel ks + 1, ask for panel ks + 3 into the buffer panel ks has left, read the fragments of ks + 1, multiply ks
    auto step = [&](int ks, int set, int bufn, int buff) __attribute__((always_inline)) {   // bufn: buffer of panel ks + 1, buff: of panel ks
        landed();                                                 // this wave's part of panel ks + 1 has landed (vmcnt), its fragment reads of panel ks are done (lgkmcnt)
        __builtin_amdgcn_s_barrier();
        request(ks + 3 < last ? ks + 3 : last, buff);
        if (live) {
            if (ks < last) read_frags(set ^ 1, bufn);
            products(set);
        }
    };
    int bf = 0;                                                   // buffer of panel ks
    int ks = 0;
    for (; ks + 2 <= t.ksteps; ks += 2) {                         // (two steps: the two register sets)
        int bn = bf == 2 ? 0 : bf + 1;
        step(ks, 0, bn, bf);
        bf = bn; bn = bf == 2 ? 0 : bf + 1;
        step(ks + 1, 1, bn, bf);
        bf = bn;
    }
    if (ks < t.ksteps) step(ks, 0, bf == 2 ? 0 : bf + 1, bf);
    __builtin_amdgcn_s_waitcnt(0x0f70);                           // the spare requests have landed before the panels are reused
    __builtin_amdgcn_s_barrier();
    // the row scales of the tile's rows and columns through LDS (the panels are free now): one load per thread instead of sixteen
    // dependent ones per lane in front of the stores
    double *sc = reinterpret_cast<double *>(sm);                  // BM row scales, then BN column scales
    if (tid < BM + BN) {
        const int g = tid < BM ? m0 + tid : n0 + tid - BM;
        sc[tid] = g < t.n ? ((gcdp)t.scale)[g] : 0.0;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    if (!live) return;
    // fold the levels (smallest first), undo the row scalings, 1 - clamp(.), zero diagonal.  Blocks above the diagonal are mirrored: the
    // block goes through the wave's own corner of LDS, transposed, so that the mirror image leaves in 256-byte rows like the block
    // itself (as 8-byte stores down a column of D the mirror took 0.7 of the kernel's 5.4 ms).
    const bool mirror = colb > rowb;
    const int col = colb + r;
    const double scol = sc[BM + 32 * wn + r] * 0.000244140625;     // 2^-12 and the column's power of two
    double *tr = reinterpret_cast<double *>(sm) + (BM + BN) + wave * (32 * 33);   // [column of the block][row of the block], 33 doubles per column
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int lr = (q & 3) + 8 * (q >> 2) + 4 * h;
        const int row = rowb + lr;
        double v = static_cast<double>(acc[NS - 1][q]);
#pragma unroll
        for (int l = NS - 2; l >= 0; --l) v = v * 0.0078125 + static_cast<double>(acc[l][q]);
        v *= sc[32 * wm + lr] * scol;
        v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);
        v = 1.0 - v;
        if (row == col) v = 0.0;
        if (mirror) tr[r * 33 + lr] = v;
        if (row < t.n && col < t.n) ((gdp)t.D)[static_cast<long long>(row) * t.nld + col] = v;
    }
    if (mirror) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the wave's own LDS writes (no other wave touches this corner)
        const int i = lane & 31, jj = lane >> 5;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int j = 2 * it + jj;
            const double v = tr[j * 33 + i];
            if (colb + j < t.n && rowb + i < t.n) ((gdp)t.D)[static_cast<long long>(colb + j) * t.nld + rowb + i] = v;
        }
    }
}
}  // namespace

size_t dist_i8_slice_bytes(int nld, int p) { return static_cast<size_t>((p + 31) / 32) * NS * 2 * nld * 16; }

void dist_i8_slices(const DistI8Task *d_tasks, int count, int max_n) {
    Ctx &c = ctx();
    const int max_nld = (max_n + 127) / 128 * 128;
    hipLaunchKernelGGL(slice_rows_kernel, dim3(max_nld / 32, count), dim3(256), 0, c.stream, d_tasks);
    launch_check("slice_rows_kernel");
}

// The tiles of the upper triangle in the order they are handed out.  The workgroups of one XCD work on one task at a time, 32 tiles
// side by side, and the digits of a task (5.4 MB at cfg2) do not fit its 4 MB L2: row block after row block every sweep streamed most
// of them from HBM again (86 MB per task, 16 GB per chunk of 188: the kernel ran at the HBM rate, 5.5 ms).  In SUPER-TILES of 8 x 4
// tiles (512 x 512 entries: 3 MB of digits for 32 tiles) the panels of the tiles in flight stay in L2.
struct TileList { DevBuf<unsigned int> d; int count = 0; };
struct TileLists { std::map<int, TileList> by_n; };     // one list per task size seen, never rewritten: launches on other streams may still read one
static TileLists &tile_lists() { return per_slot<TileLists>(); }
constexpr int SUP_R = 8, SUP_C = 4;

static const TileList &tile_list_for(int max_n) {
    TileLists &L = tile_lists();
    auto it = L.by_n.find(max_n);
    if (it != L.by_n.end()) return it->second;
    if (L.by_n.size() >= 64) { stream_sync(); SHARP_HIP_CHECK(hipDeviceSynchronize()); L.by_n.clear(); }   // (a run with that many task sizes: start over)
    const int ntm = (max_n + BM - 1) / BM, ntn = (max_n + BN - 1) / BN;
    std::vector<unsigned int> h;
    for (int I = 0; I * SUP_R < ntm; ++I)
        for (int J = 0; J * SUP_C < ntn; ++J)
            for (int ti = I * SUP_R; ti < std::min(ntm, (I + 1) * SUP_R); ++ti)
                for (int tj = J * SUP_C; tj < std::min(ntn, (J + 1) * SUP_C); ++tj)
                    if (tj * BN + BN - 1 >= ti * BM) h.push_back(static_cast<unsigned int>(ti) << 16 | static_cast<unsigned int>(tj));
    TileList &T = L.by_n[max_n];
    T.count = static_cast<int>(h.size());
    h.resize(h.size() + 8, 0u);                          // (room for the lab build's phase counters)
    T.d.ensure(h.size());
    T.d.upload(h.data(), h.size());
    stream_sync();                                       // (once per size: launches on OTHER streams will read this list)
    return T;
}

void dist_i8_products(const DistI8Task *d_tasks, int count, int max_n) {
    Ctx &c = ctx();
    const TileList &TL = tile_list_for(max_n);
    const int tiles_max = TL.count;
    const size_t lds = static_cast<size_t>(3) * PAD * 16;
    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dist_i8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const long long blocks = static_cast<long long>((count + 7) / 8) * 8 * tiles_max;
    long long slice = blocks;
    if (c.polite) {                                     // (as gemm_tn_f64_batched: a block prepared under another block's tail goes out in slices)
        const int per_cu = knobs().gemm_slice;
        if (per_cu > 0) slice = static_cast<long long>(c.num_cu) * per_cu;
    }
    for (long long b0 = 0; b0 < blocks; b0 += slice)
        hipLaunchKernelGGL(dist_i8_kernel, dim3(static_cast<unsigned>(std::min(slice, blocks - b0))), dim3(DI_THREADS), lds, c.stream, d_tasks, count,
                           TL.d.p, tiles_max, b0);
    launch_check("dist_i8_kernel");
}

void dist_i8_batched(const DistI8Task *d_tasks, int count, int max_n) {
    if (count <= 0 || max_n <= 0) return;
    KernelTimer tm("corr_dist_gemm");
    dist_i8_slices(d_tasks, count, max_n);
    dist_i8_products(d_tasks, count, max_n);
}

}  // namespace sharp

using namespace sharp;

extern "C" {
/* Test hook: D = 1 - U U^T of one matrix of unit rows (n x p row-major, host) through the sliced-integer path; D: n x n row-major. */
int sharp_dist_i8(const double *U, int n, int p, double *D) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(U && D && n >= 1 && p >= 1 && p <= 8192, "sharp_dist_i8: bad arguments");
    const int nld = (n + 127) / 128 * 128;
    DevBuf<double> dU(static_cast<size_t>(n) * p), dD(static_cast<size_t>(nld) * nld), dS(nld);
    DevBuf<signed char> dsl(dist_i8_slice_bytes(nld, p));
    dU.upload(U, static_cast<size_t>(n) * p);
    dD.zero();
    DistI8Task t{dU.p, dD.p, dsl.p, dS.p, n, p, nld, (p + 31) / 32};
    DevBuf<DistI8Task> dt(1);
    dt.upload(&t, 1);
    dist_i8_batched(dt.p, 1, n);
    std::vector<double> h(static_cast<size_t>(nld) * nld);
    dD.download(h.data(), h.size());
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) D[static_cast<size_t>(i) * n + j] = h[static_cast<size_t>(i) * nld + j];
    SHARP_API_END
}

/* Bench hook: `count` tasks of n unit rows x p (pseudo-random, generated on the host once and replicated), the digit kernel and the
 * product kernel timed apart with HIP events over `reps` launches; ms[0] = rows -> digits, ms[1] = digits -> D, ms[2] = the fp64 MFMA
 * kernel on the same tasks (k-major copies of the rows). */
int sharp_dist_i8_bench(int n, int p, int count, int reps, double *ms) {
    SHARP_API_BEGIN
    Ctx &c = ctx();
    SHARP_REQUIRE(n >= 1 && p >= 1 && p <= 8192 && count >= 1 && reps >= 1 && ms, "sharp_dist_i8_bench: bad arguments");
    const int nld = (n + 127) / 128 * 128, pp = (p + 15) / 16 * 16;
    std::vector<double> U(static_cast<size_t>(n) * p), Ut(static_cast<size_t>(pp) * nld, 0.0);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        double ss = 0, mean = 0;
        for (int k = 0; k < p; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; U[static_cast<size_t>(i) * p + k] = static_cast<double>(st >> 11) / 9007199254740992.0 - 0.5 + (i % 7) * 0.1 * (k % 5); mean += U[static_cast<size_t>(i) * p + k]; }
        mean /= p;
        for (int k = 0; k < p; ++k) { U[static_cast<size_t>(i) * p + k] -= mean; ss += U[static_cast<size_t>(i) * p + k] * U[static_cast<size_t>(i) * p + k]; }
        for (int k = 0; k < p; ++k) { U[static_cast<size_t>(i) * p + k] /= std::sqrt(ss); Ut[static_cast<size_t>(k) * nld + i] = U[static_cast<size_t>(i) * p + k]; }
    }
    DevBuf<double> dU(static_cast<size_t>(count) * n * p), dUt(static_cast<size_t>(count) * pp * nld), dD(static_cast<size_t>(count) * nld * nld), dS(static_cast<size_t>(count) * nld);
    const size_t slb = dist_i8_slice_bytes(nld, p);
    DevBuf<signed char> dsl(static_cast<size_t>(count) * slb);
    std::vector<DistI8Task> t8(count);
    std::vector<GemmTask> tg(count);
    for (int q = 0; q < count; ++q) {
        SHARP_HIP_CHECK(hipMemcpyAsync(dU.p + static_cast<size_t>(q) * n * p, U.data(), U.size() * 8, hipMemcpyHostToDevice, c.stream));
        SHARP_HIP_CHECK(hipMemcpyAsync(dUt.p + static_cast<size_t>(q) * pp * nld, Ut.data(), Ut.size() * 8, hipMemcpyHostToDevice, c.stream));
        t8[q] = DistI8Task{dU.p + static_cast<size_t>(q) * n * p, dD.p + static_cast<size_t>(q) * nld * nld, dsl.p + static_cast<size_t>(q) * slb, dS.p + static_cast<size_t>(q) * nld, n, p, nld, (p + 31) / 32};
        const double *a = dUt.p + static_cast<size_t>(q) * pp * nld;
        tg[q] = GemmTask{a, a, dD.p + static_cast<size_t>(q) * nld * nld, n, n, p, nld, nld, nld, 1, 1, 1};
    }
    stream_sync();
    DevBuf<DistI8Task> d8(count);
    DevBuf<GemmTask> dg(count);
    d8.upload(t8.data(), count);
    dg.upload(tg.data(), count);
    hipEvent_t e0, e1;
    SHARP_HIP_CHECK(hipEventCreate(&e0)); SHARP_HIP_CHECK(hipEventCreate(&e1));
    auto timed = [&](const std::function<void()> &fn) {
        fn();
        stream_sync();
        SHARP_HIP_CHECK(hipEventRecord(e0, c.stream));
        for (int r = 0; r < reps; ++r) fn();
        SHARP_HIP_CHECK(hipEventRecord(e1, c.stream));
        SHARP_HIP_CHECK(hipEventSynchronize(e1));
        float t = 0;
        SHARP_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        return static_cast<double>(t) / reps;
    };
    ms[0] = timed([&] { dist_i8_slices(d8.p, count, n); });
    ms[1] = timed([&] { dist_i8_products(d8.p, count, n); });
    ms[2] = timed([&] { gemm_tn_f64_batched(dg.p, count, n, n, "bench_f64", true, true); });
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    SHARP_API_END
}
}
