// rp.hip -- the sparse-ternary random-projection matmul:
//   E1 = t( 1/sqrt(p) * t(R_k) %*% log2(X+1) )      R/RPmat.R:32, R/SHARP.R:343-345,569-585
// for ALL K projectors in one pass over X (the reference re-reads and re-logs X once
// per k, R/SHARP.R:554-571).
//
// Kernel shape (HBM-bound, no MFMA: ~2*K*p/sqrt(m) flop per 4-byte element):
//   * one 512-thread workgroup streams one cell (a contiguous m-vector) at a time in
//     2048-gene chunks, one coalesced float4 per lane, next chunk prefetched in registers;
//   * zeros are skipped: non-zeros are compacted (wave ballot + one LDS atomic per wave)
//     into an LDS list, log2(1+x) is evaluated in fp64 only for them;
//   * each non-zero gene's packed row list (k*p+c, sign) is walked by a GW-lane group and
//     added into per-cell accumulators in LDS with ds_add_u64 on 44-bit fixed point --
//     integer accumulation is exact and order-independent, so the result is bit-reproducible
//     whatever the wave scheduling (fp64 atomics would not be);
//   * the epilogue converts to fp64, applies sqrt(s)/sqrt(p) and writes the K*p row of E.
#include "projector.hpp"
#include "upload.hpp"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>

namespace sharp {

constexpr int RP_THREADS = 512;
constexpr int RP_NV = 2;                             // float4 loads per lane per step
constexpr int RP_STEP = RP_THREADS * 4 * RP_NV;      // genes per step (4096)
constexpr int RP_CAP = 2048;                         // non-zero slots per scatter batch

struct __attribute__((aligned(16))) NzSlot {
    uint32_t gene;   // gene index
    uint32_t xbits;  // raw fp32 bits
    long long fix;   // fixed-point log2(1+x), filled by pass 2
};

template <bool VEC>
__device__ __forceinline__ float4 rp_load4(const float *col, int g0, int gend) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {   // streamed once: non-temporal, keep the projector lists resident in L2 instead
        if (g0 + 3 < gend) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(col + g0));
            return make_float4(t.x, t.y, t.z, t.w);
        }
    }
    if (g0 < gend) v.x = col[g0];
    if (g0 + 1 < gend) v.y = col[g0 + 1];
    if (g0 + 2 < gend) v.z = col[g0 + 2];
    if (g0 + 3 < gend) v.w = col[g0 + 3];
    return v;
}

struct RpStepVals { float4 v[RP_NV]; };

// one step = genes [s*step_len, min(m,(s+1)*step_len)); lane t owns quads t, t+512, ... inside it
template <bool VEC>
__device__ __forceinline__ RpStepVals rp_load_step(const float *col, int s, int step_len, int m, int tid) {
    RpStepVals r;
    const int g_begin = s * step_len;
    const int g_end = min(m, g_begin + step_len);
#pragma unroll
    for (int j = 0; j < RP_NV; ++j) r.v[j] = rp_load4<VEC>(col, g_begin + 4 * (tid + j * RP_THREADS), g_end);
    return r;
}

// Wave-autonomous structure: each wave compacts, logs and scatters the genes it loaded itself, through
// its own slice of the LDS slot list, so the only workgroup barriers are the two around the per-cell
// epilogue.  Per step a wave has ONE dependent L2 round trip: the row-list loads of all its non-zero
// genes are issued together right after compaction (fixed-stride segments need no pointer lookup) and
// the fp64 log2 pass runs while they are in flight.
template <int GW, int SLOTS, bool VEC>
__global__ __launch_bounds__(RP_THREADS, 4) void rp_scatter_kernel(
    const float *__restrict__ X, int m, int n, long long ld, int log_flag, double fix_scale, double inv_fix,
    double val, double out_scale, const uint16_t *__restrict__ ent, const uint2 *__restrict__ ovf_slot,
    const uint2 *__restrict__ ovf_info, int novf, int ncomp, int neg_base, double *__restrict__ E, long long ldE, int comp0,
    int nsteps, int step_len, const int *__restrict__ row_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVE = RP_THREADS / 64;
    constexpr int WCAP = RP_CAP / NWAVE;                                               // slots per wave
    constexpr int NG = 64 / GW, SPAN = SLOTS * GW;
    constexpr uint32_t ACC0 = RP_CAP * sizeof(NzSlot);                                 // LDS address of accumulator 0
    constexpr int U = 16;                                                              // row lists in flight per group
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem + RP_CAP * sizeof(NzSlot));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int grp = lane / GW, lg = lane % GW;
    NzSlot *list = reinterpret_cast<NzSlot *>(smem) + (tid >> 6) * WCAP;               // this wave's slots
    const int nacc = neg_base > 0 ? 2 * neg_base : ncomp;   // (dual accumulators: projector.hpp; the codes then carry no sign bit)
    for (int c = tid; c < nacc; c += RP_THREADS) acc[c] = 0ull;
    __syncthreads();

    long long cell = blockIdx.x;
    if (cell >= n) return;
    RpStepVals cur = rp_load_step<VEC>(X + cell * ld, 0, step_len, m, tid);
    while (cell < n) {
        const float *col = X + cell * ld;
        const long long next_cell = cell + gridDim.x;
        for (int st = 0; st < nsteps; ++st) {
            // prefetch the next step (or the next cell's first step): its HBM latency hides under this step
            RpStepVals nxt;
            if (st + 1 < nsteps) nxt = rp_load_step<VEC>(col, st + 1, step_len, m, tid);
            else if (next_cell < n) nxt = rp_load_step<VEC>(X + next_cell * ld, 0, step_len, m, tid);
            else {
#pragma unroll
                for (int j = 0; j < RP_NV; ++j) nxt.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int g_begin = st * step_len;
            float vals[4 * RP_NV];
#pragma unroll
            for (int j = 0; j < RP_NV; ++j) {
                vals[4 * j] = cur.v[j].x; vals[4 * j + 1] = cur.v[j].y; vals[4 * j + 2] = cur.v[j].z; vals[4 * j + 3] = cur.v[j].w;
            }
            unsigned pending = 0;
#pragma unroll
            for (int q = 0; q < 4 * RP_NV; ++q) pending |= (vals[q] != 0.0f) ? (1u << q) : 0u;

            bool more = true;
            while (more) {   // one trip unless this wave's share holds more than WCAP non-zeros (dense data)
                // ---- pass 1: wave-local compaction of pending non-zeros (ballot prefix, no atomics)
                int wn = 0;
#pragma unroll
                for (int q = 0; q < 4 * RP_NV; ++q) {
                    const unsigned long long mk = __ballot((pending >> q) & 1u);
                    if ((pending >> q) & 1u) {
                        const int pos = wn + __popcll(mk & ((1ull << lane) - 1ull));
                        if (pos < WCAP) {
                            list[pos].gene = static_cast<uint32_t>(g_begin + 4 * (tid + (q >> 2) * RP_THREADS) + (q & 3));
                            list[pos].xbits = __float_as_uint(vals[q]);
                            pending &= ~(1u << q);
                        }
                    }
                    wn += __popcll(mk);
                }
                const int nnz = wn < WCAP ? wn : WCAP;
                more = wn > WCAP;
                __builtin_amdgcn_wave_barrier();
                for (int e0 = 0; e0 < nnz; e0 += U * NG) {
                    // ---- pass 3a: issue the row-list loads of up to U genes per group (8 bytes per lane each)
                    RowWord<SLOTS> cd[U];
                    {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + grp + u * NG;
                            cd[u].x = 0xffffffffu;
                            if (e < nnz) cd[u] = load_row_word<SLOTS>(ent + static_cast<size_t>(list[e].gene) * SPAN + SLOTS * lg);
                        }
                    }
                    // ---- pass 2 (under the loads): one lane per non-zero, fixed-point log2(1+x) in fp64
                    for (int e = e0 + lane; e < nnz && e < e0 + U * NG; e += 64) {
                        const float x = __uint_as_float(list[e].xbits);
                        const double L = log_flag == 2 ? log10(1.0 + static_cast<double>(x)) : (log_flag ? log2(1.0 + static_cast<double>(x)) : static_cast<double>(x));   // 2: SHARP_unlimited2's log10
                        list[e].fix = __double2ll_rn(L * fix_scale);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // ---- pass 3b: LDS atomics
                    {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + grp + u * NG;
                            if (e < nnz) {
                                const long long fix = list[e].fix;
                                const bool sgn = neg_base == 0;     // signed codes: bit 0 of every slot of a negative lane
                                scatter_row_word<ACC0, SLOTS, true>(cd[u], static_cast<unsigned long long>((sgn && (cd[u].x & kCodeNeg)) ? -fix : fix));
                                // rare: a full segment may continue in overflow segments
                                const uint32_t firstcode = __shfl(cd[u].x, lane & ~(GW - 1));
                                if ((firstcode & kCodeMore) != 0u && novf > 0) {
                                    const uint32_t g = list[e].gene;
                                    const uint2 oi = ovf_slot[g];
                                    {
                                        for (uint32_t sg = 0; sg < oi.y; ++sg) {
                                            const RowWord<SLOTS> c = load_row_word<SLOTS>(ent + (static_cast<size_t>(oi.x) + sg) * SPAN + SLOTS * lg);
                                            scatter_row_word<ACC0, SLOTS, true>(c, static_cast<unsigned long long>((sgn && (c.x & kCodeNeg)) ? -fix : fix));
                                        }
                                    }
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            cur = nxt;
        }
        __syncthreads();   // every wave's atomics for this cell have landed
        // ---- epilogue: E[cell, comp0 + c] = (1/sqrt(p)) * (sqrt(s) * sum), clear accumulators
        double *erow = E + (row_map ? static_cast<long long>(row_map[cell]) : cell) * ldE + comp0;
        for (int c = tid; c < ncomp; c += RP_THREADS) {
            long long a = static_cast<long long>(acc[c]);
            acc[c] = 0ull;
            if (neg_base > 0) { a -= static_cast<long long>(acc[neg_base + c]); acc[neg_base + c] = 0ull; }
            erow[c] = out_scale * (val * (static_cast<double>(a) * inv_fix));
        }
        __syncthreads();
        cell = next_cell;
    }
}

// Raw (no log) mode needs a data-dependent fixed-point scale: max |x| over the block.
// (fp64 blocks: the maximum rounded UP to fp32 -- the scale only needs a bound)
template <typename T>
__global__ void absmax_kernel(const T *__restrict__ X, int m, int n, long long ld, unsigned int *out) {
    float mx = 0.f;
    const long long total = static_cast<long long>(m) * n;
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long c = i / m;
        const int g = static_cast<int>(i - c * m);
        const T v = X[c * ld + g];
        const T a = v < T(0) ? -v : v;
        float f = __uint_as_float(0x7f800000u);                                      // NaN counts as non-finite (+inf)
        if (a == a) {
            f = static_cast<float>(a);
            if (static_cast<T>(f) < a) f = __uint_as_float(__float_as_uint(f) + 1u);   // the next fp32 up
        }
        mx = fmaxf(mx, f);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(mx));  // non-negative floats order as uints
}

// max |x| of an fp64 block as a double (+inf if any value is NaN or infinite): non-negative doubles order as 64-bit integers
__global__ void absmax64_kernel(const double *__restrict__ X, int m, int n, long long ld, unsigned long long *out) {
    double mx = 0.0;
    const long long total = static_cast<long long>(m) * n;
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long c = i / m;
        const double v = X[c * ld + (i - c * m)];
        const double a = v == v ? fabs(v) : __longlong_as_double(0x7ff0000000000000ll);
        mx = fmax(mx, a);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, static_cast<unsigned long long>(__double_as_longlong(mx)));
}

// max |x| of a resident block (rounded up to fp32; +inf for a NaN): the raw mode's fixed-point scale, and what an fp64 block handed
// over already resident needs checked before its log-mode scale is chosen (dev64_ref)
float device_absmax(XRef X, int m, long long n, long long ld) {
    Ctx &c = ctx();
    DevBuf<unsigned int> mx(1);
    mx.zero();
    {
        KernelTimer t("rp_absmax");
        const int nn = static_cast<int>(n);
        if (X.f64) hipLaunchKernelGGL(absmax_kernel<double>, dim3(c.num_cu * 4), dim3(256), 0, c.stream, X.d64(), m, nn, ld, mx.p);
        else hipLaunchKernelGGL(absmax_kernel<float>, dim3(c.num_cu * 4), dim3(256), 0, c.stream, X.f32(), m, nn, ld, mx.p);
        launch_check("absmax_kernel");
    }
    unsigned int bits = 0;
    mx.download(&bits, 1);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
// An fp64 block that is ALREADY resident (the *_dev64 entry points: TPM / CPM-like doubles, which fp32 would perturb): 16-byte aligned,
// even leading dimension, finite; one pass over it finds max |x| -- beyond FLT_MAX the log-mode accumulation keeps 41 fixed-point
// bits instead of 44 (upload.hpp does the same while it copies a host block).
XRef dev64_ref(const double *dX, int m, long long n, long long ld) {
    SHARP_REQUIRE(dX, "No expression data is provided!");
    SHARP_REQUIRE((reinterpret_cast<uintptr_t>(dX) & 15u) == 0 && ld % 2 == 0 && ld >= m,
                  "an fp64 device block must be 16-byte aligned with an even leading dimension >= the number of genes");
    SHARP_REQUIRE(n >= 1 && n < (1LL << 31), "block size out of range");
    XRef r(dX);
    Ctx &c = ctx();
    DevBuf<unsigned long long> mx(1);
    mx.zero();
    {
        KernelTimer t("rp_absmax");
        hipLaunchKernelGGL(absmax64_kernel, dim3(c.num_cu * 4), dim3(256), 0, c.stream, dX, m, static_cast<int>(n), ld, mx.p);
        launch_check("absmax64_kernel");
    }
    unsigned long long bits = 0;
    mx.download(&bits, 1);
    double f;
    memcpy(&f, &bits, 8);
    SHARP_REQUIRE(std::isfinite(f), "SHARP: the expression block holds NA / NaN / Inf");
    if (f > 3.4028234663852886e38) r.log_fix_bits = 41;
    return r;
}

template <int GW, int SLOTS, bool VEC>
static void launch_rp(const ProjectorGroup &g, const Projector &pr, const float *dX, int m, int n, long long ld,
                      int log_flag, int fix_bits, double *dE, long long ldE, const int *row_map) {
    Ctx &c = ctx();
    const size_t lds = RP_CAP * sizeof(NzSlot) + static_cast<size_t>(g.acc_slots() + kDumpSlots) * 8;   // + the pad codes' dump accumulators
    const int nsteps = (m + RP_STEP - 1) / RP_STEP;
    const int step_len = ((m + nsteps - 1) / nsteps + 3) / 4 * 4;
    auto kern = rp_scatter_kernel<GW, SLOTS, VEC>;
    {   // scatter_row_word<ACC0, ...>: the dynamic LDS block must start at LDS address 0 (no static LDS in the kernel)
        hipFuncAttributes fa;
        SHARP_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
        SHARP_REQUIRE(fa.sharedSizeBytes == 0, "rp_scatter_kernel: static LDS in front of the dynamic block");
    }
    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(lds)));
    int per_cu = 1;   // resident blocks per CU (LDS- and VGPR-limited): size the persistent grid to exactly that
    SHARP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), RP_THREADS, lds));
    per_cu = std::max(1, std::min(per_cu, 4));
    long long blocks = std::min<long long>(n, static_cast<long long>(c.num_cu) * per_cu);
    const double fix_scale = std::ldexp(1.0, fix_bits), inv_fix = std::ldexp(1.0, -fix_bits);
    const double out_scale = 1.0 / std::sqrt(static_cast<double>(pr.p));
    KernelTimer t("rp_stage");
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(RP_THREADS), lds, c.stream, dX, m, n, ld, log_flag,
                       fix_scale, inv_fix, pr.val, out_scale, g.ent.p, g.ovf_slot.p, g.ovf_info.p, g.novf, g.ncomp, g.neg_base, dE, ldE, g.k0 * pr.p,
                       nsteps, step_len, row_map);
    launch_check("rp_scatter_kernel");
}

#ifdef SHARP_LAB
void project_dev_split(const Projector &pr, const ProjectorGroup &g, XRef dX, int m, int n, long long ld, int log_flag,
                       int fix_bits, double *dE, long long ldE, const int *d_row_map, unsigned ahead_token);   // tools/lab/rp2.hip
#else
// (the two-kernel form -- compaction + apply, tools/lab/rp2.hip -- and its compaction ahead of the projector build are lab code: the
// producer / consumer kernel of rp3.hip takes every input they took)
unsigned rp_compact_ahead(XRef, int, int, long long, int) { return 0; }
void rp_compact_ahead_drop() {}
void rp_trim() {}
#endif

void project_dev(const Projector &pr, XRef X, int m, int n, long long ld, int log_flag, double *dE, long long ldE,
                 const int *d_row_map, unsigned ahead_token) {
    struct DropAhead { ~DropAhead() { rp_compact_ahead_drop(); } } drop_ahead;   // lists compacted ahead serve this call or none
    const float *dX = X.f32();
    SHARP_REQUIRE(m == pr.m, "project: gene count differs from the projector's");
    SHARP_REQUIRE(ld >= m, "project: leading dimension smaller than m");
    SHARP_REQUIRE(ldE >= static_cast<long long>(pr.K) * pr.p, "project: ldE smaller than K*p");
    if (n <= 0) return;
    ctx();
    {
        // a projector of density 1/sqrt(m) >= 1/4 is not sparse: the dense form on the MFMA (also SHARP_RP_KERNEL=dense, for cross-checks)
        if (knobs().rp_kernel == 2 || (knobs().rp_kernel == 0 && m <= 16)) {
            project_dev_dense(pr, X, m, n, ld, log_flag, dE, ldE, d_row_map);
            return;
        }
    }
    int fix_bits = std::min(RP_FIX_BITS, X.log_fix_bits);
    if (!log_flag) {
        const float f = device_absmax(X, m, n, ld);
        SHARP_REQUIRE(std::isfinite(f), "project: non-finite expression value");
        int e = 0;
        std::frexp(static_cast<double>(f), &e);          // |x| < 2^e
        fix_bits = std::min(52, 62 - 12 - std::max(e, 0)); // up to 2^11 terms + sign
    }
    const bool vec = (ld % (X.f64 ? 2 : 4) == 0) && ((reinterpret_cast<uintptr_t>(X.p) & 15u) == 0);
    SHARP_REQUIRE(!X.f64 || (vec && m >= 8 && m <= (1 << 20)), "project: an fp64 block must be 16-byte aligned with an even leading dimension");
    for (const auto &g : pr.groups) {
        const int gw = g.gw;
#define SHARP_RP_CASE(GWV, SL)                                                                          \
    if (vec) launch_rp<GWV, SL, true>(g, pr, dX, m, n, ld, log_flag, fix_bits, dE, ldE, d_row_map);      \
    else launch_rp<GWV, SL, false>(g, pr, dX, m, n, ld, log_flag, fix_bits, dE, ldE, d_row_map)
        const bool fused = knobs().rp_kernel == 1;   // SHARP_RP_KERNEL=fused: the single-kernel form (always used for unaligned X)
        if (rp_pc_eligible(X, m, ld)) {   // the producer / consumer kernel (rp3.hip): the default wherever X can be read with 16-byte loads
            project_dev_pc(pr, g, X, m, n, ld, log_flag, fix_bits, dE, ldE, d_row_map);
            continue;
        }
#ifdef SHARP_LAB
        if (vec && m >= 8 && m <= (1 << 20) && (X.f64 || !fused)) {   // (20-bit gene index in the compacted entries)
            project_dev_split(pr, g, X, m, n, ld, log_flag, fix_bits, dE, ldE, d_row_map, ahead_token);
            continue;
        }
#else
        (void)fused; (void)ahead_token;
#endif
        SHARP_REQUIRE(!X.f64, "project: an fp64 block needs at least 17 genes (the producer / consumer kernel) or the dense form");
        if (gw == 16 && g.slots == 4) { SHARP_RP_CASE(16, 4); }
        else if (gw == 16) { SHARP_RP_CASE(16, 2); }
        else if (gw == 8) { SHARP_RP_CASE(8, 4); }
        else { SHARP_RP_CASE(4, 4); }
#undef SHARP_RP_CASE
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic expression generator (bench/tests only): value = f(seed, gene, cell) with an
// integer-only value path, bit-identical to oracle_synth_value() on the CPU.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t synth_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline int synth_cluster(uint32_t seed, uint32_t cell, int G) {
    const uint32_t h = synth_mix(cell * 0x9e3779b9u + seed * 0x85ebca6bu + 0x1234567u);
    unsigned long long tot = 0, cur = 1ull << 20;
    for (int c = 0; c < G; ++c) { tot += cur; cur = cur * 82 / 100; }
    const unsigned long long u = (static_cast<unsigned long long>(h) * tot) >> 32;
    unsigned long long acc = 0;
    cur = 1ull << 20;
    for (int c = 0; c < G; ++c) { acc += cur; if (u < acc) return c; cur = cur * 82 / 100; }
    return G - 1;
}
__host__ __device__ inline float synth_value(uint32_t seed, uint32_t gene, uint32_t cell, int cl, int G, int nmark) {
    const uint32_t hg = synth_mix(gene * 0x27d4eb2fu + seed);
    const uint32_t h = synth_mix(synth_mix(gene * 0x9e3779b1u + seed) ^ (cell * 0x85ebca77u + 0xc2b2ae3du));
    const uint32_t u = h >> 8;
    const bool marker = (static_cast<int>(gene / static_cast<uint32_t>(nmark)) == cl) && (gene < static_cast<uint32_t>(G * nmark));
    const uint32_t lvl = hg & 3u;
    const uint32_t z_base[4] = {16106127u, 15770583u, 15435038u, 14763950u};
    const uint32_t z_mark[4] = {3355443u, 2516582u, 2013266u, 1677722u};
    const uint32_t z = marker ? z_mark[lvl] : z_base[lvl];
    if (u < z) return 0.0f;
    uint32_t r = synth_mix(h ^ 0x68bc21ebu);
    int cnt = 1;
    const uint32_t thr = marker ? 0xD0000000u : 0x50000000u;
    while (r < thr && cnt < 64) { ++cnt; r = synth_mix(r + 0x9e3779b9u); }
    return static_cast<float>(cnt);
}

__global__ void synth_fill_kernel(uint32_t seed, int m, long long cell0, int ncell, int G, int nmark, float *X, long long ld) {
    const long long cell = blockIdx.y;
    if (cell >= ncell) return;
    const uint32_t gcell = static_cast<uint32_t>(cell0 + cell);
    const int cl = synth_cluster(seed, gcell, G);
    float *col = X + cell * ld;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < m; g += gridDim.x * blockDim.x)
        col[g] = synth_value(seed, static_cast<uint32_t>(g), gcell, cl, G, nmark);
}

}  // namespace sharp

using namespace sharp;

extern "C" {

int sharp_project_dev(int proj, const float *dX, int m, int n, long long ld, int log_flag, double *dE, long long ldE) {
    SHARP_API_BEGIN
    auto pr = get_projector(proj);
    project_dev(*pr, dX, m, n, ld, log_flag, dE, ldE, nullptr);
    SHARP_API_END
}

int sharp_project_dev64(int proj, const double *dX, int m, int n, long long ld, int log_flag, double *dE, long long ldE) {
    SHARP_API_BEGIN
    auto pr = get_projector(proj);
    project_dev(*pr, dev64_ref(dX, m, n, ld), m, n, ld, log_flag, dE, ldE, nullptr);
    SHARP_API_END
}

int sharp_project(int proj, const double *X, int m, int n, long long ld, int log_flag, double *E) {
    SHARP_API_BEGIN
    auto pr = get_projector(proj);
    SHARP_REQUIRE(X && E, "sharp_project: null buffer");
    SHARP_REQUIRE(ld >= m, "sharp_project: ld < m");
    HostBlock hb;                                       // fp32 when exact, else fp64 (upload.hpp)
    upload_block(X, m, n, ld, hb);
    const long long ldE = static_cast<long long>(pr->K) * pr->p;
    DevBuf<double> dE(static_cast<size_t>(ldE) * n);
    project_dev(*pr, hb.ref(), m, n, hb.ld, log_flag, dE.p, ldE, nullptr);
    dE.download(E, static_cast<size_t>(ldE) * n);
    SHARP_API_END
}

int sharp_synth_fill_dev(unsigned seed, int m, long long cell0, int ncell, int G, int nmark, float *dX, long long ld) {
    SHARP_API_BEGIN
    SHARP_REQUIRE(G >= 1 && G <= 64 && nmark >= 1 && ld >= m, "sharp_synth_fill_dev: bad arguments");
    if (ncell > 0) {
        Ctx &c = ctx();
        KernelTimer t("synth_fill");
        for (int c0 = 0; c0 < ncell; c0 += 65535) {   // grid.y limit
            const int nc = std::min(65535, ncell - c0);
            hipLaunchKernelGGL(synth_fill_kernel, dim3((m + 1023) / 1024, nc), dim3(256), 0, c.stream, seed, m, cell0 + c0, nc,
                               G, nmark, dX + static_cast<long long>(c0) * ld, ld);
            launch_check("synth_fill_kernel");
        }
    }
    SHARP_API_END
}

int sharp_synth_labels(unsigned seed, long long cell0, int ncell, int G, int *labels) {
    SHARP_API_BEGIN
    for (int i = 0; i < ncell; ++i) labels[i] = synth_cluster(seed, static_cast<uint32_t>(cell0 + i), G);
    SHARP_API_END
}

}  // extern "C"
