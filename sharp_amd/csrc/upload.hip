// upload.hip -- see upload.hpp.  The caller's memory is pageable, so a matrix is cut into slabs of cells: host threads narrow (or
// copy) a slab into one of two pinned staging buffers while the previous slab's DMA is in flight.
#include "upload.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>
#include <type_traits>

namespace sharp {

namespace {

struct UploadStage {
    void *pinned[2] = {nullptr, nullptr};
    size_t cap = 0;                       // bytes per staging buffer
    hipEvent_t done[2] = {nullptr, nullptr};
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        for (int q = 0; q < 2; ++q) {
            if (pinned[q]) (void)hipHostFree(pinned[q]);
            SHARP_HIP_CHECK(hipHostMalloc(&pinned[q], bytes, hipHostMallocDefault));
            if (!done[q]) SHARP_HIP_CHECK(hipEventCreateWithFlags(&done[q], hipEventDisableTiming));
        }
        cap = bytes;
    }
};
UploadStage &upload_stage() { return per_slot<UploadStage>(); }
int &last_storage() { static int s = 0; return s; }
int &last_wire() { static int s = 0; return s; }

int upload_threads() {
    unsigned hw = static_cast<unsigned>(host_cores());
    if (knobs().upload_threads > 0) hw = static_cast<unsigned>(knobs().upload_threads);
    // (a dense fp64 block, four cfg3 blocks per call: 0.80 / 0.46 / 0.36 / 0.41 / 0.48 / 0.58 s with 8 / 16 / 24 / 32 / 48 / 96 threads: more readers
    // of one pageable matrix than ~32 get in each other's way; a sparse block does not care beyond 16)
    return static_cast<int>(std::max(1u, std::min(hw ? hw : 4u, 32u)));
}
// 0 auto, 32, 64
int storage_policy() { return knobs().x_storage; }       // SHARP_X_STORAGE=fp32 / fp64
// the slot's persistent worker pool (a block is packed slab by slab: seven rounds of threads per 50 000-cell sparse block -- spawning them anew
// per slab cost a third of a block's upload)
template <typename F>
void run_threads(int nthr, long long items, F fn) {
    if (nthr == 1 || items < nthr) { for (int t = 0; t < nthr; ++t) fn(t); return; }
    host_pool_threads_hint(nthr - 1);                    // (takes effect if this slot's pool does not exist yet: an upload slot's first block)
    host_parallel_for(nthr, nthr, [&](int t) { fn(t); });
}
// does the double survive the round trip through float?  (NaN counts as exact: it is NaN either way; +-inf is exact; a finite
// value beyond FLT_MAX becomes inf and is not)
inline bool f32_exact(double x) { return !(static_cast<double>(static_cast<float>(x)) != x) || x != x; }

// What crosses PCIe.  A block is sent in the NARROWEST type that holds every one of its values exactly, decided while it is packed (no
// scan of its own): unsigned 8-bit integers (counts up to 255: most UMI data), unsigned 16-bit integers (counts up to 65535), else float, else double.  The
// attempt at a type stops at the first slab that holds a value outside it and the block starts over one type wider -- for TPM-like data
// that is the first slab, a few milliseconds.  u16 and float blocks are STORED as fp32 (the RP kernel's table covers the counts), double
// blocks as fp64.
template <typename Wt> inline bool wire_holds(double x);
template <> inline bool wire_holds<uint8_t>(double x) { return x >= 0.0 && x <= 255.0 && static_cast<double>(static_cast<uint8_t>(x)) == x; }
template <> inline bool wire_holds<uint16_t>(double x) { return x >= 0.0 && x <= 65535.0 && static_cast<double>(static_cast<uint16_t>(x)) == x; }
template <> inline bool wire_holds<float>(double x) { return f32_exact(x); }
template <> inline bool wire_holds<double>(double) { return true; }

template <typename Wt, typename Dt>
__global__ void widen_kernel(const Wt *__restrict__ src, Dt *__restrict__ dst, long long count) {
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < count; q += static_cast<long long>(gridDim.x) * blockDim.x)
        dst[q] = static_cast<Dt>(src[q]);
}

struct WireStage { DevBuf<unsigned char> slab[2]; };       // device side of a wire type narrower than the stored one
WireStage &wire_stage() { return per_slot<WireStage>(); }

// One pass over a dense matrix sent as Wt and stored as Dt.  Returns false as soon as a slab held a value Wt cannot represent (check
// only); the device block is then incomplete.
template <typename Wt, typename Dt>
bool upload_as(const double *X, int m, long long n, long long ld, Dt *dX, long long ldd, bool check, double *max_abs) {
    constexpr bool direct = std::is_same<Wt, Dt>::value;                              // the DMA writes the block itself
    const size_t slab_bytes = static_cast<size_t>(128) << 20;                         // 128 MB per staging buffer
    const long long slab = std::max<long long>(1, std::min<long long>(n, static_cast<long long>(slab_bytes / (sizeof(Wt) * ldd))));
    UploadStage &U = upload_stage();
    U.ensure(static_cast<size_t>(slab) * ldd * sizeof(Wt));
    WireStage &WS = wire_stage();
    if (!direct) for (int k = 0; k < 2; ++k) WS.slab[k].ensure(static_cast<size_t>(slab) * ldd * sizeof(Wt));
    const int nthr = upload_threads();
    Ctx &cx = ctx();
    hipStream_t s = cx.stream;
    int q = 0;
    bool used[2] = {false, false};
    std::atomic<int> inexact{0};
    std::vector<double> tmax(static_cast<size_t>(nthr), 0.0);
    for (long long c0 = 0; c0 < n; c0 += slab, q ^= 1) {
        const long long nc = std::min(slab, n - c0);
        if (used[q]) SHARP_HIP_CHECK(hipEventSynchronize(U.done[q]));                  // the DMA (and the widening) that last read this buffer
        Wt *dst = static_cast<Wt *>(U.pinned[q]);
        run_threads(nthr, nc, [&](int t) {
            const long long a = nc * t / nthr, b = nc * (t + 1) / nthr;
            int bad = 0;
            double mx = tmax[t];
            for (long long c = a; c < b; ++c) {
                const double *src = X + (c0 + c) * ld;
                Wt *d = dst + c * ldd;
                if (check) for (int g = 0; g < m; ++g) { const double x = src[g]; d[g] = static_cast<Wt>(x); bad |= !wire_holds<Wt>(x); mx = std::fabs(x) > mx ? std::fabs(x) : mx; if (x != x) mx = HUGE_VAL; }
                else for (int g = 0; g < m; ++g) { const double x = src[g]; d[g] = static_cast<Wt>(x); mx = std::fabs(x) > mx ? std::fabs(x) : mx; if (x != x) mx = HUGE_VAL; }
                for (long long g = m; g < ldd; ++g) d[g] = Wt(0);
            }
            tmax[t] = mx;
            if (bad) inexact.store(1, std::memory_order_relaxed);
        });
        if (check && inexact.load()) { SHARP_HIP_CHECK(hipStreamSynchronize(s)); return false; }
        const size_t cnt = static_cast<size_t>(nc) * ldd;
        if constexpr (direct) {
            SHARP_HIP_CHECK(hipMemcpyAsync(dX + c0 * ldd, dst, cnt * sizeof(Wt), hipMemcpyHostToDevice, s));
        } else {
            Wt *dw = reinterpret_cast<Wt *>(WS.slab[q].p);
            SHARP_HIP_CHECK(hipMemcpyAsync(dw, dst, cnt * sizeof(Wt), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL((widen_kernel<Wt, Dt>), dim3(cx.num_cu * 8), dim3(256), 0, s, dw, dX + c0 * ldd, static_cast<long long>(cnt));
            launch_check("widen_kernel");
        }
        SHARP_HIP_CHECK(hipEventRecord(U.done[q], s));
        used[q] = true;
    }
    stream_sync();
    double mx = 0;
    for (double v : tmax) mx = v > mx ? v : mx;
    // the reference would carry NA / NaN / Inf into cor() and stop in hclust ("NA/NaN/Inf in foreign function call")
    if (!(mx <= 1.7976931348623157e308)) throw Error(SHARP_ERR_ARG, "NA/NaN/Inf in the expression matrix");
    if (max_abs) *max_abs = mx;
    return true;
}

// dgCMatrix slab -> dense block: one wave per cell scatters the cell's (row index, value) pairs into its zeroed column.
// Out-of-range row indices are counted, never written.
template <typename It, typename Vt, typename Dt>
__global__ void csc_expand_kernel(const long long *__restrict__ colptr, const It *__restrict__ rowidx, const Vt *__restrict__ val,
                                  long long e_base, long long ncell, int m, Dt *__restrict__ dX, long long ld, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) >> 6;
    const long long nwave = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
    for (long long c = wave; c < ncell; c += nwave) {
        const long long e0 = colptr[c] - e_base, e1 = colptr[c + 1] - e_base;
        Dt *col = dX + c * ld;
        for (long long e = e0 + lane; e < e1; e += 64) {
            const long long g = static_cast<long long>(rowidx[e]);
            if (g >= 0 && g < m) col[g] = static_cast<Dt>(val[e]);
            else atomicAdd(bad, 1);
        }
    }
}

// The same without the zero-fill pass and without scattered 4-byte writes to HBM: one workgroup builds a cell's column tile by tile in LDS
// (zero, place the cell's entries that fall into the tile, stream the tile out with 16-byte stores), so the dense block is WRITTEN ONCE,
// coalesced -- 4 GB per 50 000 x 20 000 block instead of a 4 GB memset plus 1e8 partial-line writes -- and the expansion of a block that
// arrives while other blocks are being clustered does not fight their kernels for the memory system (round 6: with the memset + scatter form
// a block's upload took 13 ms on an idle GPU and 29-31 ms beside the clustering, exactly the clustering's own pace; with this form 13-15 ms).
// The tile is small (16 KB): the agglomeration's workgroups hold a whole CU's LDS each, and a 64 KB tile waited up to 32 ms for a CU without
// one; a cell's entries are re-scanned per tile (five tiles at 20 000 genes: from the L2).  Entries need not be sorted.  Out-of-range row
// indices are counted, never written.
constexpr int kExpandTileBytes = 16 * 1024;
template <typename It, typename Vt, typename Dt>
__global__ __launch_bounds__(256) void csc_expand_rows_kernel(const long long *__restrict__ colptr, const It *__restrict__ rowidx, const Vt *__restrict__ val,
                                                             long long e_base, long long ncell, int m, Dt *__restrict__ dX, long long ld, int *__restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    Dt *tile = reinterpret_cast<Dt *>(tile_raw);
    constexpr int TILE = kExpandTileBytes / static_cast<int>(sizeof(Dt));          // elements per tile
    constexpr int V = 16 / static_cast<int>(sizeof(Dt));                           // elements per 16-byte store
    typedef Dt vec_t __attribute__((ext_vector_type(V)));
    const int tid = threadIdx.x;
    int nbad = 0;
    const bool vec = ld % V == 0 && (reinterpret_cast<uintptr_t>(dX) & 15u) == 0;  // the library's own blocks; a caller's block may have any stride
    for (long long c = blockIdx.x; c < ncell; c += gridDim.x) {
        const long long e0 = colptr[c] - e_base, e1 = colptr[c + 1] - e_base;
        Dt *col = dX + c * ld;
        for (long long t0 = 0; t0 < ld; t0 += TILE) {
            const int len = static_cast<int>(ld - t0 < TILE ? ld - t0 : TILE);
            for (int q = tid * V; q < len; q += 256 * V) *reinterpret_cast<vec_t *>(tile + q) = vec_t(0);     // (TILE is a multiple of V: a ragged end stays inside the tile)
            __syncthreads();
            for (long long e = e0 + tid; e < e1; e += 256) {
                const long long g = static_cast<long long>(rowidx[e]);
                if (g >= t0 && g < t0 + len && g < m) tile[g - t0] = static_cast<Dt>(val[e]);
                else if (t0 == 0 && (g < 0 || g >= m)) ++nbad;
            }
            __syncthreads();
            if (vec) for (int q = tid * V; q < len; q += 256 * V) __builtin_nontemporal_store(*reinterpret_cast<const vec_t *>(tile + q), reinterpret_cast<vec_t *>(col + t0 + q));
            else for (int q = tid; q < len; q += 256) col[t0 + q] = tile[q];
            __syncthreads();
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <typename It, typename Vt, typename Dt>
void launch_csc_expand(const long long *d_colptr, const It *d_idx, const Vt *d_val, long long e_base, long long ncell, int m, Dt *dX, long long ld, int *d_bad,
                       hipStream_t s) {
    Ctx &cx = ctx();
    auto kern = csc_expand_rows_kernel<It, Vt, Dt>;
    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kExpandTileBytes));
    const int blocks = static_cast<int>(std::min<long long>(ncell, static_cast<long long>(cx.num_cu) * 4));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), kExpandTileBytes, s, d_colptr, d_idx, d_val, e_base, ncell, m, dX, ld, d_bad);
    launch_check("csc_expand_rows_kernel");
}

// Only the non-zeros cross PCIe, slab by slab through the pinned staging buffers: per entry a row index of It (uint16_t when the block has
// at most 65 536 genes, else int32) and a value of Vt (see wire_holds): 4 bytes per non-zero for counts over 20 000 genes, 12 as R holds
// them.  Returns 0 when the block is complete; at the first slab with a value Vt cannot hold (check only) 1, with a row index It cannot 2.
template <typename It, typename Vt, typename Dt>
int upload_csc_as(const int *colptr, const int *rowidx, const double *val, int m, long long n, Dt *dX, long long ldd, bool check, double *max_abs) {
    if (n <= 0) return 0;
    HostTimer ht("upload_csc");
    Ctx &cx = ctx();
    hipStream_t s = cx.stream;
    std::vector<long long> cp(static_cast<size_t>(n) + 1);
    for (long long c = 0; c <= n; ++c) cp[c] = colptr[c];
    DevBuf<long long> dcp;
    dcp.alloc_pooled(cp.size());
    dcp.upload(cp.data(), cp.size());
    DevBuf<int> dbad;
    dbad.alloc_pooled(1);
    dbad.zero();
    const long long slab_e = 16LL << 20;                                   // entries per slab
    const size_t esz = sizeof(It) + sizeof(Vt);
    UploadStage &U = upload_stage();
    U.ensure(static_cast<size_t>(slab_e) * esz + 16);                      // [values | indices] share one staging buffer
    WireStage &WS = wire_stage();                                          // ... and one device slab per staging buffer
    for (int k = 0; k < 2; ++k) WS.slab[k].ensure(static_cast<size_t>(std::max<long long>(1, std::min<long long>(slab_e, cp[n] - cp[0]))) * esz + 16);
    const int nthr = upload_threads();
    int q = 0;
    bool used[2] = {false, false};
    long long c0 = 0;
    std::atomic<int> inexact{0}, badidx{0};
    std::vector<double> tmax(static_cast<size_t>(nthr), 0.0);
    while (c0 < n) {
        // whole cells per slab; a single cell never exceeds m <= 2^31 entries but may exceed the slab: grow the slab for it
        long long c1 = c0;
        const long long e0 = cp[c0];
        while (c1 < n && cp[c1 + 1] - e0 <= slab_e) ++c1;
        if (c1 == c0) {
            c1 = c0 + 1;
            SHARP_HIP_CHECK(hipStreamSynchronize(s));
            U.ensure(static_cast<size_t>(cp[c1] - e0) * esz + 16);
            for (int k = 0; k < 2; ++k) { WS.slab[k].ensure(static_cast<size_t>(cp[c1] - e0) * esz + 16); used[k] = false; }
        }
        const long long ne = cp[c1] - e0;
        if (ne == 0) {                                                      // cells without a single non-zero: their columns are zeros
            SHARP_HIP_CHECK(hipMemsetAsync(dX + c0 * ldd, 0, static_cast<size_t>(ldd) * (c1 - c0) * sizeof(Dt), s));
        }
        if (ne > 0) {
            if (used[q]) { HostTimer hw("upload_csc_wait_staging"); SHARP_HIP_CHECK(hipEventSynchronize(U.done[q])); }
            // [values | indices] or [indices | values]: the wider type first, so that both runs are aligned
            constexpr bool vfirst = sizeof(Vt) >= sizeof(It);
            Vt *hv = vfirst ? static_cast<Vt *>(U.pinned[q]) : reinterpret_cast<Vt *>(static_cast<It *>(U.pinned[q]) + ne);
            It *hi = vfirst ? reinterpret_cast<It *>(static_cast<Vt *>(U.pinned[q]) + ne) : static_cast<It *>(U.pinned[q]);
            auto pack = [&](int t) {
                const long long a = ne * t / nthr, b = ne * (t + 1) / nthr;
                int bad = 0, bi = 0;
                double mx = tmax[t];
                if (std::is_same<It, int>::value) std::memcpy(hi + a, rowidx + e0 + a, static_cast<size_t>(b - a) * sizeof(int));
                else for (long long e = a; e < b; ++e) { const int g = rowidx[e0 + e]; hi[e] = static_cast<It>(g); bi |= static_cast<int>(static_cast<It>(g)) != g; }
                for (long long e = a; e < b; ++e) {
                    const double x = val[e0 + e];
                    hv[e] = static_cast<Vt>(x);
                    if (check) bad |= !wire_holds<Vt>(x);
                    mx = std::fabs(x) > mx ? std::fabs(x) : mx;
                    if (x != x) mx = HUGE_VAL;
                }
                tmax[t] = mx;
                if (bad) inexact.store(1, std::memory_order_relaxed);
                if (bi) badidx.store(1, std::memory_order_relaxed);
            };
            {
                HostTimer hp("upload_csc_pack");
                if (ne < (1 << 16)) { for (int t = 0; t < nthr; ++t) pack(t); }
                else run_threads(nthr, ne, pack);
            }
            if (inexact.load() || badidx.load()) {
                // a row index It cannot hold can only be an index outside [0, genes): the int32 attempt reports it; a value Vt cannot hold: one type wider
                SHARP_HIP_CHECK(hipStreamSynchronize(s));
                return badidx.load() ? 2 : 1;
            }
            Vt *dv = vfirst ? reinterpret_cast<Vt *>(WS.slab[q].p) : reinterpret_cast<Vt *>(reinterpret_cast<It *>(WS.slab[q].p) + ne);
            It *di = vfirst ? reinterpret_cast<It *>(reinterpret_cast<Vt *>(WS.slab[q].p) + ne) : reinterpret_cast<It *>(WS.slab[q].p);
            SHARP_HIP_CHECK(hipMemcpyAsync(WS.slab[q].p, U.pinned[q], static_cast<size_t>(ne) * esz, hipMemcpyHostToDevice, s));   // values and indices in one transfer
            launch_csc_expand<It, Vt, Dt>(dcp.p + c0, di, dv, e0, c1 - c0, m, dX + c0 * ldd, ldd, dbad.p, s);   // (writes every cell of the slab whole: no zero-fill pass)
            SHARP_HIP_CHECK(hipEventRecord(U.done[q], s));                  // (behind the expansion: the device slab is free with the staging buffer)
            used[q] = true;
            q ^= 1;
        }
        c0 = c1;
    }
    int bad = 0;
    { HostTimer hd("upload_csc_drain"); dbad.download(&bad, 1); }           // also drains the stream: staging and slabs are free again
    if (bad) throw Error(SHARP_ERR_ARG, "sparse input: row index outside [0, genes)");
    double mx = 0;
    for (double v : tmax) mx = v > mx ? v : mx;
    if (!(mx <= 1.7976931348623157e308)) throw Error(SHARP_ERR_ARG, "NA/NaN/Inf in the expression matrix");
    if (max_abs) *max_abs = mx;
    return 0;
}

// the attempts in order (narrowest first), into an fp32 block; false: a value needs fp64
bool upload_csc_f32(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ldd, bool check, double *max_abs) {
    bool narrow_idx = m <= 65536;      // (a row index beyond 65 535 in such a block is outside [0, genes): the int32 attempt reports it)
    int r = 1;
    if (check) {
        if (narrow_idx) { r = upload_csc_as<uint16_t, uint8_t, float>(colptr, rowidx, val, m, n, dX, ldd, true, max_abs); if (r == 2) narrow_idx = false; }
        if (!narrow_idx) r = upload_csc_as<int, uint8_t, float>(colptr, rowidx, val, m, n, dX, ldd, true, max_abs);
        if (r == 0) { last_wire() = 8; return true; }
        if (narrow_idx) { r = upload_csc_as<uint16_t, uint16_t, float>(colptr, rowidx, val, m, n, dX, ldd, true, max_abs); if (r == 2) narrow_idx = false; }
        if (!narrow_idx) r = upload_csc_as<int, uint16_t, float>(colptr, rowidx, val, m, n, dX, ldd, true, max_abs);
        if (r == 0) { last_wire() = 16; return true; }
    }
    if (narrow_idx) { r = upload_csc_as<uint16_t, float, float>(colptr, rowidx, val, m, n, dX, ldd, check, max_abs); if (r == 2) narrow_idx = false; }
    if (!narrow_idx) r = upload_csc_as<int, float, float>(colptr, rowidx, val, m, n, dX, ldd, check, max_abs);
    if (r == 0) { last_wire() = 32; return true; }
    return false;
}

void check_csc(const int *colptr, const int *rowidx, const double *val, long long n) {
    if (!colptr || (!rowidx && colptr[n] > 0) || (!val && colptr[n] > 0)) throw Error(SHARP_ERR_ARG, "sparse input: null pointer");
    if (colptr[0] < 0) throw Error(SHARP_ERR_ARG, "sparse input: negative column pointer");
    for (long long c = 0; c < n; ++c)
        if (colptr[c + 1] < colptr[c]) throw Error(SHARP_ERR_ARG, "sparse input: column pointers must be non-decreasing");
}

}  // namespace

void upload_block(const double *X, int m, long long n, long long ld, HostBlock &hb) {
    const int policy = storage_policy();
    if (n <= 0) { hb.f64 = false; hb.ld = (static_cast<long long>(m) + 3) / 4 * 4; return; }
    HostTimer ht("upload_narrow_and_copy");
    if (policy != 64) {
        const long long ldd = (static_cast<long long>(m) + 3) / 4 * 4;
        { HostTimer ha("upload_alloc"); hb.f.ensure(static_cast<size_t>(ldd) * n); }
        // counts cross PCIe as 16-bit integers (a quarter of the doubles R holds), other fp32-exact values as floats
        bool ok = policy == 0 && upload_as<uint8_t, float>(X, m, n, ld, hb.f.p, ldd, true, &hb.max_abs);
        if (ok) last_wire() = 8;
        else if (policy == 0 && (ok = upload_as<uint16_t, float>(X, m, n, ld, hb.f.p, ldd, true, &hb.max_abs))) last_wire() = 16;
        else { ok = upload_as<float, float>(X, m, n, ld, hb.f.p, ldd, policy == 0, &hb.max_abs); if (ok) last_wire() = 32; }
        if (ok) {
            hb.d.release();                               // (an fp64 copy left by an earlier block of another kind)
            hb.f64 = false; hb.ld = ldd; last_storage() = 32; return;
        }
        hb.f.release();                                   // the fp32 attempt is of no further use: an fp64 block must not cost 1.5 x its bytes
    }
    // a value fp32 cannot hold exactly (or fp64 forced): the block stays in double
    const long long ldd = (static_cast<long long>(m) + 1) / 2 * 2;
    { HostTimer ha("upload_alloc"); hb.d.ensure(static_cast<size_t>(ldd) * n); }
    upload_as<double, double>(X, m, n, ld, hb.d.p, ldd, false, &hb.max_abs);
    hb.f64 = true; hb.ld = ldd; last_storage() = 64; last_wire() = 64;
}

void upload_block_csc(const int *colptr, const int *rowidx, const double *val, int m, long long n, HostBlock &hb) {
    if (n <= 0) return;
    check_csc(colptr, rowidx, val, n);
    const int policy = storage_policy();
    if (policy != 64) {
        hb.ld = (static_cast<long long>(m) + 3) / 4 * 4;
        hb.f.ensure(static_cast<size_t>(hb.ld) * n);
        if (upload_csc_f32(colptr, rowidx, val, m, n, hb.f.p, hb.ld, policy == 0, &hb.max_abs)) {
            hb.d.release();
            hb.f64 = false; last_storage() = 32; return;
        }
        hb.f.release();
    }
    hb.ld = (static_cast<long long>(m) + 1) / 2 * 2;
    hb.d.ensure(static_cast<size_t>(hb.ld) * n);
    upload_csc_as<int, double, double>(colptr, rowidx, val, m, n, hb.d.p, hb.ld, false, &hb.max_abs);
    hb.f64 = true;
    last_storage() = 64; last_wire() = 64;
}

void upload_csc_into_f32(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ld) {
    if (n <= 0) return;
    check_csc(colptr, rowidx, val, n);
    upload_csc_f32(colptr, rowidx, val, m, n, dX, ld, false, nullptr);      // (values are narrowed: the caller's block is fp32)
}

// A packed CSC block ALREADY IN DEVICE MEMORY (a block file of the compact format, DMA'd as it is: sharp_amd/blocks.py) -> the dense block.
void expand_packed_csc_dev(const long long *d_colptr, const void *d_idx, int idx_bits, const void *d_val, int val_bits, int m, long long n,
                           void *dX, long long ld, bool dx_f64) {
    if (n <= 0) return;
    hipStream_t s = ctx().stream;
    DevBuf<int> dbad;
    dbad.alloc_pooled(1);
    dbad.zero();
#define SHARP_EXPAND(IT, VT, DT)                                                                                                         \
    launch_csc_expand<IT, VT, DT>(d_colptr, static_cast<const IT *>(d_idx), static_cast<const VT *>(d_val), 0LL, n, m, static_cast<DT *>(dX), ld, dbad.p, s)
    const int key = val_bits == 8 ? 100 + (idx_bits == 16 ? 0 : 2) + (dx_f64 ? 1 : 0)
                                  : (idx_bits == 16 ? 0 : 1) * 8 + (val_bits == 16 ? 0 : val_bits == 32 ? 1 : 2) * 2 + (dx_f64 ? 1 : 0);
    switch (key) {
        case 100: SHARP_EXPAND(uint16_t, uint8_t, float); break;
        case 101: SHARP_EXPAND(uint16_t, uint8_t, double); break;
        case 102: SHARP_EXPAND(int, uint8_t, float); break;
        case 103: SHARP_EXPAND(int, uint8_t, double); break;
        case 0: SHARP_EXPAND(uint16_t, uint16_t, float); break;
        case 1: SHARP_EXPAND(uint16_t, uint16_t, double); break;
        case 2: SHARP_EXPAND(uint16_t, float, float); break;
        case 3: SHARP_EXPAND(uint16_t, float, double); break;
        case 5: SHARP_EXPAND(uint16_t, double, double); break;
        case 8: SHARP_EXPAND(int, uint16_t, float); break;
        case 9: SHARP_EXPAND(int, uint16_t, double); break;
        case 10: SHARP_EXPAND(int, float, float); break;
        case 11: SHARP_EXPAND(int, float, double); break;
        case 13: SHARP_EXPAND(int, double, double); break;
        default: throw Error(SHARP_ERR_ARG, "packed sparse block: 64-bit values need an fp64 block");
    }
#undef SHARP_EXPAND
    int bad = 0;
    dbad.download(&bad, 1);
    if (bad) throw Error(SHARP_ERR_ARG, "sparse input: row index outside [0, genes)");
}

void upload_release_staging() {
    UploadStage &U = upload_stage();
    for (int q = 0; q < 2; ++q) if (U.pinned[q]) { (void)hipHostFree(U.pinned[q]); U.pinned[q] = nullptr; }
    U.cap = 0;
    WireStage &WS = wire_stage();
    for (int k = 0; k < 2; ++k) WS.slab[k].release();
}

int upload_last_storage() { return last_storage(); }
int upload_last_wire() { return last_wire(); }

}  // namespace sharp
