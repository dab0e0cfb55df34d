// upload.hip -- see upload.hpp.  The caller's memory is pageable, so a matrix is cut into slabs of cells: host threads narrow (or
// copy) a slab into one of two pinned staging buffers while the previous slab's DMA is in flight.
#include "upload.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>

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

int upload_threads() {
    unsigned hw = static_cast<unsigned>(host_cores());
    if (knobs().upload_threads > 0) hw = static_cast<unsigned>(knobs().upload_threads);
    return static_cast<int>(std::max(1u, std::min(hw ? hw : 4u, 32u)));
}
// 0 auto, 32, 64
int storage_policy() { return knobs().x_storage; }       // SHARP_X_STORAGE=fp32 / fp64
template <typename F>
void run_threads(int nthr, long long items, F fn) {
    if (nthr == 1 || items < nthr) { for (int t = 0; t < nthr; ++t) fn(t); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
}
// does the double survive the round trip through float?  (NaN counts as exact: it is NaN either way; +-inf is exact; a finite
// value beyond FLT_MAX becomes inf and is not)
inline bool f32_exact(double x) { return !(static_cast<double>(static_cast<float>(x)) != x) || x != x; }

// One pass over the matrix as elements of T (float: narrowed, with the exactness test when `check`; double: copied).  Returns false
// as soon as a slab held a value that fp32 cannot represent (check only); the device block is then incomplete.
template <typename T>
bool upload_as(const double *X, int m, long long n, long long ld, T *dX, long long ldd, bool check, double *max_abs) {
    const size_t slab_bytes = static_cast<size_t>(128) << 20;                         // 128 MB per staging buffer
    const long long slab = std::max<long long>(1, std::min<long long>(n, static_cast<long long>(slab_bytes / (sizeof(T) * ldd))));
    UploadStage &U = upload_stage();
    U.ensure(static_cast<size_t>(slab) * ldd * sizeof(T));
    const int nthr = upload_threads();
    hipStream_t s = ctx().stream;
    int q = 0;
    bool used[2] = {false, false};
    std::atomic<int> inexact{0}, nonfinite{0};
    std::vector<double> tmax(static_cast<size_t>(nthr), 0.0);
    for (long long c0 = 0; c0 < n; c0 += slab, q ^= 1) {
        const long long nc = std::min(slab, n - c0);
        if (used[q]) SHARP_HIP_CHECK(hipEventSynchronize(U.done[q]));                  // the DMA that last read this buffer
        T *dst = static_cast<T *>(U.pinned[q]);
        run_threads(nthr, nc, [&](int t) {
            const long long a = nc * t / nthr, b = nc * (t + 1) / nthr;
            int bad = 0;
            double mx = tmax[t];
            for (long long c = a; c < b; ++c) {
                const double *src = X + (c0 + c) * ld;
                T *d = dst + c * ldd;
                if (check) for (int g = 0; g < m; ++g) { const double x = src[g]; d[g] = static_cast<T>(x); bad |= !f32_exact(x); mx = std::fabs(x) > mx ? std::fabs(x) : mx; if (x != x) mx = HUGE_VAL; }
                else for (int g = 0; g < m; ++g) { const double x = src[g]; d[g] = static_cast<T>(x); mx = std::fabs(x) > mx ? std::fabs(x) : mx; if (x != x) mx = HUGE_VAL; }
                for (long long g = m; g < ldd; ++g) d[g] = T(0);
            }
            tmax[t] = mx;
            if (bad) inexact.store(1, std::memory_order_relaxed);
        });
        if (check && inexact.load()) { SHARP_HIP_CHECK(hipStreamSynchronize(s)); return false; }
        SHARP_HIP_CHECK(hipMemcpyAsync(dX + c0 * ldd, dst, static_cast<size_t>(nc) * ldd * sizeof(T), hipMemcpyHostToDevice, s));
        SHARP_HIP_CHECK(hipEventRecord(U.done[q], s));
        used[q] = true;
    }
    stream_sync();
    double mx = 0;
    for (double v : tmax) mx = v > mx ? v : mx;
    (void)nonfinite;
    // the reference would carry NA / NaN / Inf into cor() and stop in hclust ("NA/NaN/Inf in foreign function call")
    if (!(mx <= 1.7976931348623157e308)) throw Error(SHARP_ERR_ARG, "NA/NaN/Inf in the expression matrix");
    if (max_abs) *max_abs = mx;
    return true;
}

// dgCMatrix slab -> dense block: one wave per cell scatters the cell's (row index, value) pairs into its zeroed column.
// Out-of-range row indices are counted, never written.
template <typename T>
__global__ void csc_expand_kernel(const long long *__restrict__ colptr, const int *__restrict__ rowidx, const T *__restrict__ val,
                                  long long e_base, long long ncell, int m, T *__restrict__ dX, long long ld, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) >> 6;
    const long long nwave = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
    for (long long c = wave; c < ncell; c += nwave) {
        const long long e0 = colptr[c] - e_base, e1 = colptr[c + 1] - e_base;
        T *col = dX + c * ld;
        for (long long e = e0 + lane; e < e1; e += 64) {
            const int g = rowidx[e];
            if (g >= 0 && g < m) col[g] = val[e];
            else atomicAdd(bad, 1);
        }
    }
}

// Only the non-zeros cross PCIe (int32 index + value of T), slab by slab through the pinned staging buffers.
template <typename T>
void upload_csc_as(const int *colptr, const int *rowidx, const double *val, int m, long long n, T *dX, long long ldd) {
    if (n <= 0) return;
    HostTimer ht("upload_csc");
    Ctx &cx = ctx();
    hipStream_t s = cx.stream;
    SHARP_HIP_CHECK(hipMemsetAsync(dX, 0, static_cast<size_t>(ldd) * n * sizeof(T), s));
    std::vector<long long> cp(static_cast<size_t>(n) + 1);
    for (long long c = 0; c <= n; ++c) cp[c] = colptr[c];
    DevBuf<long long> dcp(cp.size());
    dcp.upload(cp.data(), cp.size());
    DevBuf<int> dbad(1);
    dbad.zero();
    const long long slab_e = 16LL << 20;                                   // entries per slab
    const size_t esz = sizeof(int) + sizeof(T);
    UploadStage &U = upload_stage();
    U.ensure(static_cast<size_t>(slab_e) * esz);                           // [values | indices] share one staging buffer
    DevBuf<int> didx[2];
    DevBuf<T> dval[2];
    for (int k = 0; k < 2; ++k) { const size_t cap = static_cast<size_t>(std::max<long long>(1, std::min<long long>(slab_e, cp[n] - cp[0]))); didx[k].alloc(cap); dval[k].alloc(cap); }
    const int nthr = upload_threads();
    int q = 0;
    bool used[2] = {false, false};
    long long c0 = 0;
    while (c0 < n) {
        // whole cells per slab; a single cell never exceeds m <= 2^31 entries but may exceed the slab: grow the slab for it
        long long c1 = c0;
        const long long e0 = cp[c0];
        while (c1 < n && cp[c1 + 1] - e0 <= slab_e) ++c1;
        if (c1 == c0) { c1 = c0 + 1; SHARP_HIP_CHECK(hipStreamSynchronize(s)); U.ensure(static_cast<size_t>(cp[c1] - e0) * esz); for (int k = 0; k < 2; ++k) used[k] = false; }
        const long long ne = cp[c1] - e0;
        if (ne > 0) {
            if (used[q]) SHARP_HIP_CHECK(hipEventSynchronize(U.done[q]));
            T *hv = static_cast<T *>(U.pinned[q]);                          // values first: 8-byte alignment for doubles
            int *hi = reinterpret_cast<int *>(hv + ne);
            auto pack = [&](int t) {
                const long long a = ne * t / nthr, b = ne * (t + 1) / nthr;
                std::memcpy(hi + a, rowidx + e0 + a, static_cast<size_t>(b - a) * sizeof(int));
                for (long long e = a; e < b; ++e) hv[e] = static_cast<T>(val[e0 + e]);
            };
            if (ne < (1 << 16)) { for (int t = 0; t < nthr; ++t) pack(t); }
            else run_threads(nthr, ne, pack);
            didx[q].ensure(static_cast<size_t>(ne)); dval[q].ensure(static_cast<size_t>(ne));
            SHARP_HIP_CHECK(hipMemcpyAsync(didx[q].p, hi, static_cast<size_t>(ne) * sizeof(int), hipMemcpyHostToDevice, s));
            SHARP_HIP_CHECK(hipMemcpyAsync(dval[q].p, hv, static_cast<size_t>(ne) * sizeof(T), hipMemcpyHostToDevice, s));
            SHARP_HIP_CHECK(hipEventRecord(U.done[q], s));
            used[q] = true;
            const long long ncell = c1 - c0;
            const int blocks = static_cast<int>(std::min<long long>((ncell + 3) / 4, static_cast<long long>(cx.num_cu) * 16));
            hipLaunchKernelGGL(csc_expand_kernel<T>, dim3(blocks), dim3(256), 0, s, dcp.p + c0, didx[q].p, dval[q].p, e0, ncell, m,
                               dX + c0 * ldd, ldd, dbad.p);
            launch_check("csc_expand_kernel");
            q ^= 1;
        }
        c0 = c1;
    }
    int bad = 0;
    dbad.download(&bad, 1);                                                 // also drains the stream: staging and slabs are free again
    if (bad) throw Error(SHARP_ERR_ARG, "sparse input: row index outside [0, genes)");
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
        if (upload_as<float>(X, m, n, ld, hb.f.p, ldd, policy == 0, &hb.max_abs)) {
            hb.d.release();                               // (an fp64 copy left by an earlier block of another kind)
            hb.f64 = false; hb.ld = ldd; last_storage() = 32; return;
        }
        hb.f.release();                                   // the fp32 attempt is of no further use: an fp64 block must not cost 1.5 x its bytes
    }
    // a value fp32 cannot hold exactly (or fp64 forced): the block stays in double
    const long long ldd = (static_cast<long long>(m) + 1) / 2 * 2;
    { HostTimer ha("upload_alloc"); hb.d.ensure(static_cast<size_t>(ldd) * n); }
    upload_as<double>(X, m, n, ld, hb.d.p, ldd, false, &hb.max_abs);
    hb.f64 = true; hb.ld = ldd; last_storage() = 64;
}

void upload_block_csc(const int *colptr, const int *rowidx, const double *val, int m, long long n, HostBlock &hb) {
    if (n <= 0) return;
    check_csc(colptr, rowidx, val, n);
    const int policy = storage_policy();
    bool f64 = policy == 64;
    {                                                                       // one threaded scan of the stored values decides
        const long long ne = static_cast<long long>(colptr[n]) - colptr[0];
        const int nt = ne < (1 << 16) ? 1 : upload_threads();
        std::atomic<int> inexact{0};
        std::vector<double> tmax(static_cast<size_t>(nt), 0.0);
        run_threads(nt, ne, [&](int t) {
            const long long a = ne * t / nt, b = ne * (t + 1) / nt;
            int bad = 0;
            double mx = 0;
            for (long long e = a; e < b; ++e) { const double x = val[colptr[0] + e]; bad |= !f32_exact(x); mx = std::fabs(x) > mx ? std::fabs(x) : mx; if (x != x) mx = HUGE_VAL; }
            tmax[t] = mx;
            if (bad) inexact.store(1, std::memory_order_relaxed);
        });
        double mx = 0;
        for (double v : tmax) mx = v > mx ? v : mx;
        if (!(mx <= 1.7976931348623157e308)) throw Error(SHARP_ERR_ARG, "NA/NaN/Inf in the expression matrix");
        hb.max_abs = mx;
        if (policy == 0) f64 = inexact.load() != 0;
    }
    if (f64) {
        hb.ld = (static_cast<long long>(m) + 1) / 2 * 2;
        hb.d.ensure(static_cast<size_t>(hb.ld) * n);
        upload_csc_as<double>(colptr, rowidx, val, m, n, hb.d.p, hb.ld);
    } else {
        hb.ld = (static_cast<long long>(m) + 3) / 4 * 4;
        hb.f.ensure(static_cast<size_t>(hb.ld) * n);
        upload_csc_as<float>(colptr, rowidx, val, m, n, hb.f.p, hb.ld);
    }
    hb.f64 = f64;
    last_storage() = f64 ? 64 : 32;
}

void upload_csc_into_f32(const int *colptr, const int *rowidx, const double *val, int m, long long n, float *dX, long long ld) {
    if (n <= 0) return;
    check_csc(colptr, rowidx, val, n);
    upload_csc_as<float>(colptr, rowidx, val, m, n, dX, ld);
}

void upload_release_staging() {
    UploadStage &U = upload_stage();
    for (int q = 0; q < 2; ++q) if (U.pinned[q]) { (void)hipHostFree(U.pinned[q]); U.pinned[q] = nullptr; }
    U.cap = 0;
}

int upload_last_storage() { return last_storage(); }

}  // namespace sharp
