// rp_dense.hip -- the RP matmul with the projector materialised DENSE, on the f64 MFMA (R/RPmat.R:32
// `1/sqrt(p) * t(x) %*% scdata`, whose commented-out n < 10000 branch, R/RPmat.R:36-41, is the dense `matrix` product).
// The fallback for projectors that are not sparse -- density 1/sqrt(m) >= 1/4, i.e. m <= 16 genes -- and a cross-check of the sparse
// kernels (SHARP_RP_KERNEL=dense).  Per launch group and chunk of cells:
//   R_d  (m x K*p, k-major)    = the group's packed row lists scattered into +-1 (the scale sqrt(s)/sqrt(p) goes into the store pass)
//   L    (m x cells, k-major)  = log2(1 + x) in fp64, transposed through LDS (X is cell-major)
//   E_c  (cells x K*p)         = L^T R_d            gemm_tn_f64 (v_mfma_f64_16x16x4_f64, linalg.hip)
//   E[row_map[cell]][k*p + c]  = sqrt(s)/sqrt(p) * E_c
// 2 * cells * m * K*p flop: 11.7 PFLOP for one 50 000 x 20 000 block at K = 15, against 4.7 G integer adds in the sparse form -- which is
// why this is a fallback.  Sums are fp64 FMA chains in MFMA order, not the exact fixed-point sums of rp2.hip: E agrees to 1e-13
// relative (the parity bar is 2e-12 max|E|), not bit for bit.
#include "linalg.hpp"
#include "projector.hpp"

#include <cmath>

namespace sharp {

namespace {

// one thread per (gene, slot of its first segment or of one of its overflow segments)
__global__ void rp_densify_kernel(const uint16_t *__restrict__ ent, const uint2 *__restrict__ ovf_slot, int novf, int m, int span, int slots, int ncomp,
                                  int neg_base, int max_extra, double *__restrict__ Rd) {
    const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int per_gene = span * (1 + max_extra);
    const long long g = idx / per_gene;
    if (g >= m) return;
    const int r = static_cast<int>(idx - g * per_gene);
    const int sgm = r / span, i = r - sgm * span;
    size_t seg = static_cast<size_t>(g);
    if (sgm > 0) {
        if (novf == 0) return;
        const uint2 oi = ovf_slot[g];
        if (static_cast<uint32_t>(sgm) > oi.y) return;
        seg = static_cast<size_t>(oi.x) + (sgm - 1);
    }
    const uint16_t *s = ent + seg * span;
    const int lane = i / slots;
    const uint32_t pair01 = static_cast<uint32_t>(s[slots * lane]) | (static_cast<uint32_t>(s[slots * lane + 1]) << 16);
    if (!lane_live(pair01)) return;
    const uint32_t code = s[i];
    const int comp = static_cast<int>(code >> 3);
    if (neg_base > 0) {                                   // dual accumulators: the sign is the array the code points into
        if (comp < ncomp) Rd[g * ncomp + comp] = 1.0;
        else if (comp >= neg_base && comp - neg_base < ncomp) Rd[g * ncomp + comp - neg_base] = -1.0;
    } else if (comp < ncomp) Rd[g * ncomp + comp] = (code & kCodeNeg) ? -1.0 : 1.0;      // (>= ncomp: an unused slot of a live lane)
}

// L[g][i] = f(X[cell0 + i][g]) for a 64 x 64 tile per workgroup (X: cell-major, L: gene-major)
template <typename T>
__global__ __launch_bounds__(256) void rp_log_transpose_kernel(const T *__restrict__ X, int m, long long ld, long long cell0, int nc, int log_flag,
                                                               double *__restrict__ L, long long ldl) {
    __shared__ double tile[64][65];
    const int g0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {                    // row r of the tile = cell i0 + r, genes g0 .. g0 + 63 (coalesced)
        const int i = i0 + r, g = g0 + tx;
        double v = 0.0;
        if (i < nc && g < m) {
            const double x = static_cast<double>(X[(cell0 + i) * ld + g]);
            v = log_flag == 0 ? x : (log_flag == 2 ? log10(1.0 + x) : log2(1.0 + x));
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                    // row r of the output = gene g0 + r, cells i0 .. i0 + 63 (coalesced)
        const int g = g0 + r, i = i0 + tx;
        if (g < m && i < nc) L[static_cast<long long>(g) * ldl + i] = tile[tx][r];
    }
}

__global__ void rp_dense_store_kernel(const double *__restrict__ Ec, int nc, int ncomp, double val, double out_scale, double *__restrict__ E,
                                      long long ldE, int comp0, const int *__restrict__ row_map, long long cell0) {
    const long long total = static_cast<long long>(nc) * ncomp;
    for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long i = idx / ncomp;
        const int c = static_cast<int>(idx - i * ncomp);
        const long long cell = cell0 + i;
        const long long row = row_map ? static_cast<long long>(row_map[cell]) : cell;
        E[row * ldE + comp0 + c] = out_scale * (val * Ec[idx]);     // (the sparse kernels' order of the two factors)
    }
}

}  // namespace

void project_dev_dense(const Projector &pr, XRef X, int m, int n, long long ld, int log_flag, double *dE, long long ldE,
                       const int *d_row_map) {
    Ctx &c = ctx();
    const double out_scale = 1.0 / std::sqrt(static_cast<double>(pr.p));
    // cells per chunk: L (m x chunk fp64) at most 256 MB
    long long chunk = std::max<long long>(64, ((256LL << 20) / (8LL * m)) / 64 * 64);
    chunk = std::min<long long>(chunk, (n + 63) / 64 * 64);
    DevBuf<double> L(static_cast<size_t>(m) * chunk);
    for (const auto &g : pr.groups) {
        const int span = g.span();
        DevBuf<double> Rd(static_cast<size_t>(m) * g.ncomp), Ec(static_cast<size_t>(chunk) * g.ncomp);
        DevBuf<GemmTask> dt(1);
        Rd.zero();
        {
            KernelTimer t("rp_dense_projector");
            int max_extra = 0;
            if (g.novf > 0) {
                std::vector<uint2> oi(static_cast<size_t>(g.novf));
                g.ovf_info.download(oi.data(), oi.size());
                for (const uint2 &o : oi) max_extra = std::max(max_extra, static_cast<int>(o.y));
            }
            const long long threads = static_cast<long long>(m) * span * (1 + max_extra);
            hipLaunchKernelGGL(rp_densify_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, c.stream, g.ent.p, g.ovf_slot.p,
                               g.novf, m, span, g.slots, g.ncomp, g.neg_base, max_extra, Rd.p);
            launch_check("rp_densify_kernel");
        }
        for (long long c0 = 0; c0 < n; c0 += chunk) {
            const int nc = static_cast<int>(std::min<long long>(chunk, n - c0));
            {
                KernelTimer t("rp_dense_log");
                const dim3 grid((m + 63) / 64, (nc + 63) / 64);
                if (X.f64) hipLaunchKernelGGL(rp_log_transpose_kernel<double>, grid, dim3(256), 0, c.stream, X.d64(), m, ld, c0, nc, log_flag, L.p, chunk);
                else hipLaunchKernelGGL(rp_log_transpose_kernel<float>, grid, dim3(256), 0, c.stream, X.f32(), m, ld, c0, nc, log_flag, L.p, chunk);
                launch_check("rp_log_transpose_kernel");
            }
            const GemmTask t{L.p, Rd.p, Ec.p, nc, g.ncomp, m, chunk, static_cast<long long>(g.ncomp), static_cast<long long>(g.ncomp), 0, 0, 0};
            dt.upload(&t, 1);
            gemm_tn_f64_batched(dt.p, 1, nc, g.ncomp, "rp_dense_gemm");
            {
                KernelTimer ts("rp_dense_store");
                hipLaunchKernelGGL(rp_dense_store_kernel, dim3(c.num_cu * 4), dim3(256), 0, c.stream, Ec.p, nc, g.ncomp, pr.val, out_scale, dE, ldE, g.k0 * pr.p,
                                   d_row_map, c0);
                launch_check("rp_dense_store_kernel");
            }
        }
        stream_sync();                                    // the group's temporaries are released on scope exit
    }
}

}  // namespace sharp
