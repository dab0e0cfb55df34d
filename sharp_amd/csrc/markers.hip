// markers.hip -- SURVEY.md 8(f4): get_marker_genes (R/get_marker_genes.R:25-264) on the GPU.
// Per gene: sparsity, rank(x) over all cells (average ranks for ties), per-cluster mean rank and mean expression, the AUROC
// of the rank against "cell is in cluster c" for the rr clusters of highest mean rank (ROCR's auc == Mann-Whitney
// U / (n1 n2) with average ranks), the best of them, wilcox.test's two-sided p-value (normal approximation with continuity
// and tie correction: what stats::wilcox.test uses as soon as a group has >= 50 cells or there are ties) and the fold change
// mean(cluster) / max(mean(other clusters)).
// Layout: X is cells x genes (a cell is contiguous), so the genes are first transposed into per-gene lists of their NON-ZERO
// cells (value, cluster) -- zeros are one big tie group whose rank and per-cluster counts follow from the sizes -- then every
// list is sorted by value (rocPRIM segmented radix sort of the packed 64-bit keys: value bits above the cluster id; sorting
// on a bit range of the key, 32..64, came back ordered by the LOW word for short segments with this rocPRIM) and one workgroup per
// gene turns the sorted list into the statistics.  Rank sums are accumulated as integers (2 x rank), so they are exact.
#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"

namespace sharp {

constexpr int MG_TILE = 16384;        // genes per LDS histogram tile
constexpr int MG_THREADS = 512;
constexpr int MG_MAXG = 1024;         // clusters

__device__ __forceinline__ uint32_t mg_key(float v) {          // order-preserving map float -> uint32
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float mg_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// pass 1: non-zeros per gene.  grid (cell chunks, gene tiles); LDS histogram of the tile, flushed with one atomic per gene
__global__ __launch_bounds__(MG_THREADS) void mg_count_kernel(const float *__restrict__ X, int m, long long n, long long ld, int cells_per_block,
                                                              unsigned int *__restrict__ counts) {
    __shared__ unsigned int h[MG_TILE];
    const int g0 = blockIdx.y * MG_TILE, gn = min(MG_TILE, m - g0);
    for (int q = threadIdx.x; q < gn; q += MG_THREADS) h[q] = 0u;
    __syncthreads();
    const long long c0 = static_cast<long long>(blockIdx.x) * cells_per_block, c1 = min(n, c0 + cells_per_block);
    for (long long c = c0; c < c1; ++c) {
        const float *row = X + c * ld + g0;
        for (int q = threadIdx.x; q < gn; q += MG_THREADS)
            if (row[q] != 0.0f) atomicAdd(&h[q], 1u);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < gn; q += MG_THREADS)
        if (h[q]) atomicAdd(&counts[g0 + q], h[q]);
}

// pass 2: the same sweep reserves each block's slice of every gene list, then writes key = (ordered value bits << 32) | cluster
__global__ __launch_bounds__(MG_THREADS) void mg_fill_kernel(const float *__restrict__ X, int m, long long n, long long ld, int cells_per_block,
                                                             const int *__restrict__ label, const unsigned long long *__restrict__ offsets,
                                                             unsigned int *__restrict__ cursor, unsigned long long *__restrict__ keys) {
    __shared__ unsigned int h[MG_TILE];
    const int g0 = blockIdx.y * MG_TILE, gn = min(MG_TILE, m - g0);
    for (int q = threadIdx.x; q < gn; q += MG_THREADS) h[q] = 0u;
    __syncthreads();
    const long long c0 = static_cast<long long>(blockIdx.x) * cells_per_block, c1 = min(n, c0 + cells_per_block);
    for (long long c = c0; c < c1; ++c) {
        const float *row = X + c * ld + g0;
        for (int q = threadIdx.x; q < gn; q += MG_THREADS)
            if (row[q] != 0.0f) atomicAdd(&h[q], 1u);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < gn; q += MG_THREADS) {      // h[q] becomes the block's first slot in gene q's list
        const unsigned int cnt = h[q];
        h[q] = cnt ? atomicAdd(&cursor[g0 + q], cnt) : 0u;
    }
    __syncthreads();
    for (long long c = c0; c < c1; ++c) {
        const float *row = X + c * ld + g0;
        const unsigned long long lab = static_cast<unsigned long long>(static_cast<unsigned int>(label[c] - 1));
        for (int q = threadIdx.x; q < gn; q += MG_THREADS) {
            const float v = row[q];
            if (v != 0.0f) {
                const unsigned int slot = atomicAdd(&h[q], 1u);
                keys[offsets[g0 + q] + slot] = (static_cast<unsigned long long>(mg_key(v)) << 32) | lab;
            }
        }
    }
}

// R's pnorm through erfc (two-sided p of wilcox.test: 2 * min(pnorm(z), 1 - pnorm(z)) = erfc(|z| / sqrt 2))
__device__ __forceinline__ double mg_two_sided_p(double z) { return erfc(fabs(z) * 0.70710678118654752440); }

// pass 3: one workgroup per gene over its sorted list
__global__ __launch_bounds__(256) void mg_stats_kernel(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ offsets,
                                                       int m, long long n, int G, const long long *__restrict__ csize, double theta, int rr,
                                                       double *__restrict__ out) {
    const int g = blockIdx.x;
    const unsigned long long *k = keys + offsets[g];
    const long long nz = static_cast<long long>(offsets[g + 1] - offsets[g]);
    double *o = out + static_cast<size_t>(g) * 5;
    const double dp = static_cast<double>(nz) / static_cast<double>(n);
    if (!(dp > theta)) {                                       // R/get_marker_genes.R:123,146-149
        if (threadIdx.x == 0) { o[0] = 0.0; o[1] = 0.0; o[2] = 1.0; o[3] = dp; o[4] = 0.0; }
        return;
    }
    __shared__ long long s2rank[MG_MAXG];                       // 2 * (sum of the ranks of the NON-ZERO cells of cluster c)
    __shared__ double sumx[MG_MAXG];
    __shared__ unsigned int nzc[MG_MAXG];
    __shared__ long long tie3;                                  // sum over tie groups of t^3 - t (non-zero groups)
    __shared__ long long nneg;                                  // negative entries (they rank below the zeros)
    for (int c = threadIdx.x; c < G; c += 256) { s2rank[c] = 0; sumx[c] = 0.0; nzc[c] = 0u; }
    if (threadIdx.x == 0) { tie3 = 0; nneg = 0; }
    __syncthreads();
    const long long t0 = n - nz;                                // zeros: one tie group
    const uint32_t zero_key = 0x80000000u;
    for (long long i = threadIdx.x; i < nz; i += 256) {
        const unsigned long long ki = k[i];
        const uint32_t kv = static_cast<uint32_t>(ki >> 32);
        const int c = static_cast<int>(ki & 0xffffffffu);
        long long lo = 0, hi = nz;                              // first index with value >= v; first with value > v
        { long long a = 0, b = i; while (a < b) { const long long mid = (a + b) >> 1; if (static_cast<uint32_t>(k[mid] >> 32) < kv) a = mid + 1; else b = mid; } lo = a; }
        { long long a = i + 1, b = nz; while (a < b) { const long long mid = (a + b) >> 1; if (static_cast<uint32_t>(k[mid] >> 32) <= kv) a = mid + 1; else b = mid; } hi = a; }
        const long long shift = kv > zero_key ? t0 : 0;        // positive values rank above the zeros
        const long long r2 = 2 * shift + lo + hi + 1;          // 2 * average rank of the tie group [lo, hi)
        atomicAdd(reinterpret_cast<unsigned long long *>(&s2rank[c]), static_cast<unsigned long long>(r2));
        atomicAdd(&sumx[c], static_cast<double>(mg_unkey(kv)));
        atomicAdd(&nzc[c], 1u);
        if (lo == i) { const long long t = hi - lo; atomicAdd(reinterpret_cast<unsigned long long *>(&tie3), static_cast<unsigned long long>(t * t * t - t)); }
        if (kv < zero_key && lo == i) atomicAdd(reinterpret_cast<unsigned long long *>(&nneg), static_cast<unsigned long long>(hi - lo));
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // ---- per-cluster means, candidates, AUROC, Wilcoxon, fold change: a few hundred scalar operations
    const double zero_rank2 = static_cast<double>(2 * nneg + t0 + 1);            // 2 * average rank of the zeros
    double best_auc = -1.0; int best_c = -1;
    // order(-mean rank)[1:rr]: repeatedly take the largest mean rank not taken yet (ties: lowest cluster first)
    unsigned long long taken_lo = 0ull;                                          // bitmap for the first 64 picks is not enough in general:
    (void)taken_lo;                                                              // use sumx sign trick instead: mark with nzc = 0xffffffff
    for (int pick = 0; pick < rr && pick < G; ++pick) {
        int arg = -1; double top = -1.0;
        for (int c = 0; c < G; ++c) {
            if (nzc[c] == 0xffffffffu) continue;
            const double nc = static_cast<double>(csize[c]);
            const double zc = nc - static_cast<double>(nzc[c]);
            const double mr = (0.5 * static_cast<double>(s2rank[c]) + 0.5 * zero_rank2 * zc) / nc;
            if (mr > top) { top = mr; arg = c; }
        }
        if (arg < 0) break;
        const double n1 = static_cast<double>(csize[arg]), n2 = static_cast<double>(n) - n1;
        const double R1 = top * n1;
        const double auc = (R1 - n1 * (n1 + 1.0) / 2.0) / (n1 * n2);
        if (auc > best_auc) { best_auc = auc; best_c = arg; }                     // which.max: first maximum
        // remember the mean expression before marking the cluster as taken
        sumx[arg] = sumx[arg] / n1;
        s2rank[arg] = static_cast<long long>(nzc[arg]);                           // keep the count (needed below)
        nzc[arg] = 0xffffffffu;
    }
    // mean expression of every cluster (taken ones already divided)
    double y1 = 0.0, y2 = -1.0e300;
    for (int c = 0; c < G; ++c) {
        const double mx = nzc[c] == 0xffffffffu ? sumx[c] : sumx[c] / static_cast<double>(csize[c]);
        if (c == best_c) y1 = mx; else if (mx > y2) y2 = mx;
    }
    const double n1 = static_cast<double>(csize[best_c]), n2 = static_cast<double>(n) - n1, nn = static_cast<double>(n);
    const double W = best_auc * n1 * n2;                                          // sum of ranks - n1 (n1 + 1) / 2
    double z = W - n1 * n2 / 2.0;
    const double ties = static_cast<double>(tie3) + (static_cast<double>(t0) * t0 * t0 - static_cast<double>(t0));
    const double sigma = sqrt((n1 * n2 / 12.0) * ((nn + 1.0) - ties / (nn * (nn - 1.0))));
    const double corr = z > 0.0 ? 0.5 : (z < 0.0 ? -0.5 : 0.0);
    z = (z - corr) / sigma;
    o[0] = best_auc; o[1] = static_cast<double>(best_c + 1); o[2] = mg_two_sided_p(z); o[3] = dp; o[4] = y1 / y2;
}

// the same two passes over a block held as the slots of a dgCMatrix (colptr / rowidx / val of the block's cells, resident): one wave per
// cell walks the cell's stored entries -- no dense block is ever built (R/get_marker_genes_unlimited.R:44-48 reads a@i the same way)
__global__ void mg_count_csc_kernel(const long long *__restrict__ colptr, const int *__restrict__ rowidx, const double *__restrict__ val,
                                    long long ncell, int m, unsigned int *__restrict__ counts, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) >> 6;
    const long long nwave = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
    for (long long c = wave; c < ncell; c += nwave)
        for (long long e = colptr[c] + lane; e < colptr[c + 1]; e += 64) {
            const int g = rowidx[e];
            if (g < 0 || g >= m) { atomicAdd(bad, 1); continue; }
            if (static_cast<float>(val[e]) != 0.0f) atomicAdd(&counts[g], 1u);
        }
}
__global__ void mg_fill_csc_kernel(const long long *__restrict__ colptr, const int *__restrict__ rowidx, const double *__restrict__ val,
                                   long long ncell, int m, const int *__restrict__ label, const unsigned long long *__restrict__ offsets,
                                   unsigned int *__restrict__ cursor, unsigned long long *__restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) >> 6;
    const long long nwave = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
    for (long long c = wave; c < ncell; c += nwave) {
        const unsigned long long lab = static_cast<unsigned long long>(static_cast<unsigned int>(label[c] - 1));
        for (long long e = colptr[c] + lane; e < colptr[c + 1]; e += 64) {
            const int g = rowidx[e];
            const float v = static_cast<float>(val[e]);
            if (g < 0 || g >= m || v == 0.0f) continue;
            const unsigned int slot = atomicAdd(&cursor[g], 1u);
            keys[offsets[g] + slot] = (static_cast<unsigned long long>(mg_key(v)) << 32) | lab;
        }
    }
}

// One block of a list: a dense resident fp32 block (cells x ld), or the resident slots of a sparse one.
struct MgBlock {
    const float *dX = nullptr; long long ld = 0;                                  // dense
    const long long *colptr = nullptr; const int *rowidx = nullptr; const double *val = nullptr;   // sparse
    long long n = 0;
};

// get_marker_genes' per-gene pass (R/get_marker_genes.R:120-152) over the cells of ALL blocks (R/get_marker_genes_unlimited.R:95-118:
// dd = the gene's values across every block): every block adds its non-zeros to the per-gene lists, one sort, one statistics pass.
void marker_genes_blocks_dev(const std::vector<MgBlock> &blocks, int m, const int *h_label, int G, double theta, int ng, double *h_out) {
    Ctx &c = ctx();
    SHARP_REQUIRE(G >= 2 && G <= MG_MAXG, "get_marker_genes: between 2 and 1024 clusters are supported");
    long long n = 0;
    for (const MgBlock &b : blocks) n += b.n;
    SHARP_REQUIRE(n >= 2 && m >= 1 && !blocks.empty(), "get_marker_genes: empty input");
    std::vector<long long> csize(G, 0);
    for (long long i = 0; i < n; ++i) {
        SHARP_REQUIRE(h_label[i] >= 1 && h_label[i] <= G, "get_marker_genes: labels must be 1..N.pred_cluster");
        ++csize[h_label[i] - 1];
    }
    for (int q = 0; q < G; ++q) SHARP_REQUIRE(csize[q] > 0, "get_marker_genes: empty cluster");
    DevBuf<int> d_label(n);
    d_label.upload(h_label, n);
    DevBuf<long long> d_csize(G);
    d_csize.upload(csize.data(), G);
    DevBuf<unsigned int> d_counts(m), d_cursor(m);
    DevBuf<int> d_bad(1);
    d_counts.zero(); d_cursor.zero(); d_bad.zero();
    auto dense_grid = [&](const MgBlock &b, int &cells_per_block) {
        cells_per_block = static_cast<int>(std::max<long long>(16, (b.n + 4 * c.num_cu - 1) / (4 * c.num_cu)));
        return dim3(static_cast<unsigned>((b.n + cells_per_block - 1) / cells_per_block), (m + MG_TILE - 1) / MG_TILE);
    };
    const int csc_blocks = c.num_cu * 16;
    {
        KernelTimer t("marker_count");
        for (const MgBlock &b : blocks) {
            if (b.dX) {
                int cpb = 0;
                const dim3 grid = dense_grid(b, cpb);
                hipLaunchKernelGGL(mg_count_kernel, grid, dim3(MG_THREADS), 0, c.stream, b.dX, m, b.n, b.ld, cpb, d_counts.p);
            } else {
                hipLaunchKernelGGL(mg_count_csc_kernel, dim3(csc_blocks), dim3(256), 0, c.stream, b.colptr, b.rowidx, b.val, b.n, m, d_counts.p, d_bad.p);
            }
            launch_check("mg_count_kernel");
        }
    }
    std::vector<unsigned int> cnt(m);
    int bad = 0;
    SHARP_HIP_CHECK(hipMemcpyAsync(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    d_counts.download(cnt.data(), m);
    SHARP_REQUIRE(bad == 0, "sparse input: row index outside [0, genes)");
    std::vector<unsigned long long> off(static_cast<size_t>(m) + 1, 0);
    for (int g = 0; g < m; ++g) off[g + 1] = off[g] + cnt[g];
    const unsigned long long nnz = off[m];
    SHARP_REQUIRE(nnz < (1ull << 32), "get_marker_genes: more than 2^32 - 1 non-zero values (the segmented sort's index type)");
    DevBuf<unsigned long long> d_off(off.size());
    d_off.upload(off.data(), off.size());
    DevBuf<unsigned long long> d_keys(std::max<unsigned long long>(nnz, 1)), d_sorted(std::max<unsigned long long>(nnz, 1));
    {
        KernelTimer t("marker_fill");
        long long c0 = 0;
        for (const MgBlock &b : blocks) {
            if (b.dX) {
                int cpb = 0;
                const dim3 grid = dense_grid(b, cpb);
                hipLaunchKernelGGL(mg_fill_kernel, grid, dim3(MG_THREADS), 0, c.stream, b.dX, m, b.n, b.ld, cpb, d_label.p + c0, d_off.p, d_cursor.p,
                                   d_keys.p);
            } else {
                hipLaunchKernelGGL(mg_fill_csc_kernel, dim3(csc_blocks), dim3(256), 0, c.stream, b.colptr, b.rowidx, b.val, b.n, m, d_label.p + c0,
                                   d_off.p, d_cursor.p, d_keys.p);
            }
            launch_check("mg_fill_kernel");
            c0 += b.n;
        }
    }
    if (nnz > 0) {
        KernelTimer t("marker_sort");
        size_t temp_bytes = 0;
        SHARP_HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, d_keys.p, d_sorted.p, static_cast<unsigned int>(nnz),
                                                           static_cast<unsigned int>(m), d_off.p, d_off.p + 1, 0, 64, c.stream));
        DevBuf<unsigned char> temp(temp_bytes + 16);
        SHARP_HIP_CHECK(rocprim::segmented_radix_sort_keys(temp.p, temp_bytes, d_keys.p, d_sorted.p, static_cast<unsigned int>(nnz),
                                                           static_cast<unsigned int>(m), d_off.p, d_off.p + 1, 0, 64, c.stream));
        stream_sync();
    }
    DevBuf<double> d_out(static_cast<size_t>(m) * 5);
    {
        KernelTimer t("marker_stats");
        const int rr = std::max(1, std::min(ng, G));
        hipLaunchKernelGGL(mg_stats_kernel, dim3(m), dim3(256), 0, c.stream, d_sorted.p, d_off.p, m, n, G, d_csize.p, theta, rr, d_out.p);
        launch_check("mg_stats_kernel");
    }
    d_out.download(h_out, static_cast<size_t>(m) * 5);
}

void marker_genes_dev(const float *dX, int m, long long n, long long ld, const int *h_label, int G, double theta, int ng, double *h_out) {
    MgBlock b;
    b.dX = dX; b.ld = ld; b.n = n;
    marker_genes_blocks_dev(std::vector<MgBlock>(1, b), m, h_label, G, theta, ng, h_out);
}

}  // namespace sharp

using namespace sharp;

extern "C" {

int sharp_marker_genes_dev(const float *dX, int m, long long n, long long ld, const int *label, int n_cluster, double theta, int ng,
                           double *out) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(dX && label && out, "sharp_marker_genes_dev: null argument");
    SHARP_REQUIRE(static_cast<unsigned long long>(n) * static_cast<unsigned long long>(m) < (1ull << 40), "get_marker_genes: matrix too large");
    marker_genes_dev(dX, m, n, ld, label, n_cluster, theta, ng, out);
    SHARP_API_END
}

/* the per-gene pass over the cells of a LIST of blocks (R/get_marker_genes_unlimited.R:95-118): resident fp32 blocks */
int sharp_marker_genes_blocks_dev(const float *const *dX_blocks, const long long *ncb, const long long *ldb, int nblocks, int m,
                                  const int *label, int n_cluster, double theta, int ng, double *out) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(dX_blocks && ncb && ldb && label && out && nblocks >= 1, "sharp_marker_genes_blocks_dev: null argument");
    std::vector<MgBlock> v(static_cast<size_t>(nblocks));
    for (int b = 0; b < nblocks; ++b) { SHARP_REQUIRE(dX_blocks[b] && ldb[b] >= m && ncb[b] >= 0, "sharp_marker_genes_blocks_dev: bad block"); v[b].dX = dX_blocks[b]; v[b].ld = ldb[b]; v[b].n = ncb[b]; }
    marker_genes_blocks_dev(v, m, label, n_cluster, theta, ng, out);
    SHARP_API_END
}

/* the same for a list of dgCMatrix blocks on the host (colptr[b]: ncb[b] + 1 ints, rowidx[b] 0-based, val[b]): the stored entries are
 * uploaded as they are and scattered straight into the per-gene lists -- no dense block is built */
int sharp_marker_genes_blocks_csc(const int *const *colptr, const int *const *rowidx, const double *const *val, const long long *ncb,
                                  int nblocks, int m, const int *label, int n_cluster, double theta, int ng, double *out) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(colptr && rowidx && val && ncb && label && out && nblocks >= 1, "sharp_marker_genes_blocks_csc: null argument");
    std::vector<MgBlock> v(static_cast<size_t>(nblocks));
    std::vector<DevBuf<long long>> dcp(static_cast<size_t>(nblocks));
    std::vector<DevBuf<int>> dri(static_cast<size_t>(nblocks));
    std::vector<DevBuf<double>> dvx(static_cast<size_t>(nblocks));
    std::vector<std::vector<long long>> cp(static_cast<size_t>(nblocks));
    for (int b = 0; b < nblocks; ++b) {
        const long long n = ncb[b];
        SHARP_REQUIRE(colptr[b] && n >= 0, "sparse input: null pointer");
        cp[b].resize(static_cast<size_t>(n) + 1);
        for (long long q = 0; q <= n; ++q) { cp[b][q] = static_cast<long long>(colptr[b][q]) - colptr[b][0]; SHARP_REQUIRE(q == 0 || cp[b][q] >= cp[b][q - 1], "sparse input: column pointers must be non-decreasing"); }
        const long long ne = cp[b][n];
        SHARP_REQUIRE(ne == 0 || (rowidx[b] && val[b]), "sparse input: null pointer");
        dcp[b].alloc(cp[b].size()); dcp[b].upload(cp[b].data(), cp[b].size());
        dri[b].alloc(static_cast<size_t>(std::max<long long>(ne, 1))); dvx[b].alloc(static_cast<size_t>(std::max<long long>(ne, 1)));
        if (ne) { dri[b].upload(rowidx[b] + colptr[b][0], static_cast<size_t>(ne)); dvx[b].upload(val[b] + colptr[b][0], static_cast<size_t>(ne)); }
        v[b].colptr = dcp[b].p; v[b].rowidx = dri[b].p; v[b].val = dvx[b].p; v[b].n = n;
    }
    stream_sync();
    marker_genes_blocks_dev(v, m, label, n_cluster, theta, ng, out);
    SHARP_API_END
}

int sharp_marker_genes(const double *X, int m, long long n, long long ld, const int *label, int n_cluster, double theta, int ng,
                       double *out) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(X && label && out, "sharp_marker_genes: null argument");
    const long long ldd = (static_cast<long long>(m) + 3) / 4 * 4;
    std::vector<float> h(static_cast<size_t>(ldd) * n, 0.0f);
    for (long long c2 = 0; c2 < n; ++c2)
        for (int g = 0; g < m; ++g) h[c2 * ldd + g] = static_cast<float>(X[c2 * ld + g]);
    DevBuf<float> dX(h.size());
    dX.upload(h.data(), h.size());
    stream_sync();
    marker_genes_dev(dX.p, m, n, ldd, label, n_cluster, theta, ng, out);
    SHARP_API_END
}

}  // extern "C"
