// meta.hip -- meta-clustering stages on the GPU.
//   wMetaC (R/wMetaC.R:15-226): co-association point weights w1 (:24-44, an N x N x C contraction),
//     weighted-Jaccard cluster similarity S (:60-77, getss :299-311) as a fused
//     "intersection / union of point weights" kernel, get_opt_hclust(S) on the batched GPU path,
//     then the per-cell vote (:141-161) and soft matrix x0 (:180-208) on the host (O(N*C) strings-as-ints).
//   sMetaC (R/sMetaC.R:17-210): per-label centroid means of the projected space (:58-63),
//     centroid correlation S (:67-85, fp64 MFMA), get_opt_hclust(S), second-best override (:139-148).
#include "meta.hpp"

#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "linalg.hpp"

namespace sharp {

struct WmMeta {
    int N, C, allC, pad;
    long long oCid;    // uint16 N*C (row-major: cell, column)
    long long oW;      // doubles N
    long long oS;      // doubles allC*allC
    long long oCol;    // ints allC: column of each global cluster id
    long long oStart;  // ints allC + 1: first member of each global cluster id in the member list (offsets into the task's block)
};

// ---- point weights: w0_i = 4/N * sum_j AA_ij (1 - AA_ij), AA_ij = #{c: lab_ic == lab_jc} / C; ascending j
constexpr int WW_THREADS = 1024;   // 16 waves, one cell i per wave; the fold's label table is shared through LDS

// w0_i = (4/N) sum_j AA_ij (1 - AA_ij), AA_ij = (number of the C clusterings that put i and j together) / C
// (R/wMetaC.R:24-44).  One wave per cell i, lanes over the other cells j: the labels are held column-wise
// (clustering-major) in LDS so that lanes read consecutive cells; each lane adds x(1-x) for its cells in ascending j and
// the 64 partial sums are combined in a fixed order.
__global__ __launch_bounds__(WW_THREADS) void wm_weights_kernel(const WmMeta *__restrict__ metas, const uint16_t *__restrict__ cid_all,
                                                                double *__restrict__ w_all) {
    const WmMeta M = metas[blockIdx.y];
    const int N = M.N, C = M.C;
    constexpr int NW = WW_THREADS / 64;
    if (blockIdx.x * NW >= N) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint16_t *labt = reinterpret_cast<uint16_t *>(sm);                // [C][N]  (cid is [N][C])
    double *tab = reinterpret_cast<double *>(sm + ((static_cast<size_t>(N) * C * 2 + 15) & ~static_cast<size_t>(15)));   // C+1
    const uint16_t *cid = cid_all + M.oCid;
    for (int q = threadIdx.x; q < N * C; q += WW_THREADS) { const int j = q / C, c = q - j * C; labt[c * N + j] = cid[q]; }
    for (int q = threadIdx.x; q <= C; q += WW_THREADS) { const double x = static_cast<double>(q) / static_cast<double>(C); tab[q] = x * (1 - x); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * NW + wave;
    if (i >= N) return;
    double rs = 0.0;
    for (int j = lane; j < N; j += 64) {
        int cnt = 0;
        for (int c = 0; c < C; ++c) cnt += (labt[c * N + j] == labt[c * N + i]);
        rs += tab[cnt];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rs += __shfl_xor(rs, o);
    if (lane == 0) {
        const double w0 = 4.0 / N * rs;
        w_all[M.oW + i] = (w0 + 0.01) / (1 + 0.01);
    }
}

// ---- S[a][b] = sum(w1[a n b]) / sum(w1[a u b]) for a < b, in R's set orders:
// intersection ascending over a; union = all of a ascending, then the members of b not in a.
__global__ __launch_bounds__(256) void wm_similarity_kernel(const WmMeta *__restrict__ metas, const uint16_t *__restrict__ cid_all,
                                                            const double *__restrict__ w_all, const int *__restrict__ col_all,
                                                            const uint32_t *__restrict__ mem_all, const int *__restrict__ start_all,
                                                            double *__restrict__ S_all) {
    const WmMeta M = metas[blockIdx.y];
    const int a = blockIdx.x;
    if (a >= M.allC) return;
    const int C = M.C, allC = M.allC;
    const uint16_t *cid = cid_all + M.oCid;
    const double *w = w_all + M.oW;
    const int *colof = col_all + M.oCol;
    // the members of every cluster, ascending by cell (built with the relabelling): the sums below visit the members of a and of b
    // only -- the same terms in the same order as the scan over all N cells this kernel used to do (0.69 -> 0.34 ms; staging the
    // members of a in LDS on top: 0.37)
    const uint32_t *mem = mem_all + M.oCid;
    const int *ms = start_all + M.oStart;
    double *S = S_all + M.oS;
    const int ca = colof[a];
    const int a0 = ms[a], a1 = ms[a + 1];
    if (threadIdx.x == 0) S[static_cast<long long>(a) * allC + a] = 1.0;
    for (int b = a + 1 + threadIdx.x; b < allC; b += 256) {
        const int cb = colof[b];
        double ss = 0.0;
        if (cb != ca) {
            double inter = 0.0, uni = 0.0;
            int ni = 0;
            for (int q = a0; q < a1; ++q) {
                const uint32_t i = mem[q];
                const double wi = w[i];
                uni += wi;
                if (cid[static_cast<size_t>(i) * C + cb] == b) { inter += wi; ++ni; }
            }
            if (ni) {
                for (int q = ms[b]; q < ms[b + 1]; ++q) {
                    const uint32_t i = mem[q];
                    if (cid[static_cast<size_t>(i) * C + ca] != a) uni += w[i];
                }
                ss = inter / uni;
            }
        }
        S[static_cast<long long>(a) * allC + b] = ss;
        S[static_cast<long long>(b) * allC + a] = ss;
    }
}

// ---- per-label column sums: one wave-column-chunk per cluster, members in ascending cell order
// colMeans(sE1[cluster, ]) (R/sMetaC.R:58-63).  One workgroup per (cluster, 64-column slab): eight waves take the members
// round-robin with four loads in flight each (a big final cluster has thousands of members: one sequential chain of
// dependent loads took 4 ms per call), and the 32 partial sums are added in a fixed order.
constexpr int CM_WAVES = 8;
__global__ __launch_bounds__(64 * CM_WAVES) void cluster_means_kernel(const double *__restrict__ E, long long ld, int p,
                                                                      const int *__restrict__ start, const int *__restrict__ members,
                                                                      double *__restrict__ means) {
    __shared__ double part[CM_WAVES][64];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + lane;
    const int s0 = start[t], s1 = start[t + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (col < p) {
        int q = s0 + w;
        for (; q + 3 * CM_WAVES < s1; q += 4 * CM_WAVES) {
            const double x0 = E[static_cast<long long>(members[q]) * ld + col];
            const double x1 = E[static_cast<long long>(members[q + CM_WAVES]) * ld + col];
            const double x2 = E[static_cast<long long>(members[q + 2 * CM_WAVES]) * ld + col];
            const double x3 = E[static_cast<long long>(members[q + 3 * CM_WAVES]) * ld + col];
            a0 += x0; a1 += x1; a2 += x2; a3 += x3;
        }
        for (; q < s1; q += CM_WAVES) a0 += E[static_cast<long long>(members[q]) * ld + col];
    }
    part[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && col < p) {
        double acc = 0.0;
        for (int q = 0; q < CM_WAVES; ++q) acc += part[q][lane];
        means[static_cast<long long>(t) * p + col] = acc / static_cast<double>(s1 - s0);
    }
}

__global__ void ensemble_mean_kernel(const double *__restrict__ E, long long ldE, int n, int p, int K, double *__restrict__ viE) {
    const long long tot = static_cast<long long>(n) * p;
    for (long long q = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; q < tot;
         q += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long i = q / p;
        const int c = static_cast<int>(q - i * p);
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += E[i * ldE + static_cast<long long>(k) * p + c];   // enE += pE1, k ascending
        viE[q] = s / K;
    }
}

// ------------------------------------------------------------------------------------------------
namespace {

inline bool lex_less(int a, int b) {   // order of R's table() levels: the ids as character strings
    char sa[16], sb[16];
    snprintf(sa, sizeof sa, "%d", a);
    snprintf(sb, sizeof sb, "%d", b);
    return strcmp(sa, sb) < 0;
}

struct MetaWs {
    uint16_t *h_cid = nullptr;       // pinned, grow-only: the relabelled ensembles are written here and uploaded from here
    uint32_t *h_mem = nullptr;       // pinned, same count: the cells of every cluster, ascending (member lists)
    size_t h_cid_n = 0;
    DevBuf<uint32_t> mem;
    DevBuf<int> mstart;
    DevBuf<uint16_t> cid;
    DevBuf<double> w, S, means, U, Ut, nrm, Smat;
    DevBuf<int> col, start, members;
    DevBuf<WmMeta> meta;
    DevBuf<RowPrepTask> prep;
    DevBuf<GemmTask> gemm;
};
MetaWs &mws() { return per_slot<MetaWs>(); }

}  // namespace

int first_appearance_ids(const int *labels, long long n, std::vector<int> &uid) {
    uid.resize(n);
    std::unordered_map<int, int> map;
    map.reserve(1024);
    int nu = 0;
    for (long long i = 0; i < n; ++i) {
        auto it = map.find(labels[i]);
        if (it == map.end()) { map.emplace(labels[i], nu); uid[i] = nu++; }
        else uid[i] = it->second;
    }
    return nu;
}

void wmetac_batch(const std::vector<WmTask> &tasks, bool want_x0, bool want_debug, std::vector<WmResult> &out) {
    const int T = static_cast<int>(tasks.size());
    out.assign(T, WmResult());
    if (!T) return;
    Ctx &c = ctx();
    MetaWs &W = mws();
    // R = unique(x): global cluster ids in column-major first-appearance order (R/wMetaC.R:60-67)
    std::vector<WmMeta> metas(T);
    // the relabelled label matrices of all folds, back to back, in pinned memory (offsets known before the relabelling)
    std::vector<long long> cid_off(T + 1, 0);
    for (int t = 0; t < T; ++t) cid_off[t + 1] = cid_off[t] + static_cast<long long>(std::max(tasks[t].N, 0)) * std::max(tasks[t].C, 0);
    if (static_cast<size_t>(cid_off[T]) > W.h_cid_n) {
        if (W.h_cid) { (void)hipHostFree(W.h_cid); W.h_cid = nullptr; W.h_cid_n = 0; }
        if (W.h_mem) { (void)hipHostFree(W.h_mem); W.h_mem = nullptr; }
        SHARP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&W.h_cid), static_cast<size_t>(cid_off[T]) * sizeof(uint16_t), hipHostMallocDefault));
        SHARP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&W.h_mem), static_cast<size_t>(cid_off[T]) * sizeof(uint32_t), hipHostMallocDefault));
        W.h_cid_n = static_cast<size_t>(cid_off[T]);
    }
    uint16_t *const hc = W.h_cid;
    uint32_t *const hm = W.h_mem;
    std::vector<std::vector<int>> mstarts(T);            // per fold: first member of every global cluster id, + the end
    std::vector<std::vector<int>> colof(T);
    long long oCid = 0, oW = 0, oS = 0, oCol = 0, oStart = 0;
    int maxN = 0, maxAll = 0;
    size_t max_lds = 0;
    // per-fold relabelling on host threads (25 folds x 15 clusterings x 2000 cells at cfg2: 0.8 ms on one thread)
    std::vector<int> allCs(T, 0), err(T, 0);
    auto relabel = [&](int t) {
        const WmTask &tk = tasks[t];
        if (!(tk.N >= 2 && tk.C >= 1 && tk.nC)) { err[t] = 1; return; }
        uint16_t *cid = hc + cid_off[t];
        int allC = 0;
        std::vector<int> uid, pos;
        for (int col = 0; col < tk.C; ++col) {
            const int nu = first_appearance_ids(tk.nC + static_cast<size_t>(col) * tk.N, tk.N, uid);
            if (allC + nu > 65535) { err[t] = 2; return; }
            for (int i = 0; i < tk.N; ++i) cid[static_cast<size_t>(i) * tk.C + col] = static_cast<uint16_t>(allC + uid[i]);
            for (int q = 0; q < nu; ++q) colof[t].push_back(col);
            // the column's cells by cluster, ascending inside a cluster (counting sort): block [col * N, (col + 1) * N) of the list
            pos.assign(static_cast<size_t>(nu) + 1, 0);
            for (int i = 0; i < tk.N; ++i) ++pos[uid[i] + 1];
            for (int q = 0; q < nu; ++q) pos[q + 1] += pos[q];
            for (int q = 0; q < nu; ++q) mstarts[t].push_back(col * tk.N + pos[q]);
            uint32_t *mcol = hm + cid_off[t] + static_cast<long long>(col) * tk.N;
            for (int i = 0; i < tk.N; ++i) mcol[pos[uid[i]]++] = static_cast<uint32_t>(i);
            allC += nu;
        }
        mstarts[t].push_back(tk.C * tk.N);
        allCs[t] = allC;
    };
    {
        HostTimer ht("wmetac_relabel");
        host_parallel_for(T, 16, relabel);
    }
    for (int t = 0; t < T; ++t) {
        const WmTask &tk = tasks[t];
        SHARP_REQUIRE(err[t] != 1, "wMetaC: empty label matrix");
        SHARP_REQUIRE(err[t] != 2, "wMetaC: more than 65535 base clusters");
        const int allC = allCs[t];
        SHARP_REQUIRE(allC >= 3, "wMetaC: fewer than 3 base clusters in the ensemble");
        WmMeta &M = metas[t];
        M.N = tk.N; M.C = tk.C; M.allC = allC; M.pad = 0;
        M.oCid = oCid; oCid += static_cast<long long>(tk.N) * tk.C;
        M.oW = oW; oW += tk.N;
        M.oS = oS; oS += static_cast<long long>(allC) * allC;
        M.oCol = oCol; oCol += allC;
        M.oStart = oStart; oStart += allC + 1;
        maxN = std::max(maxN, tk.N); maxAll = std::max(maxAll, allC);
        max_lds = std::max(max_lds, ((static_cast<size_t>(tk.N) * tk.C * 2 + 15) & ~static_cast<size_t>(15)) + (tk.C + 1) * 8);
        out[t].allC = allC;
    }
    SHARP_REQUIRE(max_lds <= 150 * 1024, "wMetaC: N x C label block does not fit in LDS");
    W.cid.ensure(oCid); W.w.ensure(oW); W.S.ensure(oS); W.col.ensure(oCol); W.meta.ensure(T); W.mem.ensure(oCid); W.mstart.ensure(oStart);
    {
        HostTimer ht("wmetac_upload");
        std::vector<int> hcol(oCol);
        std::vector<int> hstart(oStart);
        for (int t = 0; t < T; ++t) {
            std::copy(colof[t].begin(), colof[t].end(), hcol.begin() + metas[t].oCol);   // (metas[t].oCid == cid_off[t])
            std::copy(mstarts[t].begin(), mstarts[t].end(), hstart.begin() + metas[t].oStart);
        }
        W.cid.upload(hc, oCid);
        W.mem.upload(hm, oCid);
        W.mstart.upload(hstart.data(), oStart);
        W.col.upload(hcol.data(), oCol);
        W.meta.upload(metas.data(), T);
        stream_sync();
    }
    {
        SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wm_weights_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(max_lds)));
        KernelTimer tm("wmetac_weights");
        hipLaunchKernelGGL(wm_weights_kernel, dim3((maxN + WW_THREADS / 64 - 1) / (WW_THREADS / 64), T), dim3(WW_THREADS), max_lds, c.stream, W.meta.p,
                           W.cid.p, W.w.p);
        launch_check("wm_weights_kernel");
    }
    {
        KernelTimer tm("wmetac_similarity");
        hipLaunchKernelGGL(wm_similarity_kernel, dim3(maxAll, T), dim3(256), 0, c.stream, W.meta.p, W.cid.p, W.w.p, W.col.p, W.mem.p, W.mstart.p, W.S.p);
        launch_check("wm_similarity_kernel");
    }
    // hres = get_opt_hclust(S, ...)  (R/wMetaC.R:98-99)
    std::vector<HcTask> hts(T);
    for (int t = 0; t < T; ++t) {
        hts[t].d_mat = W.S.p + metas[t].oS;
        hts[t].ld = metas[t].allC; hts[t].n = metas[t].allC; hts[t].p = metas[t].allC;
        hts[t].symmetric = true;
        hts[t].prm = tasks[t].prm;
    }
    std::vector<HcResult> hres;
    get_opt_hclust_batch(hts, false, hres);
    std::vector<double> h_w, h_S;
    if (want_debug) {
        h_w.resize(oW); h_S.resize(oS);
        W.w.download(h_w.data(), oW);
        W.S.download(h_S.data(), oS);
    }
    // per-cell vote and soft matrix on the host
    HostTimer ht_vote("wmetac_vote");
    // the folds are independent: vote them on a few host threads (each writes only its own result)
    auto vote_fold = [&](int t) {
        const WmTask &tk = tasks[t];
        WmResult &R = out[t];
        const int N = tk.N, C = tk.C;
        const std::vector<int> &tf = hres[t].f;
        R.rc = hres[t].rc;
        if (want_debug) {
            R.w1.assign(h_w.begin() + metas[t].oW, h_w.begin() + metas[t].oW + N);
            R.S.assign(h_S.begin() + metas[t].oS, h_S.begin() + metas[t].oS + static_cast<long long>(R.allC) * R.allC);
            R.tf = tf;
        }
        const uint16_t *cid = hc + cid_off[t];
        R.finalC.resize(N);
        std::vector<int> second(N, -1), uv(C), uc(C);
        // rank of every meta id in R's table() level order (ids compared as character strings)
        const int maxid = *std::max_element(tf.begin(), tf.end());
        std::vector<int> order(maxid + 1), lexrank(maxid + 1, 0);
        for (int q = 0; q <= maxid; ++q) order[q] = q;
        std::sort(order.begin() + 1, order.end(), [](int a, int b) { return lex_less(a, b); });
        for (int q = 1; q <= maxid; ++q) lexrank[order[q]] = q;
        auto vote_of = [&](int i, int col) { return tf[cid[static_cast<size_t>(i) * C + col]]; };
        for (int i = 0; i < N; ++i) {
            int nu = 0;
            for (int col = 0; col < C; ++col) {
                const int v = vote_of(i, col);
                int q = 0;
                for (; q < nu; ++q) if (uv[q] == v) break;
                if (q == nu) { uv[nu] = v; uc[nu] = 0; ++nu; }
                ++uc[q];
            }
            // names(sort(table(d), decreasing = TRUE)[1]): most votes, ties -> first level in string order
            int best = -1, sec = -1;
            for (int q = 0; q < nu; ++q)
                if (best < 0 || uc[q] > uc[best] || (uc[q] == uc[best] && lexrank[uv[q]] < lexrank[uv[best]])) best = q;
            for (int q = 0; q < nu; ++q) {
                if (q == best) continue;
                if (sec < 0 || uc[q] > uc[sec] || (uc[q] == uc[sec] && lexrank[uv[q]] < lexrank[uv[sec]])) sec = q;
            }
            R.finalC[i] = uv[best];
            second[i] = sec >= 0 ? uv[sec] : -1;
        }
        auto uniq = [&](std::vector<int> &uC) {
            uC.clear();
            for (int i = 0; i < N; ++i) if (std::find(uC.begin(), uC.end(), R.finalC[i]) == uC.end()) uC.push_back(R.finalC[i]);
        };
        std::vector<int> uC;
        uniq(uC);
        if (uC.size() == 1) {   // R/wMetaC.R:148-161: take the runner-up wherever one exists
            for (int i = 0; i < N; ++i) { if (second[i] >= 0) R.finalC[i] = second[i]; else R.rc |= SHARP_WARN_NA_VOTE; }
            uniq(uC);
        }
        R.ncl = static_cast<int>(uC.size());
        if (want_x0) {
            R.x0.assign(static_cast<size_t>(N) * R.ncl, 0.0);
            for (int i = 0; i < N; ++i) {
                const int xind = static_cast<int>(std::find(uC.begin(), uC.end(), R.finalC[i]) - uC.begin());
                int own = 0;
                for (int col = 0; col < C; ++col) own += (vote_of(i, col) == uC[xind]);
                R.x0[static_cast<size_t>(xind) * N + i] = 1.0;
                for (int q = 0; q < R.ncl; ++q) {
                    if (q == xind) continue;
                    int y = 0;
                    for (int col = 0; col < C; ++col) y += (vote_of(i, col) == uC[q]);
                    if (y) R.x0[static_cast<size_t>(q) * N + i] = 0.5 * static_cast<double>(y) / static_cast<double>(own);
                }
            }
        }
    };
    {
        host_parallel_for(T, T >= 4 ? 16 : 1, vote_fold);
    }
}

void cluster_means_dev(const double *d_E, long long ld, int n, int p, const std::vector<int> &uid, int nC, double *d_means,
                       const int *row_of_cell) {
    Ctx &c = ctx();
    MetaWs &W = mws();
    std::vector<int> start(nC + 1, 0), members(n);
    for (int i = 0; i < n; ++i) ++start[uid[i] + 1];
    for (int t = 0; t < nC; ++t) start[t + 1] += start[t];
    {
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < n; ++i) members[fill[uid[i]]++] = row_of_cell ? row_of_cell[i] : i;   // ascending cell index inside a cluster
    }
    W.start.ensure(nC + 1); W.members.ensure(n);
    W.start.upload(start.data(), nC + 1);
    W.members.upload(members.data(), n);
    KernelTimer tm("smetac_centroids");
    hipLaunchKernelGGL(cluster_means_kernel, dim3(nC, (p + 63) / 64), dim3(64 * CM_WAVES), 0, c.stream, d_E, ld, p, W.start.p, W.members.p,
                       d_means);
    launch_check("cluster_means_kernel");
    stream_sync();
}

SmResult smetac_from_means(const double *d_means, int nC, int p, long long ncells, HcParams prm) {
    SHARP_REQUIRE(nC >= 3, "sMetaC: fewer than 3 clusters to combine");
    MetaWs &W = mws();
    const int nld = (nC + 127) / 128 * 128;
    // S = cor(aG[a,], aG[b,]), diag 1 (R/sMetaC.R:67-85): centre + normalise rows, then one MFMA GEMM
    const int p_pad = (p + 15) / 16 * 16;
    W.U.ensure(static_cast<size_t>(nC) * p); W.Ut.ensure(static_cast<size_t>(p_pad) * nld); W.nrm.ensure(nC);
    W.Smat.ensure(static_cast<size_t>(nC) * nC);
    W.prep.ensure(1); W.gemm.ensure(1);
    RowPrepTask rp{d_means, p, nC, p, nld, p_pad, 0, W.U.p, W.Ut.p, W.nrm.p, nullptr};
    W.prep.upload(&rp, 1);
    row_prep_batched(W.prep.p, 1, nC, p);
    GemmTask g{W.Ut.p, W.Ut.p, W.Smat.p, nC, nC, p, nld, nld, nC, 2, 1, 0};
    W.gemm.upload(&g, 1);
    gemm_tn_f64_batched(W.gemm.p, 1, nC, nC, "smetac_centroid_corr_gemm");
    { HostTimer ht("smetac_corr_wait"); stream_sync(); }
    // k-range adjustment (R/sMetaC.R:103-119)
    const long long mm = ncells / 10000;
    if (ncells < 1000000) {
        const int baseN = static_cast<int>(std::min<long long>(std::max<long long>(mm, 2), 10));
        if (prm.minN == 2 && std::min(prm.maxN, nC) - baseN >= 3) prm.minN = baseN;
    } else {
        const int mm3 = static_cast<int>(ncells / 50000), mm2 = static_cast<int>(ncells / 5000);
        prm.maxN = std::max(prm.maxN, mm2);
        prm.minN = std::max(prm.minN, mm3);
    }
    HcTask t;
    t.d_mat = W.Smat.p; t.ld = nC; t.n = nC; t.p = nC; t.symmetric = true; t.prm = prm;
    std::vector<HcTask> ts{t};
    std::vector<HcResult> hr;
    get_opt_hclust_batch(ts, true, hr);
    HcResult &H = hr[0];
    SmResult R;
    R.rc = H.rc; R.maxsil = H.maxsil;
    R.tf = H.f;
    const int nuf = *std::max_element(H.f.begin(), H.f.end());
    if (H.nk > 1 && nuf == 2 && H.maxsil > prm.sil_thre) {   // R/sMetaC.R:139-148: second largest msil
        std::vector<double> s0(H.msil);
        std::sort(s0.begin(), s0.end());
        const double s1 = s0[H.nk - 2];
        int s2 = 0;
        for (int q = 0; q < H.nk; ++q) if (H.msil[q] == s1) { s2 = q; break; }   // quirk 9: first of tied columns
        R.tf.assign(H.v.begin() + static_cast<size_t>(s2) * nC, H.v.begin() + static_cast<size_t>(s2 + 1) * nC);
        if (decision_log_on()) decision_log_override(prm.dec_level, prm.dec_block, prm.minN + s2);
    }
    R.optN = *std::max_element(R.tf.begin(), R.tf.end());
    return R;
}

void ensemble_mean_dev(const double *d_E, long long ldE, int n, int p, int K, double *d_viE) {
    Ctx &c = ctx();
    KernelTimer tm("ensemble_mean");
    hipLaunchKernelGGL(ensemble_mean_kernel, dim3(c.num_cu * 8), dim3(256), 0, c.stream, d_E, ldE, n, p, K, d_viE);
    launch_check("ensemble_mean_kernel");
}

}  // namespace sharp

using namespace sharp;

extern "C" {

int sharp_wMetaC(const int *nC, int N, int C, int hmethod, int enN_cluster, int minN, int maxN, double sil_thre,
                 double height_Ntimes, int *finalC, double *x0, int *ncl, double *w1_out, double *S_out, int *allC_out,
                 int *tf_out) {
    int warn = 0;
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(nC && finalC && ncl, "sharp_wMetaC: null argument");
    WmTask t;
    t.nC = nC; t.N = N; t.C = C;
    t.prm.hmethod = hmethod > 0 ? hmethod : 1;
    t.prm.N_cluster = enN_cluster;
    t.prm.minN = minN > 0 ? minN : 2;
    t.prm.maxN = maxN > 0 ? maxN : 40;
    t.prm.sil_thre = sil_thre;
    t.prm.height_Ntimes = height_Ntimes > 0 ? height_Ntimes : 2.0;
    std::vector<WmTask> ts{t};
    std::vector<WmResult> rs;
    wmetac_batch(ts, x0 != nullptr, w1_out || S_out || tf_out, rs);
    WmResult &R = rs[0];
    warn = R.rc;
    std::copy(R.finalC.begin(), R.finalC.end(), finalC);
    *ncl = R.ncl;
    if (x0) std::copy(R.x0.begin(), R.x0.end(), x0);
    if (w1_out) std::copy(R.w1.begin(), R.w1.end(), w1_out);
    if (S_out) std::copy(R.S.begin(), R.S.end(), S_out);
    if (tf_out) std::copy(R.tf.begin(), R.tf.end(), tf_out);
    if (allC_out) *allC_out = R.allC;
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return warn;
}

int sharp_sMetaC(const int *labels, const double *sE1, long long n, int p, int hmethod, int finalN_cluster, int minN, int maxN,
                 double sil_thre, double height_Ntimes, int *finalColor, int *tf_out, int *nC_out) {
    int warn = 0;
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(labels && sE1 && finalColor, "sharp_sMetaC: null argument");
    SHARP_REQUIRE(n < (1LL << 31), "sharp_sMetaC: too many cells for one call");
    std::vector<int> uid;
    const int nC = first_appearance_ids(labels, n, uid);
    if (nC_out) *nC_out = nC;
    DevBuf<double> dE(static_cast<size_t>(n) * p), dM(static_cast<size_t>(nC) * p);
    dE.upload(sE1, static_cast<size_t>(n) * p);
    cluster_means_dev(dE.p, p, static_cast<int>(n), p, uid, nC, dM.p);
    HcParams prm;
    prm.hmethod = hmethod > 0 ? hmethod : 1;
    prm.N_cluster = finalN_cluster;
    prm.minN = minN > 0 ? minN : 2;
    prm.maxN = maxN > 0 ? maxN : 40;
    prm.sil_thre = sil_thre;
    prm.height_Ntimes = height_Ntimes > 0 ? height_Ntimes : 2.0;
    SmResult R = smetac_from_means(dM.p, nC, p, n, prm);
    warn = R.rc;
    for (long long i = 0; i < n; ++i) finalColor[i] = R.tf[uid[i]];
    if (tf_out) std::copy(R.tf.begin(), R.tf.end(), tf_out);
    }
    catch (const sharp::Error &e) { sharp::set_error(e.what()); return e.code; }
    catch (const std::exception &e) { sharp::set_error(e.what()); return SHARP_ERR; }
    return warn;
}

}  // extern "C"
