// projector.hip -- host construction of the ranM() projectors (R/ranM.R:11-33,
// R/ranM2.R:44-68, R/RPmat.R:82-99) and upload as packed gene-major row lists.
#include "projector.hpp"

#include <cmath>
#include <map>
#include <mutex>
#include <random>
#include <thread>

#include "rrng.hpp"

namespace sharp {

namespace {

// One projector: m*p draws of sample(c(+v,0,-v), replace=TRUE, prob=c(q,P,q)).
// ProbSampleReplace sorts the probabilities descending (revsort: P, then the q of
// element 3, then the q of element 1) and walks the cumulative sums with one
// unif_rand() per element: u <= P -> 0 ; u <= P+q -> -v ; else +v.
void draw_projector(int m, int p, double seed, std::vector<uint32_t> &rowptr, std::vector<int32_t> &ent) {
    const double s = std::sqrt(static_cast<double>(m));
    double pr[3] = {1.0 / (2.0 * s), 1.0 - 1.0 / s, 1.0 / (2.0 * s)};
    double tot = 0.0;
    for (double v : pr) if (v > 0.0) tot += v;         // FixupProb
    for (double &v : pr) v /= tot;
    const double cut0 = pr[1];
    const double cut1 = pr[1] + pr[2];
    uint32_t useed;
    if (std::fmod(seed, 1.0) == 0.0) {
        useed = static_cast<uint32_t>(static_cast<int32_t>(seed));
    } else {  // the reference's 0.5 sentinel = "do not call set.seed()"
        std::random_device rd;
        useed = rd();
    }
    RRng rng(useed);
    rowptr.assign(static_cast<size_t>(m) + 1, 0);
    ent.clear();
    ent.reserve(static_cast<size_t>(static_cast<double>(m) * p / s * 1.1) + 64);
    for (int g = 0; g < m; ++g) {
        for (int c = 0; c < p; ++c) {
            const double u = rng.unif();
            if (u <= cut0) continue;
            ent.push_back(u <= cut1 ? ~c : c);
        }
        rowptr[g + 1] = static_cast<uint32_t>(ent.size());
    }
}

std::mutex g_mu;
std::map<int, std::shared_ptr<Projector>> g_table;
int g_next = 1;

}  // namespace

std::shared_ptr<Projector> build_projector(int m, int p, int K, const double *seeds) {
    SHARP_REQUIRE(m >= 2 && p >= 1 && K >= 1, "projector: need m >= 2, p >= 1, K >= 1");
    SHARP_REQUIRE(p <= kMaxCompPerGroup, "projector: reduced dimension p too large for one launch group");
    auto pr = std::make_shared<Projector>();
    pr->m = m; pr->p = p; pr->K = K;
    pr->val = std::sqrt(std::sqrt(static_cast<double>(m)));
    pr->h_rowptr.resize(K);
    pr->h_ent.resize(K);
    {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 4;
        const int nthr = static_cast<int>(std::min<unsigned>(hw, K));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back([&, t] {
                for (int k = t; k < K; k += nthr) draw_projector(m, p, seeds[k], pr->h_rowptr[k], pr->h_ent[k]);
            });
        for (auto &th : pool) th.join();
    }
    // pack into groups whose component count fits one scatter launch
    const int per_group = std::max(1, kMaxCompPerGroup / p);
    for (int k0 = 0; k0 < K; k0 += per_group) {
        ProjectorGroup grp;
        grp.k0 = k0;
        grp.kcount = std::min(per_group, K - k0);
        grp.ncomp = grp.kcount * p;
        // gather the group's entries gene-major
        std::vector<uint32_t> rowptr(static_cast<size_t>(m) + 1, 0);
        std::vector<uint16_t> flat;
        size_t total = 0;
        for (int k = k0; k < k0 + grp.kcount; ++k) total += pr->h_ent[k].size();
        flat.reserve(total);
        int max_len = 0;
        for (int g = 0; g < m; ++g) {
            for (int kk = 0; kk < grp.kcount; ++kk) {
                const auto &rp = pr->h_rowptr[k0 + kk];
                const auto &en = pr->h_ent[k0 + kk];
                for (uint32_t q = rp[g]; q < rp[g + 1]; ++q) {
                    const int32_t e = en[q];
                    const int c = e >= 0 ? e : ~e;
                    flat.push_back(static_cast<uint16_t>(kk * p + c) | (e < 0 ? 0x8000u : 0u));
                }
            }
            rowptr[g + 1] = static_cast<uint32_t>(flat.size());
            max_len = std::max<int>(max_len, static_cast<int>(rowptr[g + 1] - rowptr[g]));
        }
        grp.nnz = static_cast<long long>(flat.size());
        grp.mean_len = static_cast<double>(grp.nnz) / m;
        grp.max_len = max_len;
        // lanes per gene: the smallest width whose 4*gw-entry segment covers mean + ~3 sd
        const double cover = grp.mean_len + 3.0 * std::sqrt(grp.mean_len) + 1.0;
        grp.gw = cover <= 16 ? 4 : (cover <= 32 ? 8 : 16);
        const int span = 4 * grp.gw;
        std::vector<uint32_t> ovf_gene;
        std::vector<uint2> ovf_info;
        long long nseg = m;
        for (int g = 0; g < m; ++g) {
            const uint32_t len = rowptr[g + 1] - rowptr[g];
            if (len > static_cast<uint32_t>(span)) {
                const uint32_t extra = (len - span + span - 1) / span;
                ovf_gene.push_back(static_cast<uint32_t>(g));
                ovf_info.push_back(make_uint2(static_cast<uint32_t>(nseg), extra));
                nseg += extra;
            }
        }
        grp.nseg = nseg;
        grp.novf = static_cast<int>(ovf_gene.size());
        std::vector<uint16_t> ent(static_cast<size_t>(nseg + 1) * span, 0xFFFFu);
        size_t ov = 0;
        for (int g = 0; g < m; ++g) {
            const uint32_t len = rowptr[g + 1] - rowptr[g];
            const uint16_t *src = flat.data() + rowptr[g];
            size_t extra_base = 0;
            if (len > static_cast<uint32_t>(span)) extra_base = ovf_info[ov++].x;
            for (uint32_t i = 0; i < len; ++i) {
                const uint32_t sgm = i / span, r = i % span;
                const uint32_t q = r / grp.gw, lane = r % grp.gw;
                const size_t seg = sgm == 0 ? static_cast<size_t>(g) : extra_base + (sgm - 1);
                ent[seg * span + 4 * lane + q] = src[i];
            }
        }
        grp.ent.alloc(ent.size());
        grp.ent.upload(ent.data(), ent.size());
        ovf_gene.push_back(0xFFFFFFFFu);               // keep the tables non-empty
        ovf_info.push_back(make_uint2(0u, 0u));
        grp.ovf_gene.alloc(ovf_gene.size());
        grp.ovf_info.alloc(ovf_info.size());
        grp.ovf_gene.upload(ovf_gene.data(), ovf_gene.size());
        grp.ovf_info.upload(ovf_info.data(), ovf_info.size());
        stream_sync();
        pr->groups.push_back(std::move(grp));
    }
    return pr;
}

int register_projector(std::shared_ptr<Projector> pr) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int h = g_next++;
    g_table[h] = std::move(pr);
    return h;
}
std::shared_ptr<Projector> get_projector(int handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_table.find(handle);
    SHARP_REQUIRE(it != g_table.end(), "unknown projector handle");
    return it->second;
}
void drop_projector(int handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_table.erase(handle);
}

}  // namespace sharp

using namespace sharp;

extern "C" {

int sharp_projector_create(int m, int p, int K, const double *seeds, int *handle) {
    SHARP_API_BEGIN
    ctx();
    SHARP_REQUIRE(seeds && handle, "sharp_projector_create: null argument");
    *handle = register_projector(build_projector(m, p, K, seeds));
    SHARP_API_END
}
int sharp_projector_destroy(int handle) {
    SHARP_API_BEGIN
    stream_sync();
    drop_projector(handle);
    SHARP_API_END
}
int sharp_projector_info(int handle, int *m, int *p, int *K, long long *nnz_total) {
    SHARP_API_BEGIN
    auto pr = get_projector(handle);
    if (m) *m = pr->m;
    if (p) *p = pr->p;
    if (K) *K = pr->K;
    if (nnz_total) *nnz_total = pr->nnz_total();
    SHARP_API_END
}
int sharp_projector_triplets(int handle, int k, int *gene, int *col, signed char *sign, long long *nnz) {
    SHARP_API_BEGIN
    auto pr = get_projector(handle);
    SHARP_REQUIRE(k >= 0 && k < pr->K, "sharp_projector_triplets: k out of range");
    const auto &rp = pr->h_rowptr[k];
    const auto &en = pr->h_ent[k];
    if (nnz) *nnz = static_cast<long long>(en.size());
    if (gene) {
        SHARP_REQUIRE(col && sign, "sharp_projector_triplets: null output");
        for (int g = 0; g < pr->m; ++g)
            for (uint32_t q = rp[g]; q < rp[g + 1]; ++q) {
                const int32_t e = en[q];
                gene[q] = g;
                col[q] = e >= 0 ? e : ~e;
                sign[q] = e >= 0 ? 1 : -1;
            }
    }
    SHARP_API_END
}

}  // extern "C"
