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

// Bulk Mersenne-Twister: the same stream as RRng::unif() (R's set.seed() scrambling + MT19937 + tempering),
// produced 624 outputs at a time with loops the host compiler can vectorise.
struct MtBulk {
    static constexpr int N = 624, M = 397;
    uint32_t st[N];
    explicit MtBulk(uint32_t seed) {
        for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
        seed = 69069u * seed + 1u;
        for (int j = 0; j < N; ++j) { seed = 69069u * seed + 1u; st[j] = seed; }
    }
    static inline uint32_t tw(uint32_t hi, uint32_t lo) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    void next(uint32_t *out) {   // regenerate the state, write the 624 tempered outputs
        for (int k = 0; k < N - M; ++k) st[k] = st[k + M] ^ tw(st[k], st[k + 1]);                  // reads old values only
        for (int k = N - M; k < 2 * (N - M); ++k) st[k] = st[k + M - N] ^ tw(st[k], st[k + 1]);    // new [0,227) + old
        for (int k = 2 * (N - M); k < N - 1; ++k) st[k] = st[k + M - N] ^ tw(st[k], st[k + 1]);
        st[N - 1] = st[M - 1] ^ tw(st[N - 1], st[0]);
        for (int k = 0; k < N; ++k) {
            uint32_t y = st[k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[k] = y;
        }
    }
};

// largest 32-bit y with unif(y) <= cut, unif(y) = fixup((double)y * 2^-32-ish) exactly as unif_rand() computes it
uint32_t unif_threshold(double cut) {
    const double c = 2.3283064365386963e-10;
    auto u = [&](uint64_t y) {
        const double v = static_cast<double>(static_cast<uint32_t>(y)) * c;
        constexpr double kHalfUlp = 0.5 * 2.328306437080797e-10;
        if (v <= 0.0) return kHalfUlp;
        if (1.0 - v <= 0.0) return 1.0 - kHalfUlp;
        return v;
    };
    int64_t y = static_cast<int64_t>(cut / c);
    if (y > 0xffffffffLL) y = 0xffffffffLL;
    if (y < 0) y = 0;
    while (y < 0xffffffffLL && u(static_cast<uint64_t>(y + 1)) <= cut) ++y;
    while (y > 0 && !(u(static_cast<uint64_t>(y)) <= cut)) --y;
    return static_cast<uint32_t>(y);
}

// One projector: m*p draws of sample(c(+v,0,-v), replace=TRUE, prob=c(q,P,q)).
// ProbSampleReplace sorts the probabilities descending (revsort: P, then the q of
// element 3, then the q of element 1) and walks the cumulative sums with one
// unif_rand() per element: u <= P -> 0 ; u <= P+q -> -v ; else +v.
// The comparisons are done on the raw 32-bit outputs against exact integer thresholds.
void draw_projector(int m, int p, double seed, std::vector<uint32_t> &rowptr, std::vector<int32_t> &ent) {
    const double s = std::sqrt(static_cast<double>(m));
    double pr[3] = {1.0 / (2.0 * s), 1.0 - 1.0 / s, 1.0 / (2.0 * s)};
    double tot = 0.0;
    for (double v : pr) if (v > 0.0) tot += v;         // FixupProb
    for (double &v : pr) v /= tot;
    const uint32_t t0 = unif_threshold(pr[1]);
    const uint32_t t1 = unif_threshold(pr[1] + pr[2]);
    uint32_t useed;
    if (std::fmod(seed, 1.0) == 0.0) {
        useed = static_cast<uint32_t>(static_cast<int32_t>(seed));
    } else {  // the reference's 0.5 sentinel = "do not call set.seed()"
        std::random_device rd;
        useed = rd();
    }
    MtBulk mt(useed);
    rowptr.assign(static_cast<size_t>(m) + 1, 0);
    ent.clear();
    ent.reserve(static_cast<size_t>(static_cast<double>(m) * p / s * 1.1) + 64);
    uint32_t buf[MtBulk::N + 8];
    const unsigned long long total = static_cast<unsigned long long>(m) * p;
    std::vector<uint32_t> hit_gene;
    hit_gene.reserve(ent.capacity());
    const uint32_t up = static_cast<uint32_t>(p);
    for (unsigned long long base = 0; base < total; base += MtBulk::N) {
        mt.next(buf);
        const int cnt = static_cast<int>(std::min<unsigned long long>(MtBulk::N, total - base));
        for (int q = cnt; q < cnt + 8 && q < MtBulk::N + 8; ++q) buf[q] = 0;     // neutral tail for the 8-wide test
        for (int i = 0; i < cnt; i += 8) {
            // 99.3 % of the draws are zeros of the projector: test eight at a time, locate the few hits by division
            uint32_t mx = buf[i];
            for (int j = 1; j < 8; ++j) mx = buf[i + j] > mx ? buf[i + j] : mx;
            if (mx <= t0) continue;
            for (int j = 0; j < 8 && i + j < cnt; ++j) {
                const uint32_t y = buf[i + j];
                if (y > t0) {
                    const unsigned long long idx = base + static_cast<unsigned long long>(i + j);   // element i = r*p + c (byrow fill)
                    const uint32_t g = static_cast<uint32_t>(idx / up), c = static_cast<uint32_t>(idx - static_cast<unsigned long long>(g) * up);
                    hit_gene.push_back(g);
                    ent.push_back(y <= t1 ? ~static_cast<int32_t>(c) : static_cast<int32_t>(c));
                }
            }
        }
    }
    for (uint32_t g : hit_gene) ++rowptr[g + 1];
    for (int g = 0; g < m; ++g) rowptr[g + 1] += rowptr[g];
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
