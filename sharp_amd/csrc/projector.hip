// projector.hip -- host construction of the ranM() projectors (R/ranM.R:11-33,
// R/ranM2.R:11-35, R/RPmat.R:14-31) and upload as packed gene-major row lists.
#include "projector.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
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
void draw_thresholds(int m, uint32_t &t0, uint32_t &t1) {
    const double s = std::sqrt(static_cast<double>(m));
    double pr[3] = {1.0 / (2.0 * s), 1.0 - 1.0 / s, 1.0 / (2.0 * s)};
    double tot = 0.0;
    for (double v : pr) if (v > 0.0) tot += v;         // FixupProb
    for (double &v : pr) v /= tot;
    t0 = unif_threshold(pr[1]);
    t1 = unif_threshold(pr[1] + pr[2]);
}
uint32_t seed_word(double seed) {
    if (std::fmod(seed, 1.0) == 0.0) return static_cast<uint32_t>(static_cast<int32_t>(seed));
    std::random_device rd;                              // the reference's 0.5 sentinel = "do not call set.seed()"
    return rd();
}
void draw_projector(int m, int p, double seed, std::vector<uint32_t> &rowptr, std::vector<int32_t> &ent) {
    const double s = std::sqrt(static_cast<double>(m));
    uint32_t t0, t1;
    draw_thresholds(m, t0, t1);
    MtBulk mt(seed_word(seed));
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


// ---------------------------------------------------------------------------------------------
// Device construction.  One workgroup per projector regenerates R's Mersenne-Twister state 624 words at a time
// (three dependent phases of <= 227 independent words, double-buffered in LDS: 3 barriers per regeneration),
// tempers each word in registers and tests it against the integer thresholds; the 0.7 % hits are staged in LDS and
// appended to the projector's hit list (element index i = r*p + c of the byrow fill, bit 31 = negative).
// ---------------------------------------------------------------------------------------------
#include "mt_jump_table.inc"

constexpr int PD_THREADS = 640;          // one lane per Mersenne-Twister state word in the jump (624); the draw phases use 227
constexpr int PD_XS = 19937 + 624 + 227;   // raw words needed to jump: x_0 .. x_(19936+623), rounded up to whole 227-word steps
constexpr int PD_STAGE = 4096;

__device__ __forceinline__ uint32_t pd_tw(uint32_t hi, uint32_t lo) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
__device__ __forceinline__ uint32_t pd_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ __launch_bounds__(PD_THREADS) void proj_draw_kernel(const uint32_t *__restrict__ seeds, unsigned long long total, uint32_t t0,
                                                               uint32_t t1, int flush_every, uint32_t *__restrict__ hits,
                                                               unsigned int cap, unsigned int *__restrict__ nhits,
                                                               int *__restrict__ err) {
    // One workgroup per (projector, segment of 2^MT_JUMP_LOG2 draws).  Segment t > 0 first jumps the seed state t segments ahead:
    // with g_t(x) = x^(t 2^MT_JUMP_LOG2) mod phi (mt_jump_table.inc), the state window at that offset is w_j = XOR_{i : g_t has x^i}
    // x_(i+j), j = 0..623, over the raw word sequence x -- 20 560 words regenerated into LDS, then 624 independent XOR
    // chains.  All segments of a projector therefore start at once instead of one after the other.
    constexpr int N = 624, M = 397, D = N - M;   // D = 227
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    uint32_t *xs = reinterpret_cast<uint32_t *>(psm);                 // [PD_XS]   (jump only)
    uint32_t(*st)[N] = reinterpret_cast<uint32_t(*)[N]>(xs + PD_XS);  // [2][N]
    uint32_t *stage = reinterpret_cast<uint32_t *>(st + 2);           // [PD_STAGE]
    unsigned int *scount_p = reinterpret_cast<unsigned int *>(stage + PD_STAGE);
#define scount (*scount_p)
    const int k = blockIdx.x, seg = blockIdx.y, nseg = gridDim.y, tid = threadIdx.x;
    const unsigned long long d0 = static_cast<unsigned long long>(seg) << MT_JUMP_LOG2;
    const unsigned long long dlim = d0 + (1ull << MT_JUMP_LOG2);
    const unsigned long long d1 = dlim < total ? dlim : total;
    uint32_t *out = hits + (static_cast<size_t>(k) * nseg + seg) * cap;
    // set.seed(): 50 warm-up steps of x <- 69069 x + 1, one for the position word, then one per state word
    if (tid == 0) {
        uint32_t seed = seeds[k];
        for (int j = 0; j < 51; ++j) seed = 69069u * seed + 1u;
        for (int j = 0; j < N; ++j) { seed = 69069u * seed + 1u; st[0][j] = seed; }
        scount = 0u;
    }
    __syncthreads();
    if (seg > 0) {
        for (int j = tid; j < N; j += PD_THREADS) xs[j] = st[0][j];
        __syncthreads();
        for (int n0 = 0; n0 + N < PD_XS; n0 += D) {                   // x_(n+624) = x_(n+397) ^ tw(x_n, x_(n+1)), 227 at a time
            if (tid < D && n0 + N + tid < PD_XS) xs[n0 + N + tid] = xs[n0 + M + tid] ^ pd_tw(xs[n0 + tid], xs[n0 + tid + 1]);
            __syncthreads();
        }
        // One lane per state word, every coefficient visited: 32 independent LDS reads per polynomial word, masked by the
        // coefficient, so the reads pipeline (a loop over the set bits only, with its data-dependent trip count, waited for
        // every read in turn and took 1.2 ms per jump; this form takes the LDS time of 19 937 x 624 word reads).
        const unsigned int *g = mt_jump_poly[seg - 1];
        const int j = tid < N ? tid : 0;
        uint32_t acc0 = 0u, acc1 = 0u;
        for (int w = 0; w < N; ++w) {
            const uint32_t word = g[w];                                // same word in every lane
            const uint32_t *x = xs + 32 * w + j;
#pragma unroll
            for (int b = 0; b < 32; b += 2) {
                acc0 ^= x[b] & (0u - ((word >> b) & 1u));
                acc1 ^= x[b + 1] & (0u - ((word >> (b + 1)) & 1u));
            }
        }
        __syncthreads();
        if (tid < N) st[0][tid] = acc0 ^ acc1;
        __syncthreads();
    }
    unsigned int gcount = 0;   // hits already flushed (uniform)
    int cur = 0, since = 0;
    bool overflow = false;
    auto emit = [&](uint32_t word, unsigned long long idx) {
        if (idx < d1) {
            const uint32_t y = pd_temper(word);
            if (y > t0) {
                const unsigned int pos = atomicAdd(&scount, 1u);
                if (pos < PD_STAGE) stage[pos] = static_cast<uint32_t>(idx) | (y <= t1 ? 0x80000000u : 0u);
            }
        }
    };
    for (unsigned long long base = d0; base < d1; base += N) {
        const uint32_t *c = st[cur];
        uint32_t *nx = st[cur ^ 1];
        if (tid < D) {                                   // words [0, 227): old values only
            const uint32_t v = c[tid + M] ^ pd_tw(c[tid], c[tid + 1]);
            nx[tid] = v;
            emit(v, base + tid);
        }
        __syncthreads();
        if (tid < D) {                                   // words [227, 454): new [0, 227) + old
            const int q = D + tid;
            const uint32_t v = nx[q - D] ^ pd_tw(c[q], c[q + 1]);
            nx[q] = v;
            emit(v, base + q);
        }
        __syncthreads();
        if (tid < N - 2 * D) {                           // words [454, 624): new [227, 397) + old; the last one wraps to new word 0
            const int q = 2 * D + tid;
            const uint32_t v = nx[q - D] ^ pd_tw(c[q], q == N - 1 ? nx[0] : c[q + 1]);
            nx[q] = v;
            emit(v, base + q);
        }
        __syncthreads();
        cur ^= 1;
        if (++since == flush_every || base + N >= d1) {
            const unsigned int cnt = scount;
            if (cnt > PD_STAGE || gcount + cnt > cap) overflow = true;
            const unsigned int cc = cnt > PD_STAGE ? PD_STAGE : cnt;
            for (unsigned int i = tid; i < cc && gcount + i < cap; i += PD_THREADS) out[gcount + i] = stage[i];
            gcount += cc;
            since = 0;
            __syncthreads();
            if (tid == 0) scount = 0u;
            __syncthreads();
        }
    }
    if (tid == 0) {
        nhits[k * nseg + seg] = gcount < cap ? gcount : cap;
        if (overflow) *err = 1;
    }
#undef scount
}

// entries per gene over the projectors [k0, k0+kcount) of one group
__global__ void proj_count_kernel(const uint32_t *__restrict__ hits, unsigned int cap, const unsigned int *__restrict__ nhits, int k0,
                                  uint32_t p, unsigned int *__restrict__ len) {
    const int k = k0 + blockIdx.y;
    const size_t list = static_cast<size_t>(k) * gridDim.z + blockIdx.z;      // (projector, draw segment)
    const unsigned int n = nhits[list];
    const uint32_t *h = hits + list * cap;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t idx = h[i] & 0x7fffffffu, g = idx / p, neg = h[i] >> 31;
        atomicAdd(&len[g], neg ? 0x10000u : 1u);                                // low half: positive codes, high half: negative
    }
}
// codes (kk*p + c, bit 15 = negative) into the gene's CSR slot range, in arrival order
__global__ void proj_fill_kernel(const uint32_t *__restrict__ hits, unsigned int cap, const unsigned int *__restrict__ nhits, int k0,
                                 uint32_t p, const uint32_t *__restrict__ rowptr, unsigned int *__restrict__ fill,
                                 uint16_t *__restrict__ flat) {
    const int kk = blockIdx.y, k = k0 + kk;
    const size_t list = static_cast<size_t>(k) * gridDim.z + blockIdx.z;
    const unsigned int n = nhits[list];
    const uint32_t *h = hits + list * cap;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t w = h[i], idx = w & 0x7fffffffu;
        const uint32_t g = idx / p, c = idx - g * p;
        const unsigned int slot = atomicAdd(&fill[g], 1u);
        flat[rowptr[g] + slot] = static_cast<uint16_t>(kk * p + c) | ((w >> 31) ? 0x8000u : 0u);
    }
}
// Writes a gene's codes (src: n codes as comp | 0x8000 for negative, any order) into its segments.  Lanes hold four codes of ONE
// sign: the positive codes take the first ceil(np/4) lanes of the gene's lane sequence, the negative codes the next ceil(nn/4); lane
// l lives in segment l / gw (the gene's own segment g, then its overflow segments from extra_base), slots 4 (l % gw) .. + 3, and its
// slot q goes out with the lane group's q-th atomic instruction.  A 64-bit LDS atomic is served per 16 contiguous lanes over 16
// eight-byte bank pairs (MI355X_MICROARCH.md, LDS: the ds_write_b64 row): one cycle plus one per extra code of the same class
// (component mod 16) in the instruction.  So the codes of a class are dealt to different slot columns, the most populous classes
// first, within the room of their sign's lanes.
// The lanes the gene does not use keep kCodePad in every slot; the unused slots of a lane that holds codes get the lane's pad_code
// (projector.hpp); every slot of a negative lane carries the sign bit, the consumer reads it from slot 0.
constexpr uint32_t kPlacePerm = 64;   // codes of a gene whose class order fits the caller's scratch (longer lists take the slow loop)
// seg0: the segment index of the gene's own segment in `ent` (g for the global table; 0 when `ent` is a private image of the gene's segments,
// whose overflow segments then follow from extra_base = 1)
__host__ __device__ inline void place_gene(const uint16_t *src, uint32_t n, uint32_t gw, uint32_t S /* slots per lane: 2 or 4 */, int ncomp, int neg_base,
                                           size_t g, size_t extra_base, uint16_t *ent, unsigned char *perm /* kPlacePerm bytes of scratch */, size_t seg0) {
    const uint32_t span = S * gw;
    const bool dual = neg_base > 0;                      // negative entries go to a second accumulator array: no signs, dense lanes
    const int dump_base = dual ? 2 * neg_base : ncomp;   // the dump accumulators sit behind every real one
    uint32_t np = 0;
    for (uint32_t i = 0; i < n; ++i) np += (!dual && (src[i] & 0x8000u)) ? 0u : 1u;
    const uint32_t cap[2] = {(np + S - 1u) / S, (n - np + S - 1u) / S};   // lanes per sign
    const uint32_t lane0[2] = {0u, cap[0]};
    unsigned long long cnt[2] = {0ull, 0ull};            // codes per (sign, column), 4 x 16 bit
    unsigned long long colmask = 0ull;                   // classes present per column, 4 x 16 bit
    auto slot_ptr = [&](uint32_t lane, uint32_t q) -> uint16_t * {
        const uint32_t sgm = lane / gw;
        const size_t seg = sgm == 0 ? seg0 : extra_base + (sgm - 1);
        return ent + seg * span + S * (lane % gw) + q;
    };
    // codes per class (16 x 16 bit)
    unsigned long long per[4] = {0ull, 0ull, 0ull, 0ull};
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t c = src[i] & 15u;
        per[c >> 2] += 1ull << ((c & 3u) * 16u);
    }
    auto class_count = [&](uint32_t c) { return static_cast<uint32_t>(per[c >> 2] >> ((c & 3u) * 16u)) & 0xffffu; };
    // Most constrained first: the classes in order of decreasing population (a class of four codes needs all four columns; the singles
    // fit anywhere), each code to the column that has the most room left for its sign among those that do not hold its class yet.
    // Only a class of more than four codes, or a sign whose columns are full, forces two codes of a class into one column.
    auto place = [&](uint32_t i) {
        const uint32_t c = src[i] & 15u;                // (neg_base % 16 == 0: both accumulators of a component are of one class)
        const uint32_t s = (!dual && (src[i] & 0x8000u)) ? 1u : 0u;
        const unsigned long long cs = s ? cnt[1] : cnt[0];
        uint32_t q = 4u, room = 0u;
        for (uint32_t qq = 0; qq < S; ++qq) {             // room for the sign and the class not in the column yet: the emptiest
            const uint32_t used = static_cast<uint32_t>(cs >> (16u * qq)) & 0xffffu;
            if (used < cap[s] && !((colmask >> (16u * qq + c)) & 1ull) && (q == 4u || cap[s] - used > room)) { q = qq; room = cap[s] - used; }
        }
        if (q == 4u)
            for (uint32_t qq = 0; qq < S; ++qq) {         // a conflict cannot be avoided: the emptiest column with room (S cap[s] >= the sign's codes)
                const uint32_t used = static_cast<uint32_t>(cs >> (16u * qq)) & 0xffffu;
                if (used < cap[s] && (q == 4u || cap[s] - used > room)) { q = qq; room = cap[s] - used; }
            }
        colmask |= 1ull << (16u * q + c);
        const uint32_t comp = (src[i] & 0x7fffu) + ((dual && (src[i] & 0x8000u)) ? static_cast<uint32_t>(neg_base) : 0u);
        *slot_ptr(lane0[s] + (static_cast<uint32_t>(cs >> (16u * q)) & 0xffffu), q) = static_cast<uint16_t>(comp << 3);
        if (s) cnt[1] += 1ull << (16u * q); else cnt[0] += 1ull << (16u * q);
    };
    // the classes in order of decreasing population; within a class the codes keep their order
    unsigned char order[16];
    {
        uint32_t done_classes = 0u;
        for (uint32_t round = 0; round < 16u; ++round) {
            uint32_t c = 16u, best = 0u;
            for (uint32_t cc = 0; cc < 16u; ++cc)
                if (!((done_classes >> cc) & 1u) && (c == 16u || class_count(cc) > best)) { c = cc; best = class_count(cc); }
            done_classes |= 1u << c;
            order[round] = static_cast<unsigned char>(c);
        }
    }
    if (n <= kPlacePerm) {                               // one counting sort by class rank, then one pass
        unsigned long long off[4] = {0ull, 0ull, 0ull, 0ull};   // first position of every class, 16 x 16 bit
        uint32_t run = 0u;
        for (uint32_t r = 0; r < 16u; ++r) {
            const uint32_t c = order[r];
            off[c >> 2] |= static_cast<unsigned long long>(run) << ((c & 3u) * 16u);
            run += class_count(c);
        }
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t c = src[i] & 15u, sh = (c & 3u) * 16u;
            perm[static_cast<uint32_t>(off[c >> 2] >> sh) & 0xffffu] = static_cast<unsigned char>(i);
            off[c >> 2] += 1ull << sh;
        }
        for (uint32_t j = 0; j < n; ++j) place(perm[j]);
    } else {
        for (uint32_t r = 0; r < 16u; ++r) {
            if (class_count(order[r]) == 0u) break;
            for (uint32_t i = 0; i < n; ++i) if ((src[i] & 15u) == order[r]) place(i);
        }
    }
    // the unused slots of the lanes that hold codes: a dump accumulator whose class the slot's column does not hold yet (and no other
    // pad of the column has taken), so that the padding costs its instruction no LDS cycle either
    for (uint32_t l = 0; l < cap[0] + cap[1]; ++l)       // (every lane of a sign's range holds at least one code)
        for (uint32_t q = 0; q < S; ++q) {
            uint16_t *sp = slot_ptr(l, q);
            if (*sp == static_cast<uint16_t>(kCodePad)) {
                uint32_t d = (l + static_cast<uint32_t>(g)) % gw;                       // (start rotated by the gene: genes that share an instruction
                for (uint32_t t = 0; t < 16u; ++t) {                                    //  then rarely pick the same dump accumulator)
                    const uint32_t dd = (d + t) % 16u, cls = (static_cast<uint32_t>(dump_base) + dd) & 15u;
                    if (!((colmask >> (16u * q + cls)) & 1ull)) { d = dd; colmask |= 1ull << (16u * q + cls); break; }
                }
                *sp = pad_code(dump_base, d);
            }
            if (l >= lane0[1]) *sp |= static_cast<uint16_t>(kCodeNeg);
        }
    if (cap[0] + cap[1] > gw) ent[seg0 * span] |= static_cast<uint16_t>(kCodeMore);   // (slot 0 of lane 0: the one slot every consumer masks anyway)
}

// One thread per gene: order the gene's codes by component (the order of the host build: projector, then column) and write them into the
// fixed-stride lane-major segments (+ overflow segments).  With 20 000 threads on 256 CUs nothing hides a memory access, and sorting and
// placing in global memory cost ~1500 dependent round trips per thread (0.53 ms at cfg2, as long as half the Mersenne-Twister draw, on the
// step's critical path): the gene's codes are fetched ONCE into LDS, ranked there (independent compares instead of an insertion sort's
// chain), placed into an LDS image of the gene's segments, and the image leaves in 16-byte stores.  Genes beyond the LDS room (more than
// kLayCodes codes or kLaySegs segments: none at the shapes in use) keep the global path.
constexpr int kLayCodes = 128, kLaySegs = 3, kLaySpan = 64;
__global__ __launch_bounds__(64) void proj_layout_kernel(int m, const uint32_t *__restrict__ rowptr, uint16_t *__restrict__ flat, int gw, int slots, int ncomp, int neg_base,
                                   const uint2 *__restrict__ ovf_slot, const uint2 *__restrict__ ovf_info, int novf,
                                   uint16_t *__restrict__ ent) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= m) return;
    const uint32_t b = rowptr[g], len = rowptr[g + 1] - b;
    uint16_t *src = flat + b;
    size_t extra_base = 0;
    uint32_t extra = 0;
    if (novf > 0) { const uint2 o = ovf_slot[g]; extra_base = o.x; extra = o.y; }
    __shared__ unsigned char sperm[64][kPlacePerm];       // (blockDim.x = 64)
    __shared__ __attribute__((aligned(16))) uint16_t raw[64][kLayCodes], sorted[64][kLayCodes];
    __shared__ __attribute__((aligned(16))) uint16_t image[64][kLaySegs * kLaySpan];
    const uint32_t span = static_cast<uint32_t>(slots) * static_cast<uint32_t>(gw);
    if (len <= static_cast<uint32_t>(kLayCodes) && 1u + extra <= static_cast<uint32_t>(kLaySegs) && span <= static_cast<uint32_t>(kLaySpan)) {
        uint16_t *rw = raw[threadIdx.x], *so = sorted[threadIdx.x], *im = image[threadIdx.x];
        for (uint32_t i = 0; i < len; i += 8) {            // eight independent loads at a time
            uint16_t t[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) t[u] = src[i + u < len ? i + u : len - 1];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) if (i + u < len) rw[i + u] = t[u];
        }
        for (uint32_t i = 0; i < len; ++i) {               // rank = position in the order by component (arrival order among equals, like the insertion sort)
            const uint32_t key = rw[i] & 0x7fffu;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < len; ++j) { const uint32_t kj = rw[j] & 0x7fffu; rank += (kj < key || (kj == key && j < i)) ? 1u : 0u; }
            so[rank] = rw[i];
        }
        const uint32_t nimg = (1u + extra) * span;
        for (uint32_t i = 0; i < nimg; ++i) im[i] = static_cast<uint16_t>(kCodePad);
        place_gene(so, len, static_cast<uint32_t>(gw), static_cast<uint32_t>(slots), ncomp, neg_base, static_cast<size_t>(g), 1, im, sperm[threadIdx.x], 0);
        // the image out: the gene's own segment, then its overflow segments (span u16 = 32 ... 128 bytes each, 16-byte aligned on both sides)
        for (uint32_t sgm = 0; sgm <= extra; ++sgm) {
            uint16_t *dst = ent + (sgm == 0 ? static_cast<size_t>(g) : extra_base + (sgm - 1)) * span;
            const uint16_t *from = im + sgm * span;
            for (uint32_t i = 0; i < span; i += 8) *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(from + i);
        }
        return;
    }
    for (uint32_t i = 1; i < len; ++i) {                 // insertion sort: lists are a few dozen entries, nearly sorted
        const uint16_t v = src[i];
        uint32_t j = i;
        while (j > 0 && (src[j - 1] & 0x7fffu) > (v & 0x7fffu)) { src[j] = src[j - 1]; --j; }
        src[j] = v;
    }
    place_gene(src, len, static_cast<uint32_t>(gw), static_cast<uint32_t>(slots), ncomp, neg_base, static_cast<size_t>(g), extra_base, ent, sperm[threadIdx.x], static_cast<size_t>(g));
}
__global__ void proj_fill_u16_kernel(uint16_t *p, size_t n, uint16_t v) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) p[i] = v;
}

// projector handles: a table per device slot (a projector's row lists live on its slot's device); handle numbers are unique across slots
std::mutex g_mu;
struct ProjTable { std::map<int, std::shared_ptr<Projector>> t; };
std::map<int, std::shared_ptr<Projector>> &g_table_ref() { return per_slot<ProjTable>().t; }
#define g_table g_table_ref()
int g_next = 1;

}  // namespace

// The shape of a group's row-list segments: gw lanes per gene, `slots` codes per lane.  `cover` = mean + 3 sd codes per gene.  A 64-bit
// LDS atomic is served per 16 contiguous lanes, one cycle when their bank classes differ: with ONE gene per 16-lane group the packer
// controls every class an instruction touches (place_gene), with two genes in a group (8 lanes each) their classes collide at random --
// the apply kernel at K = 5, 8 lanes x 4 slots, spent 3.3 of its 7.3 LDS cycles per atomic instruction on such conflicts, and the LDS
// is what bounds that kernel (row lists served from the CU's L1 instead of L2 leave its time unchanged).  Hence 16 lanes x 2 slots
// from 17 codes per gene on; only short lists (K = 1, 2) keep 4-lane groups: there the vector instructions per gene weigh more.
static void choose_shape(ProjectorGroup &grp, double cover) {
    if (cover <= 16) { grp.gw = 4; grp.slots = 4; }
    else if (cover <= 32) { grp.gw = 16; grp.slots = 2; }
    else { grp.gw = 16; grp.slots = 4; }
    if (knobs().rp_shape == 1 && cover > 16 && cover <= 32) { grp.gw = 8; grp.slots = 4; }   // SHARP_RP_SHAPE=1: the 8 x 4 form (A/B runs)
}

int dual_neg_base(int ncomp) {
    if (!knobs().rp_dual) return 0;
    const int base = (ncomp + 15) / 16 * 16;             // (a multiple of 16: a component's two accumulators are of one LDS bank class)
    // both arrays inside the 13-bit component field of a code, and in at most 64 KB of LDS (two workgroups per CU)
    return (2 * base <= kMaxCompPerGroup && 2 * base * 8 <= 65536) ? base : 0;
}

static std::shared_ptr<Projector> build_projector_host(int m, int p, int K, const double *seeds) {
    SHARP_REQUIRE(m >= 2 && p >= 1 && K >= 1, "projector: need m >= 2, p >= 1, K >= 1");
    SHARP_REQUIRE(p <= kMaxCompPerGroup, "projector: reduced dimension p too large for one launch group");
    auto pr = std::make_shared<Projector>();
    pr->m = m; pr->p = p; pr->K = K;
    pr->val = std::sqrt(std::sqrt(static_cast<double>(m)));
    pr->h_rowptr.resize(K);
    pr->h_ent.resize(K);
    {
        const unsigned hw = static_cast<unsigned>(host_cores());
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
        grp.neg_base = dual_neg_base(grp.ncomp);
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
        choose_shape(grp, cover);
        const int span = grp.slots * grp.gw;
        std::vector<uint32_t> ovf_gene;
        std::vector<uint2> ovf_info;
        long long nseg = m;
        std::vector<uint16_t> lanes(static_cast<size_t>(m));   // lanes of four same-sign codes the gene occupies
        for (int g = 0; g < m; ++g) {
            int np = 0, nn = 0;
            for (uint32_t q = rowptr[g]; q < rowptr[g + 1]; ++q) { if (flat[q] & 0x8000u) ++nn; else ++np; }
            lanes[g] = static_cast<uint16_t>(grp.neg_base ? code_lanes(np + nn, 0, grp.slots) : code_lanes(np, nn, grp.slots));
            if (lanes[g] > grp.gw) {
                const uint32_t extra = (lanes[g] - grp.gw + grp.gw - 1) / grp.gw;
                ovf_gene.push_back(static_cast<uint32_t>(g));
                ovf_info.push_back(make_uint2(static_cast<uint32_t>(nseg), extra));
                nseg += extra;
            }
        }
        grp.nseg = nseg;
        grp.novf = static_cast<int>(ovf_gene.size());
        std::vector<uint16_t> ent(static_cast<size_t>(nseg + 1) * span, static_cast<uint16_t>(kCodePad));
        size_t ov = 0;
        for (int g = 0; g < m; ++g) {
            const uint32_t len = rowptr[g + 1] - rowptr[g];
            const uint16_t *src = flat.data() + rowptr[g];
            size_t extra_base = 0;
            if (lanes[g] > grp.gw) extra_base = ovf_info[ov++].x;
            unsigned char perm[kPlacePerm];
            place_gene(src, len, static_cast<uint32_t>(grp.gw), static_cast<uint32_t>(grp.slots), grp.ncomp, grp.neg_base, static_cast<size_t>(g), extra_base, ent.data(), perm, static_cast<size_t>(g));
        }
        grp.ent.alloc(ent.size());
        grp.ent.upload(ent.data(), ent.size());
        ovf_gene.push_back(0xFFFFFFFFu);               // keep the tables non-empty
        ovf_info.push_back(make_uint2(0u, 0u));
        grp.ovf_gene.alloc(ovf_gene.size());
        grp.ovf_info.alloc(ovf_info.size());
        grp.ovf_gene.upload(ovf_gene.data(), ovf_gene.size());
        {
            std::vector<uint2> slot(static_cast<size_t>(m), make_uint2(0u, 0u));
            for (int q = 0; q < grp.novf; ++q) slot[ovf_gene[q]] = ovf_info[q];
            grp.ovf_slot.alloc(slot.size());
            grp.ovf_slot.upload(slot.data(), slot.size());
            stream_sync();                               // (slot is a local)
        }
        grp.ovf_info.upload(ovf_info.data(), ovf_info.size());
        stream_sync();
        pr->groups.push_back(std::move(grp));
    }
    return pr;
}


// Device build: draws on the GPU (proj_draw_kernel), per-gene counts, then -- after one 80 KB download to size the
// segments on the host -- fill + layout kernels.  Same row lists as the host build (tests compare the triplets and the projections).
static std::shared_ptr<Projector> build_projector_device(int m, int p, int K, const double *seeds, const std::function<void()> *after_draw) {
    SHARP_REQUIRE(m >= 2 && p >= 1 && K >= 1, "projector: need m >= 2, p >= 1, K >= 1");
    SHARP_REQUIRE(p <= kMaxCompPerGroup, "projector: reduced dimension p too large for one launch group");
    Ctx &c = ctx();
    auto pr = std::make_shared<Projector>();
    pr->m = m; pr->p = p; pr->K = K;
    pr->val = std::sqrt(std::sqrt(static_cast<double>(m)));
    pr->h_rowptr.resize(K);
    pr->h_ent.resize(K);
    uint32_t t0, t1;
    draw_thresholds(m, t0, t1);
    std::vector<uint32_t> useed(K);
    for (int k = 0; k < K; ++k) useed[k] = seed_word(seeds[k]);
    const unsigned long long total = static_cast<unsigned long long>(m) * p;
    const int S = static_cast<int>((total + (1ull << MT_JUMP_LOG2) - 1) >> MT_JUMP_LOG2);    // draw segments per projector
    const double expect = static_cast<double>(std::min<unsigned long long>(total, 1ull << MT_JUMP_LOG2)) / std::sqrt(static_cast<double>(m));
    pr->hit_cap = static_cast<unsigned int>(expect * 1.25 + 8.0 * std::sqrt(expect) + 4096.0);   // per (projector, segment) list
    pr->draw_segments = S;
    pr->d_hits.alloc_pooled(static_cast<size_t>(K) * S * pr->hit_cap);
    pr->d_nhits.alloc_pooled(static_cast<size_t>(K) * S);
    // regenerations between flushes of the LDS stage: about a quarter of its capacity in expected hits
    const int flush_every = std::max(1, static_cast<int>(PD_STAGE / 4 / (624.0 / std::sqrt(static_cast<double>(m)))));
    DevBuf<uint32_t> d_seed;
    DevBuf<int> d_err;
    d_seed.alloc_pooled(K); d_err.alloc_pooled(1);     // (pooled: released every call, and every user synchronises the stream first)
    d_seed.upload(useed.data(), K);
    d_err.zero();
    {
        KernelTimer t("projector_draw");
        const size_t lds = (static_cast<size_t>(PD_XS) + 2 * 624 + PD_STAGE + 4) * 4;
        SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(proj_draw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(lds)));
        hipLaunchKernelGGL(proj_draw_kernel, dim3(K, S), dim3(PD_THREADS), lds, c.stream, d_seed.p, total, t0, t1, flush_every,
                           pr->d_hits.p, pr->hit_cap, pr->d_nhits.p, d_err.p);
        launch_check("proj_draw_kernel");
    }
    if (after_draw) (*after_draw)();
    pr->h_nhits.resize(static_cast<size_t>(K) * S);
    const int per_group = std::max(1, kMaxCompPerGroup / p);
    bool first = true;
    for (int k0 = 0; k0 < K; k0 += per_group) {
        ProjectorGroup grp;
        grp.k0 = k0;
        grp.kcount = std::min(per_group, K - k0);
        grp.ncomp = grp.kcount * p;
        DevBuf<unsigned int> d_len, d_fill;
        d_len.alloc_pooled(static_cast<size_t>(m) + 1); d_fill.alloc_pooled(static_cast<size_t>(m) + 1);
        d_len.zero(); d_fill.zero();
        hipLaunchKernelGGL(proj_count_kernel, dim3(16, grp.kcount, S), dim3(256), 0, c.stream, pr->d_hits.p, pr->hit_cap, pr->d_nhits.p, k0,
                           static_cast<uint32_t>(p), d_len.p);
        launch_check("proj_count_kernel");
        std::vector<unsigned int> len(static_cast<size_t>(m) + 1);
        int err = 0;
        if (first) {                                     // (all three copies behind ONE synchronisation: each one costs a wake-up of the host thread)
            SHARP_HIP_CHECK(hipMemcpyAsync(&err, d_err.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            SHARP_HIP_CHECK(hipMemcpyAsync(pr->h_nhits.data(), pr->d_nhits.p, pr->h_nhits.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, c.stream));
        }
        d_len.download(len.data(), len.size());          // synchronises the stream
        if (first) {
            SHARP_REQUIRE(err == 0, "projector: hit list overflow in the device build");
            first = false;
        }
        grp.neg_base = dual_neg_base(grp.ncomp);
        const int neg_base = grp.neg_base;
        std::vector<uint32_t> rowptr(static_cast<size_t>(m) + 1, 0);
        int max_len = 0;
        std::vector<uint16_t> lanes(static_cast<size_t>(m));   // lanes of four same-sign codes the gene occupies
        for (int g = 0; g < m; ++g) {
            const int np = static_cast<int>(len[g] & 0xffffu), nn = static_cast<int>(len[g] >> 16);
            lanes[g] = static_cast<uint16_t>(np);           // (positive codes for now: the lanes are sized below, once the group's shape is known)
            len[g] = static_cast<unsigned int>(np + nn);
            rowptr[g + 1] = rowptr[g] + len[g];
            max_len = std::max<int>(max_len, static_cast<int>(len[g]));
        }
        grp.nnz = rowptr[m];
        grp.mean_len = static_cast<double>(grp.nnz) / m;
        grp.max_len = max_len;
        const double cover = grp.mean_len + 3.0 * std::sqrt(grp.mean_len) + 1.0;
        choose_shape(grp, cover);
        const int span = grp.slots * grp.gw;
        std::vector<uint32_t> ovf_gene;
        std::vector<uint2> ovf_info;
        long long nseg = m;
        for (int g = 0; g < m; ++g) {
            const int np = lanes[g], nn = static_cast<int>(len[g]) - np;
            lanes[g] = static_cast<uint16_t>(neg_base ? code_lanes(np + nn, 0, grp.slots) : code_lanes(np, nn, grp.slots));
            if (lanes[g] > grp.gw) {
                const uint32_t extra = (lanes[g] - grp.gw + grp.gw - 1) / grp.gw;
                ovf_gene.push_back(static_cast<uint32_t>(g));
                ovf_info.push_back(make_uint2(static_cast<uint32_t>(nseg), extra));
                nseg += extra;
            }
        }
        grp.nseg = nseg;
        grp.novf = static_cast<int>(ovf_gene.size());
        ovf_gene.push_back(0xFFFFFFFFu);               // keep the tables non-empty
        ovf_info.push_back(make_uint2(0u, 0u));
        grp.ovf_gene.alloc_pooled(ovf_gene.size());
        grp.ovf_info.alloc_pooled(ovf_info.size());
        grp.ovf_gene.upload(ovf_gene.data(), ovf_gene.size());
        std::vector<uint2> slot(static_cast<size_t>(m), make_uint2(0u, 0u));   // (alive until the synchronisation at the end of the group)
        for (int q = 0; q < grp.novf; ++q) slot[ovf_gene[q]] = ovf_info[q];
        grp.ovf_slot.alloc_pooled(slot.size());
        grp.ovf_slot.upload(slot.data(), slot.size());
        grp.ovf_info.upload(ovf_info.data(), ovf_info.size());
        DevBuf<uint32_t> d_rowptr;
        d_rowptr.alloc_pooled(rowptr.size());
        d_rowptr.upload(rowptr.data(), rowptr.size());
        DevBuf<uint16_t> d_flat;
        d_flat.alloc_pooled(static_cast<size_t>(std::max<long long>(grp.nnz, 1)));
        const size_t nent = static_cast<size_t>(nseg + 1) * span;
        grp.ent.alloc_pooled(nent);
        hipLaunchKernelGGL(proj_fill_u16_kernel, dim3(256), dim3(256), 0, c.stream, grp.ent.p, nent, static_cast<uint16_t>(kCodePad));
        hipLaunchKernelGGL(proj_fill_kernel, dim3(16, grp.kcount, S), dim3(256), 0, c.stream, pr->d_hits.p, pr->hit_cap, pr->d_nhits.p, k0,
                           static_cast<uint32_t>(p), d_rowptr.p, d_fill.p, d_flat.p);
        hipLaunchKernelGGL(proj_layout_kernel, dim3((m + 63) / 64), dim3(64), 0, c.stream,      // (one thread per gene, a wave per workgroup: every CU gets some)
                           m, d_rowptr.p, d_flat.p, grp.gw, grp.slots, grp.ncomp, grp.neg_base,
                           grp.ovf_slot.p, grp.ovf_info.p, grp.novf, grp.ent.p);
        launch_check("proj_layout_kernel");
        stream_sync();                                   // the temporaries above are released on scope exit
        pr->groups.push_back(std::move(grp));
    }
    pr->device_built = true;
    return pr;
}

// Host lists (one CSR per projector, column order) for sharp_projector_triplets(): made on demand from the hit list.
void ensure_host_lists(Projector &pr, int k) {
    if (!pr.device_built || !pr.h_rowptr[k].empty()) return;
    const int S = pr.draw_segments;
    unsigned int n = 0;
    for (int t = 0; t < S; ++t) n += pr.h_nhits[static_cast<size_t>(k) * S + t];
    std::vector<uint32_t> h(n);
    unsigned int off = 0;
    for (int t = 0; t < S; ++t) {
        const unsigned int nt = pr.h_nhits[static_cast<size_t>(k) * S + t];
        if (nt) SHARP_HIP_CHECK(hipMemcpyAsync(h.data() + off, pr.d_hits.p + (static_cast<size_t>(k) * S + t) * pr.hit_cap,
                                               static_cast<size_t>(nt) * 4, hipMemcpyDeviceToHost, ctx().stream));
        off += nt;
    }
    stream_sync();
    std::sort(h.begin(), h.end(), [](uint32_t a, uint32_t b) { return (a & 0x7fffffffu) < (b & 0x7fffffffu); });
    auto &rp = pr.h_rowptr[k];
    auto &en = pr.h_ent[k];
    rp.assign(static_cast<size_t>(pr.m) + 1, 0);
    en.resize(n);
    const uint32_t up = static_cast<uint32_t>(pr.p);
    for (unsigned int i = 0; i < n; ++i) {
        const uint32_t idx = h[i] & 0x7fffffffu, g = idx / up, c = idx - g * up;
        ++rp[g + 1];
        en[i] = (h[i] >> 31) ? ~static_cast<int32_t>(c) : static_cast<int32_t>(c);
    }
    for (int g = 0; g < pr.m; ++g) rp[g + 1] += rp[g];
}

std::shared_ptr<Projector> build_projector(int m, int p, int K, const double *seeds, const std::function<void()> *after_draw) {
    const unsigned long long draws = static_cast<unsigned long long>(m) * static_cast<unsigned long long>(p);
    if (knobs().proj_host || draws >= (1ull << 31) || draws > (static_cast<unsigned long long>(MT_JUMP_COUNT + 1) << MT_JUMP_LOG2))
    {
        if (after_draw) (*after_draw)();
        return build_projector_host(m, p, K, seeds);
    }
    return build_projector_device(m, p, K, seeds, after_draw);
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
    std::shared_ptr<Projector> gone;                     // ~Projector drains its device: outside the registry lock, so that the other
    {                                                    // workers' lookups do not wait behind it
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_table.find(handle);
        if (it == g_table.end()) return;
        gone = std::move(it->second);
        g_table.erase(it);
    }
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
    drop_pending_front();                                // (a front prepared with this projector would outlive its handle)
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
    ensure_host_lists(*pr, k);
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
