// rp2.hip -- the RP matmul (R/RPmat.R:32, R/SHARP.R:343-345,569-585) as a two-kernel
// producer/consumer pipeline over chunks of cells:
//   rp_compact_kernel: streams X once (the only HBM-bound part) and appends one entry per non-zero to its cell's list
//       in a chunk buffer: (gene, count) for integer counts, else the gene and its 44-bit fixed-point fp64 log2(1+x).  Every wave
//       is independent (no workgroup barrier) and keeps two 1024-gene units in flight behind the one it is writing out.
//   rp_apply_kernel: per cell, walks the list in 64-entry batches; each GW-lane group takes the entries held by its own
//       lanes: gene and term of entry u reach the group by DPP row broadcast, each lane fetches its 8 bytes of that gene's packed row
//       list (L2 resident) and adds +-term into the per-cell accumulators in LDS with ds_add_u64: a code is its accumulator's LDS
//       address after one AND (projector.hpp).  Four batches are in flight per wave (atomics, row lists, terms, entry words); the
//       next cell's first batches are set up before the two barriers around the epilogue that scales by sqrt(s)/sqrt(p) and
//       writes the K*p row of E.  The loop holds no LDS operation but the atomics and waits for no load it has just issued.
// The chunks go compact(0) apply(0) compact(1) ... on one stream (two chunk buffers; SHARP_RP_SERIAL=0: compaction on a second
// stream, beside the previous chunk's apply -- no faster, see project_dev_split); integer accumulation keeps E bit-reproducible
// whatever the interleaving.
#include "rp_shared.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace sharp {

constexpr int CP_THREADS = 256;
#ifndef SHARP_AP_THREADS
#define SHARP_AP_THREADS 512
#endif
constexpr int AP_THREADS = SHARP_AP_THREADS;

constexpr int CP_TAB = 256;             // log2(1+x) fixed-point table for integer counts x < CP_TAB
// compacted entry: bits 19..0 gene, bits 27..20 the count (table entries), bit 31: the 64-bit term is stored beside the word
constexpr uint32_t kEntryGeneMask = 0xfffffu, kEntryFull = 0x80000000u;
constexpr int kEntryCountShift = 20;

// fix(x) = round(log2(1+x) * 2^fix_bits) for x = 0..CP_TAB-1, evaluated by the same device libm as the general path
__global__ void rp_fixtab_kernel(double fix_scale, int log10_mode, long long *__restrict__ tab) {
    const int x = threadIdx.x;
    if (x < CP_TAB) tab[x] = __double2ll_rn((log10_mode ? log10(1.0 + static_cast<double>(x)) : log2(1.0 + static_cast<double>(x))) * fix_scale);
}

// One wave per (cell, unit); units of a chunk are dealt round-robin to the waves of a persistent grid.  Both RP kernels were bound by
// the vector ALU's issue rate, not by memory (tools/micro/valu_rate.hip: a compare, a conditional move, a DPP or a three-operand
// instruction holds its SIMD for 4.3 cycles; this kernel ran 53 % and rp_apply 82 % VALU-busy, so they could not overlap either), hence
// the form with the fewest vector instructions per candidate:
//   * candidate q of every lane (16 per unit) is tested by ONE compare whose ballot is also its compaction: the non-zeros of the
//     ballot write (gene, x) into the wave's LDS window at run + mbcnt(ballot) -- no per-lane counts, no prefix scan, no conditional
//     moves; the scalar unit keeps the running offset and the unit's total;
//   * the cell-list slot of unit i + 1 is reserved (one returning atomic per unit) while unit i is written out, so its round trip
//     is never waited for;
//   * one lane per NON-ZERO (11 % of the candidates) turns x into the entry: (gene, count) for integer counts below 256, else the
//     gene and the 44-bit fixed-point fp64 log2(1 + x) beside it -- same bits either way -- and the list goes out coalesced.
// The order of a cell's list is irrelevant: the consumer adds integers.
// T = the storage type of the block: float (counts, fp32-exact data) or double (TPM-like values: the term is log2(1 + x) of the
// double itself, as the reference computes it).
template <typename T> struct CpSlot;
template <> struct __attribute__((aligned(8))) CpSlot<float> { uint32_t gene; float x; };
template <> struct __attribute__((aligned(16))) CpSlot<double> { double x; uint32_t gene; uint32_t pad; };

template <typename T>
__global__ __launch_bounds__(CP_THREADS, sizeof(T) == 4 ? 6 : 3) void rp_compact_kernel(const T *__restrict__ X, int m, long long ld, long long cell0,
                                                                int ncell, int log_flag, double fix_scale, int cap,
                                                                unsigned int *__restrict__ counts, uint32_t *__restrict__ genes,
                                                                long long *__restrict__ fixes) {
    __shared__ CpSlot<T> win[CP_THREADS / 64][CP_UNIT / 2];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the unit bookkeeping runs on the SALU)
    const int units = (m + CP_UNIT - 1) / CP_UNIT;
    const long long total = static_cast<long long>(ncell) * units;
    const long long stride = static_cast<long long>(gridDim.x) * (CP_THREADS / 64);
    const long long it0 = static_cast<long long>(blockIdx.x) * (CP_THREADS / 64) + w;
    if (it0 >= total) return;
    // (cell, unit) of the unit being worked on and of the unit being fetched (two strides ahead), advanced by (stride / units,
    // stride % units) with a carry
    const int sdiv = static_cast<int>(stride / units), smod = static_cast<int>(stride % units);
    int c = static_cast<int>(it0 / units), u = static_cast<int>(it0 % units);
    int fc = c, fu = u;
    auto advance = [&](int &cc, int &uu) {
        cc += sdiv;
        uu += smod;
        if (uu >= units) { uu -= units; ++cc; }
    };
    auto fetch = [&]() -> CpVals<T> {                     // the unit at (fc, fu), clamped to the chunk's last one; then one stride on
        const bool in = fc < ncell;
        const int uu = in ? fu : units - 1;
        const T *unit = X + (cell0 + (in ? fc : ncell - 1)) * ld + static_cast<long long>(uu) * CP_UNIT;
        CpVals<T> r;
        if (uu * CP_UNIT + CP_UNIT <= m) {                 // wave-uniform: only the last unit of a cell is ragged
            r = cp_load_unit<T, false>(unit, 0, lane);
        } else {                                           // its candidates beyond the last gene become zeros: nothing below tests a bound
            r = cp_load_unit<T, true>(unit, static_cast<int>(ld - static_cast<long long>(uu) * CP_UNIT), lane);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (uu * CP_UNIT + CpLayout<T>::gene_of(lane, q) >= m) r.v[q] = T(0);
        }
        advance(fc, fu);
        return r;
    };
    // the non-zeros of a unit: a scalar
    auto count_unit = [&](const CpVals<T> &b) -> int {
        int n = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) n += __popcll(__ballot(b.v[q] != T(0)));
        return n;
    };
    // The reservation of a unit's slot in its cell's list is lane 0's returning atomic, ISSUED here and waited for an iteration later
    // (reserved()): written with the builtin, the compiler waits for the result on the spot -- a full memory round trip per unit, with
    // the unit fetched just before it -- so both halves are inline assembly, and the order of the vector-memory operations between them
    // is part of the contract: reserve(), then the CpLayout<T>::LOADS loads of ONE fetch(), then reserved() -- `s_waitcnt vmcnt(LOADS)`
    // waits for everything older than those loads (vmcnt counts loads, stores and atomics in issue order), i.e. for the atomic, and
    // leaves the fetch in flight.
    auto reserve = [&](int cc, int n) -> unsigned int {
        unsigned int ret = 0u;
        if (lane == 0 && cc < ncell) {
            unsigned int *cp = counts + cc;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(cp), "v"(static_cast<unsigned int>(n)) : "memory");
        }
        return ret;
    };
    auto reserved = [&](unsigned int ret) -> unsigned int {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ret) : "n"(CpLayout<T>::LOADS) : "memory");
        return static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(ret)));
    };
    // Two units per wave: b1 is being written out, b2 is in flight behind it (waited for, by the compiler's own count, where it is
    // copied into b1 at the end of the iteration).
    CpVals<T> b1 = fetch();
    int cnt1 = count_unit(b1);
    unsigned int res1 = reserve(c, cnt1);
    CpVals<T> b2 = fetch();
    int cn = c, un = u;                                    // (cell, unit) of b2
    advance(cn, un);
    for (; c < ncell; advance(c, u)) {
        unsigned int base = reserved(res1);                // (first: nothing may be issued between the fetch and this wait)
        const int ubase = u * CP_UNIT;
        CpSlot<T> *wp = win[w];
        uint32_t *gout = genes + static_cast<long long>(c) * cap;
        long long *fout = fixes + static_cast<long long>(c) * cap;
        // a unit goes out in two halves of eight candidates per lane: a half has at most CP_UNIT / 2 non-zeros, the window's size
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int run = 0;
#pragma unroll
            for (int q = 8 * h; q < 8 * h + 8; ++q) {
                const bool nz = b1.v[q] != T(0);
                const unsigned long long mk = __ballot(nz);
                if (nz) {
                    const int pos = run + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mk >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mk), 0u)));
                    CpSlot<T> sl;
                    sl.gene = static_cast<uint32_t>(ubase + CpLayout<T>::gene_of(lane, q));
                    sl.x = b1.v[q];
                    wp[pos] = sl;
                }
                run += __popcll(mk);
            }
            __builtin_amdgcn_wave_barrier();
            for (int e0 = 0; e0 < run; e0 += 64) {         // one lane per non-zero
                const int e = e0 + lane;
                const bool live = e < run;
                const CpSlot<T> sl = wp[live ? e : 0];
                const T x = sl.x;
                const uint32_t gg = sl.gene;
                // Integer counts below CP_TAB (scRNA counts, the synthetic data) travel as (gene, count) in ONE 32-bit word: the
                // consumer takes their term from the same 256-entry table; everything else (non-integer or large values, raw mode)
                // sets kEntryFull and stores its 64-bit term beside the word -- 4 instead of 12 bytes per non-zero written and read.
                uint32_t entry = gg | kEntryFull;
                long long fx = 0ll;
                bool full = true;
                if (log_flag) {
                    const unsigned xi = (x >= T(0) && x < T(CP_TAB)) ? static_cast<unsigned>(x) : 0u;
                    const bool tab = static_cast<T>(xi) == x && x < T(CP_TAB);
                    if (tab) { entry = gg | (xi << kEntryCountShift); full = false; }
                    if (__ballot(live && !tab) != 0ull) {      // the general path
                        if (!tab) fx = __double2ll_rn((log_flag == 2 ? log10(1.0 + static_cast<double>(x)) : log2(1.0 + static_cast<double>(x))) * fix_scale);
                    }
                } else {
                    fx = __double2ll_rn(static_cast<double>(x) * fix_scale);
                }
                if (live) { gout[base + e] = entry; if (full) fout[base + e] = fx; }
            }
            __builtin_amdgcn_wave_barrier();
            base += static_cast<unsigned int>(run);
        }
        b1 = b2;                                           // (the compiler's wait for the unit in flight)
        cnt1 = count_unit(b1);                             // the next unit's slot is asked for an iteration before it is used
        res1 = reserve(cn, cnt1);
        advance(cn, un);
        b2 = fetch();
    }
}

// A batch = 64 list entries, one per lane; lane group `grp` (GW lanes) works through ITS OWN lanes' entries (u = 0 .. GW-1): gene and
// term of entry u reach the group's lanes by DPP row broadcast (no LDS scratch: the LDS pipe does the atomics and nothing else),
// each lane fetches its 8 bytes of that gene's row list and adds the +-term at its four codes.
// DUAL (ProjectorGroup::neg_base > 0): the group's negative entries have accumulators of their own, neg_base behind the positive ones; the
// codes carry no sign, every lane adds the term as it is, and the epilogue takes the difference of the two arrays.  Measured per
// instruction on gfx950 (tools/micro/valu_rate.hip): a DPP move, a compare, a conditional move or a three-operand instruction holds its
// SIMD for 4.3 cycles, a plain two-operand one for 2.5, and the signed form spends 22 vector instructions per gene batch slot where
// this one spends 11 -- the stage was bound by the vector ALU's issue rate in BOTH its kernels (82 % and 53 % busy), which is why they
// never overlapped.
template <int GW, int SLOTS, bool DUAL>
__global__ __launch_bounds__(AP_THREADS, AP_THREADS >= 512 ? 4 : 4) void rp_apply_kernel(
    int ncell, long long cell0, int cap, const unsigned int *__restrict__ counts, const uint32_t *__restrict__ genes,
    const long long *__restrict__ fixes, const long long *__restrict__ fixtab, const uint16_t *__restrict__ ent, unsigned int dummy_seg,
    const uint2 *__restrict__ ovf_slot, const uint2 *__restrict__ ovf_info, int novf, int ncomp, int neg_base, double inv_fix, double val,
    double out_scale, double *__restrict__ E, long long ldE, int comp0, const int *__restrict__ row_map, unsigned int *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = AP_THREADS / 64, SPAN = SLOTS * GW, U = GW;   // a batch = 64 entries = U per group
    typedef RowWord<SLOTS> Row;                                // a lane's SLOTS codes of one gene: one 4- or 8-byte load
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane % GW;
    const int nacc = DUAL ? 2 * neg_base : ncomp;             // accumulators in front of the dump slots
    for (int c = tid; c < nacc; c += AP_THREADS) acc[c] = 0ull;
    __syncthreads();

    // Per-cell state of this wave: a four-stage pipeline over the cell's batches.  While the atomics of batch b run, the row lists of
    // batch b+NW, the terms of batch b+2NW (table look-up on the entry words fetched one iteration earlier) and the entry words of batch
    // b+3NW are in flight: no load is waited for in the iteration that issues it, and the loop holds no LDS operation but the atomics.
    // A cell's first batches are set up BEFORE the previous cell's barrier and epilogue, so their dependent round trips run under them.
    int nnz = 0, nb = 0;
    const uint32_t *gsrc = genes;
    const long long *fsrc = fixes;
    uint32_t gC = 0u, gL = 0u, wR = 0u;   // genes of the batch being added / of the batch after the next; entry words of the one after that
    long long fC = 0ll, fL = 0ll;         // their terms
    Row cd[U], cdn[U];
    auto load_word = [&](int b) -> uint32_t {                  // lane = entry of batch b; unconditional, clamped
        const int e = (b << 6) + lane;
        return gsrc[e < nnz ? e : 0];
    };
    auto decode = [&](int b, uint32_t w, uint32_t &g, long long &f) {
        const int e = (b << 6) + lane;
        long long ff = fixtab[(w >> kEntryCountShift) & 0xffu];   // 2 KB, cache resident (a read of an LDS copy would queue behind the atomics)
        if (__ballot((w & kEntryFull) != 0u) != 0ull) {        // rare, wave-uniform test: a value outside the table
            if (w & kEntryFull) ff = fsrc[e < nnz ? e : 0];
        }
        g = e < nnz ? (w & kEntryGeneMask) : dummy_seg;
        f = e < nnz ? ff : 0ll;
    };
    const unsigned char *entb = reinterpret_cast<const unsigned char *>(ent);
    auto load_lists = [&](uint32_t g, Row (&dst)[U]) {
        const uint32_t gofs = g * static_cast<uint32_t>(SPAN * 2);    // byte offset of the lane's own entry's segment (32 bits: < 2^21 segments)
        static_for<U>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            dst[u] = load_row_word<SLOTS>(entb + (group_bcast<GW, u>(gofs) + static_cast<uint32_t>(2 * SLOTS * lg)));
        });
    };
    long long row_next = 0;               // E row of the cell being set up (row_map is read here, a cell ahead: read in front of the
                                          // epilogue it was a dependent load in every cell's critical path, 8 % of the kernel)
    auto begin_cell = [&](long long ci) {
        row_next = row_map ? static_cast<long long>(row_map[cell0 + ci]) : cell0 + ci;
        nnz = static_cast<int>(counts[ci]);
        nb = (nnz + 63) >> 6;
        gsrc = genes + ci * cap;
        fsrc = fixes + ci * cap;
        if (wave < nb) {
            const uint32_t w0 = load_word(wave), w1 = load_word(wave + NW);
            wR = load_word(wave + 2 * NW);
            decode(wave, w0, gC, fC);
            decode(wave + NW, w1, gL, fL);
            load_lists(gC, cd);
        }
    };
    // one pipeline step: the atomics of batch b from the row lists in `cur`, the row lists of batch b+NW into `nxt`
    auto step = [&](int b, Row (&cur)[U], Row (&nxt)[U]) {
        const uint32_t gN = gL;
        const long long fN = fL;
        load_lists(gN, nxt);
        decode(b + 2 * NW, wR, gL, fL);
        asm volatile("" : "+v"(gL));      // the old entry words are dead before the new ones are asked for (else the loop ends on a copy
        wR = load_word(b + 3 * NW);       // of the word just requested, i.e. on its whole round trip)
        const uint32_t plo = static_cast<uint32_t>(fC), phi = static_cast<uint32_t>(static_cast<unsigned long long>(fC) >> 32);
        uint32_t more = 0u;
        if constexpr (DUAL) {
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                // (the broadcasts with every lane enabled: a DPP read of a disabled lane returns nothing)
                const uint32_t lo = group_bcast<GW, u>(plo), hi = group_bcast<GW, u>(phi);
                scatter_row_word<0, SLOTS, false>(cur[u], (static_cast<unsigned long long>(hi) << 32) | lo);
                more |= cur[u].x;
            });
        } else {
            const long long fneg = -fC;
            const uint32_t nlo = static_cast<uint32_t>(fneg), nhi = static_cast<uint32_t>(static_cast<unsigned long long>(fneg) >> 32);
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const bool neg = (cur[u].x & kCodeNeg) != 0u;     // a lane's four codes share their sign
                const uint32_t bpl = group_bcast<GW, u>(plo), bph = group_bcast<GW, u>(phi);
                const uint32_t bnl = group_bcast<GW, u>(nlo), bnh = group_bcast<GW, u>(nhi);
                const uint32_t lo = neg ? bnl : bpl, hi = neg ? bnh : bph;
                scatter_row_word<0, SLOTS, true>(cur[u], (static_cast<unsigned long long>(hi) << 32) | lo);
                more |= cur[u].x;
            });
        }
        // rare: a gene may continue in overflow segments (flag in slot 0 of its first lane).  One scalar test per batch.  (Broadcasts
        // again, not LDS shuffles: a returning LDS operation anywhere in the loop makes every iteration wait for its atomics.)
        if (__ballot((more & kCodeMore) != 0u) != 0ull) {
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const unsigned long long full = __ballot((cur[u].x & kCodeMore) != 0u);
                const uint32_t g = group_bcast<GW, u>(gC);
                const uint32_t bpl = group_bcast<GW, u>(plo), bph = group_bcast<GW, u>(phi);
                if ((full >> (lane & ~(GW - 1))) & 1ull) {
                    const long long f = static_cast<long long>((static_cast<unsigned long long>(bph) << 32) | bpl);
                    const uint2 oi = ovf_slot[g];
                    for (uint32_t sg = 0; sg < oi.y; ++sg) {
                        const Row c2 = load_row_word<SLOTS>(ent + (static_cast<size_t>(oi.x) + sg) * SPAN + SLOTS * lg);
                        if constexpr (DUAL) scatter_row_word<0, SLOTS, false>(c2, static_cast<unsigned long long>(f));
                        else scatter_row_word<0, SLOTS, true>(c2, static_cast<unsigned long long>((c2.x & kCodeNeg) ? -f : f));
                    }
                }
            });
        }
        gC = gN;
        fC = fN;
    };
    // Cells are handed out by a counter: workgroup w starts with cell w, every further cell is the next one nobody has taken (cells differ
    // in their number of non-zeros, and ncell / gridDim.x is no integer: 16.3 cells per workgroup statically meant 17 for some).  The
    // index after next is fetched by thread 0 under the epilogue and passed on through LDS at the epilogue's closing barrier.
    unsigned int *qslot = reinterpret_cast<unsigned int *>(acc + nacc + kDumpSlots);
    long long ci = blockIdx.x, cnext = ncell;
    if (tid == 0) *qslot = gridDim.x + atomicAdd(queue, 1u);
    __syncthreads();
    cnext = *qslot;
    __syncthreads();
    if (ci < ncell) begin_cell(ci);
    for (; ci < ncell; ci = cnext, cnext = *qslot) {
        if (wave < nb) {
            for (int b = wave; b < nb; b += 2 * NW) {          // two steps per trip: the two row-list sets swap roles, nothing is copied
                step(b, cd, cdn);
                if (b + NW < nb) step(b + NW, cdn, cd);
            }
        }
        const long long row = row_next;
        if (cnext < ncell) begin_cell(cnext);                     // (its row lists travel under the barrier and the epilogue)
        __syncthreads();   // every wave's atomics for this cell have landed (and everybody has read the queue slot)
        if (tid == 0) *qslot = gridDim.x + atomicAdd(queue, 1u);
        double *erow = E + row * ldE + comp0;
        for (int c = tid; c < ncomp; c += AP_THREADS) {
            long long a = static_cast<long long>(atomicExch(&acc[c], 0ull));   // read and clear in one LDS operation (ds_wrxchg_rtn_b64)
            if constexpr (DUAL) a -= static_cast<long long>(atomicExch(&acc[neg_base + c], 0ull));
            // streaming store: E is next read by another kernel, and kept out of the L2 it does not push row lists (and the compaction's
            // entries) out -- apply 344 -> 331 us, the compaction beside it 175 -> 159 us per launch
            __builtin_nontemporal_store(out_scale * (val * (static_cast<double>(a) * inv_fix)), &erow[c]);
        }
        __syncthreads();
    }
}

namespace {
constexpr int kRing = 16;                // entry buffers: two in rotation, or one per chunk of a block compacted ahead (rp_compact_ahead)
// What rp_compact_ahead has compacted: consumed by the project_dev call that presents its token, by no other.
struct Precompact {
    unsigned token = 0;                  // 0: nothing
    const void *X = nullptr;
    bool f64 = false;
    int m = 0, n = 0, log_flag = 0, fix_bits = 0, nbuf = 0, done = 0;
    long long ld = 0, chunk = 0;
};
struct SplitWs {
    DevBuf<unsigned int> counts;         // entries per cell, all chunks of a call (zeroed once per call)
    DevBuf<uint32_t> genes[kRing];
    DevBuf<long long> fixes[kRing];
    DevBuf<long long> fixtab;
    double fixtab_scale = 0.0;
    hipEvent_t ev_compact[kRing] = {}, ev_apply[kRing] = {}, ev_start = nullptr;
    Precompact pre;
    unsigned next_token = 1;
};
SplitWs &sws() { return per_slot<SplitWs>(); }

// cells per chunk of the RP stage: two (genes, fix) buffers of <= 2 GB each, sized for the worst case (every gene non-zero);
// few, equal chunks: each launch pays a tail, and a chunk must give every workgroup several cells
long long rp_chunk_cells(int m, int n) {
    const int cap = (m + 3) / 4 * 4;
    long long chunk = std::max<long long>(512, (2048LL << 20) / (static_cast<long long>(cap) * 12));
    chunk = std::min<long long>(chunk, 16384);
    if (knobs().rp_chunk > 0) chunk = std::max(64, knobs().rp_chunk);
    chunk = std::min<long long>(chunk, n);
    const long long nch = (n + chunk - 1) / chunk;
    return (n + nch - 1) / nch;
}
void ensure_fixtab(SplitWs &W, double fix_scale, int log_flag, hipStream_t st) {
    const double tab_key = log_flag == 2 ? -fix_scale : fix_scale;       // the table depends on the scale and on the log base
    if (W.fixtab.n == 0 || W.fixtab_scale != tab_key) {
        W.fixtab.ensure(CP_TAB);
        hipLaunchKernelGGL(rp_fixtab_kernel, dim3(1), dim3(CP_TAB), 0, st, fix_scale, log_flag == 2 ? 1 : 0, W.fixtab.p);
        launch_check("rp_fixtab_kernel");
        W.fixtab_scale = tab_key;
    }
}
void ensure_ring(SplitWs &W, int nbuf, long long chunk, int cap) {
    for (int q = 0; q < nbuf; ++q) {
        W.genes[q].ensure(chunk * cap); W.fixes[q].ensure(chunk * cap);
        if (!W.ev_compact[q]) { SHARP_HIP_CHECK(hipEventCreateWithFlags(&W.ev_compact[q], hipEventDisableTiming)); SHARP_HIP_CHECK(hipEventCreateWithFlags(&W.ev_apply[q], hipEventDisableTiming)); }
    }
    if (!W.ev_start) SHARP_HIP_CHECK(hipEventCreateWithFlags(&W.ev_start, hipEventDisableTiming));
}
// one chunk's compaction into ring buffer q, on stream st
void launch_compact(SplitWs &W, XRef dX, int m, long long ld, long long c0, int nc, int log_flag, double fix_scale, int cap, int q, hipStream_t st) {
    Ctx &c = ctx();
    const int units = (m + CP_UNIT - 1) / CP_UNIT;
    const long long waves = static_cast<long long>(nc) * units;
    // a persistent grid of exactly the workgroups that are resident together (the units are dealt to the waves statically)
    static int cp_occ[2] = {0, 0};
    int &occ = cp_occ[dX.f64 ? 1 : 0];
    if (occ == 0) {
        const void *kf = dX.f64 ? reinterpret_cast<const void *>(rp_compact_kernel<double>) : reinterpret_cast<const void *>(rp_compact_kernel<float>);
        SHARP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, CP_THREADS, 0));
        occ = std::max(1, occ);
    }
    const int cp_per_cu = std::min(occ, knobs().rp_cp_wgs);
    const int blocks = static_cast<int>(std::min<long long>((waves + 3) / 4, static_cast<long long>(c.num_cu) * cp_per_cu));
    KernelTimer tc("rp_compact", st);
    if (dX.f64)
        hipLaunchKernelGGL(rp_compact_kernel<double>, dim3(blocks), dim3(CP_THREADS), 0, st, dX.d64(), m, ld, c0, nc, log_flag, fix_scale,
                           cap, W.counts.p + c0, W.genes[q].p, W.fixes[q].p);
    else
        hipLaunchKernelGGL(rp_compact_kernel<float>, dim3(blocks), dim3(CP_THREADS), 0, st, dX.f32(), m, ld, c0, nc, log_flag, fix_scale,
                           cap, W.counts.p + c0, W.genes[q].p, W.fixes[q].p);
    launch_check("rp_compact_kernel");
}
}  // namespace

bool rp_split_eligible(XRef X, int m, long long ld) {
    const bool vec = (ld % (X.f64 ? 2 : 4) == 0) && ((reinterpret_cast<uintptr_t>(X.p) & 15u) == 0);
    return vec && m >= 8 && m <= (1 << 20) && (X.f64 || knobs().rp_kernel != 1) && knobs().rp_kernel != 2 && !(knobs().rp_kernel == 0 && m <= 16);
}

// The compaction needs X only, not the projectors: a caller that is about to BUILD its projectors (1.9 ms of latency-bound kernels at
// cfg2: Mersenne-Twister draws, packing) starts the compaction of the block first, on the second stream, and it runs beside that build
// instead of behind it.  Every chunk gets a buffer of its own when the block's entries fit `budget` (worst case: every gene non-zero,
// 12 B each), else the first two chunks go ahead and the rest keep the rotation.  Returns the token project_dev takes (0: nothing done).
unsigned rp_compact_ahead(XRef dX, int m, int n, long long ld, int log_flag) {
    Ctx &c = ctx();
    SplitWs &W = sws();
    W.pre = Precompact();
    if (!knobs().rp_ahead || c.polite || !log_flag || n < 4096 || !rp_split_eligible(dX, m, ld) || rp_pc_eligible(dX, m, ld)) return 0;
    const int cap = (m + 3) / 4 * 4;
    const long long chunk = rp_chunk_cells(m, n);
    const int nchunks = static_cast<int>((n + chunk - 1) / chunk);
    size_t free_b = 0, total_b = 0;
    SHARP_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const double need = static_cast<double>(nchunks) * chunk * cap * 12.0;
    const bool have = nchunks <= kRing && W.genes[nchunks - 1].n >= static_cast<size_t>(chunk) * cap;      // (already allocated by an earlier call)
    const int nbuf = (nchunks <= kRing && (have || (need <= 16.0e9 && need <= 0.25 * static_cast<double>(free_b)))) ? nchunks : 2;
    const int done = std::min(nbuf, nchunks);
    const int fix_bits = std::min(RP_FIX_BITS, dX.log_fix_bits);
    const double fix_scale = std::ldexp(1.0, fix_bits);
    ensure_ring(W, nbuf, chunk, cap);
    ensure_fixtab(W, fix_scale, log_flag, c.stream);
    W.counts.ensure(static_cast<size_t>(n) + nchunks);
    hipStream_t s2 = c.stream2;
    SHARP_HIP_CHECK(hipEventRecord(W.ev_start, c.stream));                  // X is complete, the buffers' last readers are enqueued
    SHARP_HIP_CHECK(hipStreamWaitEvent(s2, W.ev_start, 0));
    {
        KernelTimer t("rp_stage_ahead", s2);                                // (bench.py adds it to rp_stage: the stage's work, wherever it ran)
        SHARP_HIP_CHECK(hipMemsetAsync(W.counts.p, 0, (static_cast<size_t>(n) + nchunks) * 4, s2));
        for (int ch = 0; ch < done; ++ch) {
            const long long c0 = ch * chunk;
            const int nc = static_cast<int>(std::min<long long>(chunk, n - c0));
            launch_compact(W, dX, m, ld, c0, nc, log_flag, fix_scale, cap, ch, s2);
            SHARP_HIP_CHECK(hipEventRecord(W.ev_compact[ch], s2));
        }
    }
    Precompact &P = W.pre;
    P.token = W.next_token++; if (W.next_token == 0) W.next_token = 1;
    P.X = dX.p; P.f64 = dX.f64; P.m = m; P.n = n; P.ld = ld; P.log_flag = log_flag; P.fix_bits = fix_bits; P.nbuf = nbuf; P.done = done; P.chunk = chunk;
    return P.token;
}
void rp_compact_ahead_drop() { sws().pre = Precompact(); }
void rp_trim() {
    SplitWs &W = sws();
    W.pre = Precompact();
    for (int q = 2; q < kRing; ++q) { W.genes[q].release(); W.fixes[q].release(); }
}

template <int GW, int SLOTS, bool DUAL>
static void launch_apply(const ProjectorGroup &g, const Projector &pr, int ncell, long long cell0, int cap, const unsigned int *counts,
                         const uint32_t *genes, const long long *fixes, double inv_fix, double *dE, long long ldE, const int *row_map,
                         unsigned int *queue, hipStream_t st) {
    Ctx &c = ctx();
    const size_t lds = static_cast<size_t>(g.acc_slots() + kDumpSlots) * 8 + 8;     // accumulators, dump accumulators, the cell queue's slot
    auto kern = rp_apply_kernel<GW, SLOTS, DUAL>;
    SHARP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    {   // scatter_row_word<0, ...>: the accumulators sit at LDS address 0, i.e. the kernel must not have static LDS in front of the dynamic block
        hipFuncAttributes fa;
        SHARP_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
        SHARP_REQUIRE(fa.sharedSizeBytes == 0, "rp_apply_kernel: static LDS in front of the accumulators");
    }
    int per_cu = 1;
    SHARP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), AP_THREADS, lds));
    per_cu = std::max(1, std::min(per_cu, 4));
    per_cu = std::min(per_cu, knobs().rp_ap_wgs);
    const long long blocks = std::min<long long>(ncell, static_cast<long long>(c.num_cu) * per_cu);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(AP_THREADS), lds, st, ncell, cell0, cap, counts, genes, fixes,
                       static_cast<const long long *>(sws().fixtab.p),
                       g.ent.p, static_cast<unsigned int>(g.nseg), g.ovf_slot.p, g.ovf_info.p, g.novf, g.ncomp, g.neg_base, inv_fix, pr.val,
                       1.0 / std::sqrt(static_cast<double>(pr.p)), dE, ldE, g.k0 * pr.p, row_map, queue);
    launch_check("rp_apply_kernel");
}

// X must be 16-byte aligned with ld % 4 == 0.  One projector group (K*p <= 12288) per call.
void project_dev_split(const Projector &pr, const ProjectorGroup &g, XRef dX, int m, int n, long long ld, int log_flag,
                       int fix_bits, double *dE, long long ldE, const int *d_row_map, unsigned ahead_token) {
    Ctx &c = ctx();
    SplitWs &W = sws();
    // One stream by default: the two kernels do not overlap when they share the chip (each is limited by the memory requests a CU keeps
    // in flight), so chunk c + 1 compacted beside chunk c's apply buys nothing and the cross-stream events between the
    // launches cost 2 % (K = 5: 2.24 against 2.29 ms per stage; K = 15 the same either way).  SHARP_RP_SERIAL=0: two streams.
    // Two streams (chunk c + 1 compacted beside chunk c's apply) where the apply kernel has dual accumulators: with its vector work halved
    // the two kernels overlap a little (K = 5: 1.88 against 1.97 ms per stage); with signed codes (K = 15) they do not, and the
    // cross-stream events cost 2 %.  Never for a block prepared under another block's tail (Ctx::polite: stream2 is a high-priority stream).
    const int ts = knobs().rp_two_streams;
    const bool two_streams = !c.polite && (ts < 0 ? g.neg_base > 0 : ts > 0);
    hipStream_t s2 = two_streams ? c.stream2 : c.stream;
    const int cap = (m + 3) / 4 * 4;                         // worst case: every gene non-zero
    const long long chunk = rp_chunk_cells(m, n);
    const int nchunks = static_cast<int>((n + chunk - 1) / chunk);
    // chunks compacted ahead of the projector build (rp_compact_ahead) by THIS call's front: theirs are the first `done` ring buffers
    const Precompact &P = W.pre;
    const bool pre = ahead_token != 0 && P.token == ahead_token && P.X == dX.p && P.f64 == dX.f64 && P.m == m && P.n == n && P.ld == ld &&
                     P.log_flag == log_flag && P.fix_bits == fix_bits && P.chunk == chunk;
    const int nbuf = pre ? P.nbuf : 2, done = pre ? P.done : 0;
    ensure_ring(W, nbuf, chunk, cap);
    const double fix_scale = std::ldexp(1.0, fix_bits), inv_fix = std::ldexp(1.0, -fix_bits);
    ensure_fixtab(W, fix_scale, log_flag, c.stream);
    KernelTimer t("rp_stage");                                 // the whole stage, measured on the main stream
    const bool two = s2 != c.stream;                           // (one stream: its order is all the synchronisation there is to do)
    if (two) {
        SHARP_HIP_CHECK(hipEventRecord(W.ev_start, c.stream));
        SHARP_HIP_CHECK(hipStreamWaitEvent(s2, W.ev_start, 0));
    }
    W.counts.ensure(static_cast<size_t>(n) + nchunks);          // entries per cell, then one cell-queue counter per chunk (apply kernel)
    if (!pre) SHARP_HIP_CHECK(hipMemsetAsync(W.counts.p, 0, (static_cast<size_t>(n) + nchunks) * 4, s2));
    else SHARP_HIP_CHECK(hipMemsetAsync(W.counts.p + n, 0, static_cast<size_t>(nchunks) * 4, c.stream));   // (the queue counters again: a second projector group re-reads the lists)
    for (int ch = 0; ch < nchunks; ++ch) {
        const int q = ch % nbuf;
        const long long c0 = ch * chunk;
        const int nc = static_cast<int>(std::min<long long>(chunk, n - c0));
        if (ch < done) {
            SHARP_HIP_CHECK(hipStreamWaitEvent(c.stream, W.ev_compact[q], 0));                  // compacted ahead, on the second stream
        } else {
            if (two && ch >= nbuf) SHARP_HIP_CHECK(hipStreamWaitEvent(s2, W.ev_apply[q], 0));    // buffer q free again
            launch_compact(W, dX, m, ld, c0, nc, log_flag, fix_scale, cap, q, s2);
            if (two) {
                SHARP_HIP_CHECK(hipEventRecord(W.ev_compact[q], s2));
                SHARP_HIP_CHECK(hipStreamWaitEvent(c.stream, W.ev_compact[q], 0));
            }
        }
        {
            KernelTimer ta("rp_apply");
#define SHARP_AP(GWV, SL, DU) launch_apply<GWV, SL, DU>(g, pr, nc, c0, cap, W.counts.p + c0, W.genes[q].p, W.fixes[q].p, inv_fix, dE, ldE, d_row_map, W.counts.p + n + ch, c.stream)
            const bool dual = g.neg_base > 0;
            if (g.gw == 16 && g.slots == 4) { if (dual) SHARP_AP(16, 4, true); else SHARP_AP(16, 4, false); }
            else if (g.gw == 16) { if (dual) SHARP_AP(16, 2, true); else SHARP_AP(16, 2, false); }
            else if (g.gw == 8) { if (dual) SHARP_AP(8, 4, true); else SHARP_AP(8, 4, false); }
            else { if (dual) SHARP_AP(4, 4, true); else SHARP_AP(4, 4, false); }
#undef SHARP_AP
        }
        if (two) SHARP_HIP_CHECK(hipEventRecord(W.ev_apply[q], c.stream));
    }
    // lists that have been overwritten (a rotation shorter than the block) serve no second projector group
    if (pre && done < nchunks) W.pre = Precompact();
}

}  // namespace sharp
