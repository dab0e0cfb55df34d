// rp3.hip -- the RP matmul (R/RPmat.R:32, R/SHARP.R:343-345,569-585) as ONE persistent kernel with
// specialised waves.  A workgroup of eight waves owns one cell at a time:
//   * its NP producer waves stream the NEXT cell's column of X (1024-gene units, D units in flight per wave, non-temporal 16-byte
//     loads) and append one 32-bit entry per non-zero -- (gene, table index of the value) -- to that cell's list IN LDS; a value
//     that is not in the 1024-entry term table (TPM-like doubles, counts of 256 and more) keeps its 64-bit fixed-point term in
//     a per-workgroup scratch block in global memory, and entries beyond the LDS list's capacity go there too;
//   * its consumer waves walk the list of the CURRENT cell in 64-entry batches exactly as rp_apply_kernel (rp2.hip) does: gene
//     and term of an entry reach their lane group by DPP row broadcast, each lane fetches its 4 or 8 bytes of the gene's packed row
//     list (L2 resident) and adds the term into the cell's K*p accumulators in LDS with ds_add_u64;
//   * two barriers per cell: behind the first (all atomics of cell i landed, the list of cell i + 1 complete) every wave helps
//     convert, scale and store the K*p row of E; behind the second the accumulators are clear.
// Against the two-kernel form this removes the entry lists' round trip through global memory (0.55 GB written and read back per
// 50 000-cell block), the decode of the count, the reservation atomics on global counters, one pass of ballots per unit, one
// launch and one tail per chunk -- and the two instruction streams (HBM stream + compaction, L2 gathers + LDS atomics) interleave
// inside every CU instead of two grids competing for it.  The sum is the same integer sum: E is bit for bit what rp2.hip / rp.hip
// produce (tests/test_rp_gpu.py).
#include "rp_shared.hpp"

#include <cmath>
#include <cstdlib>

namespace sharp {

// Workgroup shapes (two workgroups per CU either way), chosen per projector group by launch_pc:
//   A = 8 waves: 2 producers with 3 units in flight each + 6 consumers (128 registers per lane: the 16-lane x 4-slot row lists of K = 15),
//   B = 12 waves: 4 producers with 2 units in flight + 8 consumers (80 registers per lane: the K = 5 shapes -- 24 waves per CU keep
//       the LDS and the vector memory path busier: cfg3 block 1.44 -> 1.35 ms, cfg4 share 5.71 -> 5.29 ms on one box).
// The producers run at a raised wave priority: the consumers wait for them at the cell's first barrier, never the other way round.
// Entry word: bits 19..0 gene, bits 29..20 table index, bit 31: the value is outside the table and the entry has a 64-bit term of its own.
// fp32 blocks (counts: such entries are rare): the producer wave parks the values in a per-workgroup scratch block in global memory, reads
// them back one lane per entry and stores the terms there (two drained waits per such unit).  fp64 blocks (TPM / CPM-like doubles: EVERY entry is one): the entry travels with its VALUE (the bits of
// the double) in a 64-bit slot per LDS list entry beside the word -- only entries beyond the LDS room go through scratch -- and the CONSUMER
// lane that decodes the entry turns it into the term log2(1 + x) in fixed point (round 5).  Round 4 had the producer wave read its parked values back and convert them itself: two
// s_waitcnt vmcnt(0) per unit (its own stores had to land first, which drained the units it had in flight) and all of an fp64 block's
// log2 evaluations -- ~250 fp64 instructions per 64 entries -- on the two producer waves while six consumer waves waited for them: the fp64
// stage took 3.8 x as long as the fp32 one for twice the bytes, and the two-kernel form was faster.  Now a producer stores and moves on
// (one wait per CELL, and only if something of the cell went to global memory), and the conversions run 64 lanes wide on the waves that
// were waiting.
// The table holds fix(f(x)) for every float x in [1, 256) whose low 16 bits are zero -- every integer count below 256 is one --
// indexed by (bits(x) - bits(1.0f)) >> 16: the index IS the high half of the value's bits, no conversion on either side.
constexpr int PC_TAB = 1024;
constexpr uint32_t kPcGeneMask = 0xfffffu, kPcFull = 0x80000000u, kPcOne = 0x3F800000u;
constexpr uint32_t kPcBadMask = 0xFC00FFFFu;      // (bits - kPcOne) & this != 0: not a table value
// An entry word lives in LDS (the first lcap of a cell) or in the scratch block: accessed through pointers of an explicit address
// space, so that the compiler forms a DS and a GLOBAL operation under the two halves of the test -- left to itself it selects
// between the two generic pointers and emits one FLAT operation, and a pending FLAT operation makes its wait-count pass drain
// every load in flight (s_waitcnt vmcnt(0) in front of the unit fetched three units ago).
typedef __attribute__((address_space(3))) uint32_t pc_lds_u32;
typedef __attribute__((address_space(1))) uint32_t pc_glb_u32;
typedef __attribute__((address_space(1))) long long pc_glb_i64;
typedef __attribute__((address_space(3))) long long pc_lds_i64;
__device__ __forceinline__ void pc_put_word(uint32_t *lst, uint32_t *swb, uint32_t lcap, uint32_t pos, uint32_t word) {
    if (pos < lcap) *(pc_lds_u32 *)(lst + pos) = word;
    else *(pc_glb_u32 *)(swb + pos) = word;
}
__device__ __forceinline__ uint32_t pc_get_word(const uint32_t *lst, const uint32_t *swb, uint32_t lcap, uint32_t pos) {
    uint32_t w;
    if (pos < lcap) w = *(const pc_lds_u32 *)(lst + pos);
    else w = __builtin_nontemporal_load((const pc_glb_u32 *)(swb + pos));
    return w;
}

__global__ void rp_pc_fixtab_kernel(double fix_scale, int log_flag, long long *__restrict__ tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PC_TAB) return;
    const double x = static_cast<double>(__uint_as_float((static_cast<uint32_t>(i) << 16) + kPcOne));
    const double f = log_flag == 2 ? log10(1.0 + x) : (log_flag ? log2(1.0 + x) : x);     // the expressions of the general path
    tab[i] = __double2ll_rn(f * fix_scale);
}

struct PcParams {
    const void *X;
    int m;
    long long ld;
    long long cell0;          // this launch covers cells [cell0, cell0 + ncell) of the block
    int ncell, log_flag;
    double fix_scale;
    const long long *fixtab;
    const uint16_t *ent;
    unsigned int dummy_seg;
    const uint2 *ovf_slot;
    int ncomp, neg_base;
    double inv_fix, val, out_scale;
    double *E;
    long long ldE;
    int comp0;
    const int *row_map;
    int lcap;                 // entries per LDS list (a multiple of 64)
    int cap;                  // entries per scratch list (>= m)
    uint32_t *sw;             // scratch per workgroup: [2 buffers][cap] entry words (those beyond the LDS room of their list) ...
    long long *st;            // ... and [2][cap] 64-bit terms (first the value as a double, then its term)
};

// A unit in flight: the destination registers of LOADS 16-byte loads issued by inline assembly and waited for by a COUNTED
// s_waitcnt.  Issued through the compiler (__builtin_nontemporal_load) the units of a ring of D register sets were drained at
// every use: its wait-count pass puts s_waitcnt vmcnt(2) or vmcnt(0) in front of a unit that has (D - 1) * LOADS younger loads
// behind it (conservative merges over this loop's paths, and a write-after-write guard wherever a destination register is
// reused as a temporary).  The pass neither sees these loads nor this wait; what it inserts for the operations it does see can
// only wait for more, never for less, because its count of younger operations is never above the true one.
typedef uint32_t pc_u4 __attribute__((ext_vector_type(4)));
template <typename T> struct PcUnit { pc_u4 r[CpLayout<T>::LOADS]; };
__device__ __forceinline__ float pc_val(const PcUnit<float> &u, int q) { return __uint_as_float(u.r[q >> 2][q & 3]); }
__device__ __forceinline__ double pc_val(const PcUnit<double> &u, int q) {
    return __hiloint2double(static_cast<int>(u.r[q >> 1][2 * (q & 1) + 1]), static_cast<int>(u.r[q >> 1][2 * (q & 1)]));
}
// `unit` points at the unit's first value (wave-uniform).  lim < CP_UNIT (a cell's last, ragged unit): a load that would run past the
// column's `lim` values reads the unit's first values instead (process() zeroes what lies beyond the last gene).
template <typename T>
__device__ __forceinline__ void pc_issue(PcUnit<T> &u, const T *unit, int lim, int lane) {
    constexpr int V = CpLayout<T>::VEC, L = CpLayout<T>::LOADS;
#define SHARP_PC_LD(j, base, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(u.r[j]) : "v"(voff), "s"(base), "n"(imm))
    if (lim >= CP_UNIT) {
        const uint32_t voff = static_cast<uint32_t>(lane) * 16u;
        SHARP_PC_LD(0, unit, 0); SHARP_PC_LD(1, unit, 1024); SHARP_PC_LD(2, unit, 2048); SHARP_PC_LD(3, unit, 3072);
        if constexpr (L == 8) {
            const T *hi = unit + 4096 / static_cast<int>(sizeof(T));      // (the instruction's offset field ends at 4095)
            SHARP_PC_LD(4, hi, 0); SHARP_PC_LD(5, hi, 1024); SHARP_PC_LD(6, hi, 2048); SHARP_PC_LD(7, hi, 3072);
        }
    } else {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int g = V * (lane + 64 * j);
            const uint32_t voff = static_cast<uint32_t>(g + V - 1 < lim ? g : 0) * static_cast<uint32_t>(sizeof(T));
            SHARP_PC_LD(j, unit, 0);
        }
    }
#undef SHARP_PC_LD
}
// waits until at most N vector-memory operations younger than this unit's loads are outstanding, i.e. until the unit has arrived
template <int N>
__device__ __forceinline__ void pc_wait(PcUnit<float> &u) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(u.r[0]), "+v"(u.r[1]), "+v"(u.r[2]), "+v"(u.r[3]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void pc_wait(PcUnit<double> &u) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(u.r[0]), "+v"(u.r[1]), "+v"(u.r[2]), "+v"(u.r[3]), "+v"(u.r[4]), "+v"(u.r[5]), "+v"(u.r[6]), "+v"(u.r[7]) : "n"(N));
}

// PC_THREADS / 64 waves: NP producer waves with D units in flight each; the other waves consume.
// MODE: 0 signed codes, 1 dual accumulators (ProjectorGroup::neg_base).  (The count-class mode of round 4, measured slower, lives in tools/lab/rp3_cls_lab.hip.)
template <typename T, int GW, int SLOTS, int MODE, int PC_THREADS, int NP, int D>
__global__ __launch_bounds__(PC_THREADS, 2 * (PC_THREADS / 64) / 4) void rp_pc_kernel(const PcParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool DUAL = MODE == 1;
    constexpr bool WIDE = std::is_same<T, double>::value;      // a 64-bit term slot per LDS list entry (layout: [2 lists][lcap words], then [2 lists][lcap terms])
    constexpr int PC_NW = PC_THREADS / 64;
    constexpr int NC = PC_NW - NP, SPAN = SLOTS * GW, U = GW;
    typedef RowWord<SLOTS> Row;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 64-bit slots in front of the control words: the accumulators and their dump slots
    const int nslots = (DUAL ? 2 * P.neg_base : P.ncomp) + kDumpSlots;
    uint32_t *ctl = reinterpret_cast<uint32_t *>(acc + nslots);     // [0], [1]: entries in list 0 / 1; [2] E row; [3] producers done
    uint32_t *lists = ctl + 8;
    long long *lterms = reinterpret_cast<long long *>(lists + 2 * P.lcap);       // (WIDE only; 8-byte aligned: everything before it is a multiple of 8 bytes)
    for (int c = tid; c < nslots; c += PC_THREADS) acc[c] = 0ull;
    if (tid < 8) ctl[tid] = 0u;
    __syncthreads();
    // cell j of this workgroup (static: a workgroup takes ~100 cells, their costs average out)
    const int G = static_cast<int>(gridDim.x);
    const int nmine = (P.ncell - static_cast<int>(blockIdx.x) + G - 1) / G;
    auto cell_of = [&](int j) __attribute__((always_inline)) -> long long { return P.cell0 + static_cast<long long>(blockIdx.x) + static_cast<long long>(j) * G; };
    uint32_t *const sw = P.sw + static_cast<size_t>(blockIdx.x) * 2 * P.cap;
    long long *const st = P.st + static_cast<size_t>(blockIdx.x) * 2 * P.cap;

    // End of a cell for EVERY wave: behind the first barrier all atomics of cell `it` have landed and the list of cell it + 1 is
    // complete; the consumer waves convert, scale and store the K*p row of E (read and clear in one LDS operation); behind the second
    // barrier the accumulators are clear.  The producer waves only pass the barriers: a store among their pending vector-memory
    // operations would make the compiler's wait-count pass drain the units they have in flight (mixed loads and stores count as
    // out of order), and the E row index reaches everybody through LDS (ctl[2], put there by the first consumer wave) for the same reason.
    auto cell_end = [&](int it) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (it >= 0 && wave >= NP) {
            const long long row = static_cast<long long>(static_cast<int>(ctl[2]));
            double *erow = P.E + row * P.ldE + P.comp0;
            for (int c = tid - NP * 64; c < P.ncomp; c += PC_THREADS - NP * 64) {
                long long a = static_cast<long long>(atomicExch(&acc[c], 0ull));
                if constexpr (DUAL) a -= static_cast<long long>(atomicExch(&acc[P.neg_base + c], 0ull));
                __builtin_nontemporal_store(P.out_scale * (P.val * (static_cast<double>(a) * P.inv_fix)), &erow[c]);
            }
            if (tid == NP * 64) ctl[it & 1] = 0u;     // that list has been consumed: the cell after next appends to it
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    if (wave >= NP) {
        // ------------------------------------------------------------------ consumers
        const int cw = wave - NP;
        const int lg = lane % GW;
        const unsigned char *entb = reinterpret_cast<const unsigned char *>(P.ent);
        cell_end(-1);                                   // (the producers compact the first cell)
        // The cell being consumed: its list, its entry count, and this wave's four-stage pipeline over the list's batches (while the
        // atomics of batch b run, the row lists of batch b + NC, the terms of batch b + 2 NC and the entry words of batch b + 3 NC are in
        // flight).  A cell's first batches are set up BEFORE the previous cell's barriers and epilogue whenever the producers have
        // that cell's list complete by then (ctl[3] counts their completions; they normally run a cell ahead): the dependent round
        // trips -- entry word, term, row list -- then travel under the barriers and the epilogue instead of behind them.
        const uint32_t *lst = lists;
        const uint32_t *swb = sw;                         // scratch words of this cell's list
        const long long *stb = st;
        const long long *ltm = lterms;                    // (WIDE) the LDS term slots of this cell's list
        int nnz = 0, nb = 0;
        const int gcap = P.lcap;
        uint32_t gC = 0u, gL = 0u, wR = 0u;   // genes of the batch being added / of the one after the next; entry words of the one after that
        long long fC = 0ll, fL = 0ll;         // their terms
        Row cd[U], cdn[U];
        auto load_word = [&](int bt) __attribute__((always_inline)) -> uint32_t {                 // lane = entry of batch bt; unconditional, clamped
            const int e = (bt << 6) + lane;
            const int ec = e < nnz ? e : 0;
            if ((bt << 6) + 64 <= gcap) return *(const pc_lds_u32 *)(lst + ec);
            return pc_get_word(lst, swb, static_cast<uint32_t>(gcap), static_cast<uint32_t>(ec));
        };
        auto decode = [&](int bt, uint32_t w, uint32_t &g, long long &f) __attribute__((always_inline)) {
            const int e = (bt << 6) + lane;
            long long ff = P.fixtab[(w >> 20) & 0x3ffu];            // 8 KB, cache resident
            if (__ballot((w & kPcFull) != 0u) != 0ull) {           // wave-uniform test: a value outside the table (rare in an fp32 block, all of an fp64 block)
                if (w & kPcFull) {
                    const int ec = e < nnz ? e : 0;
                    if constexpr (WIDE) {                          // the entry carries its VALUE: the expressions of the general path, one lane per entry
                        long long bits;
                        if (ec < gcap) bits = *(const pc_lds_i64 *)(ltm + ec);
                        else bits = __builtin_nontemporal_load((const pc_glb_i64 *)(stb + ec));
                        const double x = __longlong_as_double(bits);
                        const double fv = P.log_flag == 2 ? log10(1.0 + x) : (P.log_flag ? log2(1.0 + x) : x);
                        ff = __double2ll_rn(fv * P.fix_scale);
                    } else {                                       // (fp32 blocks: the producer stored the term)
                        ff = __builtin_nontemporal_load((const pc_glb_i64 *)(stb + ec));
                    }
                }
            }
            g = e < nnz ? (w & kPcGeneMask) : P.dummy_seg;
            f = e < nnz ? ff : 0ll;
        };
        auto load_lists = [&](uint32_t g, Row (&dst)[U]) __attribute__((always_inline)) {
            const uint32_t gofs = g * static_cast<uint32_t>(SPAN * 2);
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                dst[u] = load_row_word<SLOTS>(entb + (group_bcast<GW, u>(gofs) + static_cast<uint32_t>(2 * SLOTS * lg)));
            });
        };
        auto step = [&](int bt, Row (&cur)[U], Row (&nxt)[U]) __attribute__((always_inline)) {
            const uint32_t gN = gL;
            const long long fN = fL;
            load_lists(gN, nxt);
            decode(bt + 2 * NC, wR, gL, fL);
            asm volatile("" : "+v"(gL));
            wR = load_word(bt + 3 * NC);
            const uint32_t plo = static_cast<uint32_t>(fC), phi = static_cast<uint32_t>(static_cast<unsigned long long>(fC) >> 32);
            uint32_t more = 0u;
            if constexpr (DUAL) {
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const uint32_t lo = group_bcast<GW, u>(plo), hi = group_bcast<GW, u>(phi);
                    scatter_row_word<0, SLOTS, false>(cur[u], (static_cast<unsigned long long>(hi) << 32) | lo);
                    more |= cur[u].x;
                });
            } else {
                const long long fneg = -fC;
                const uint32_t nlo = static_cast<uint32_t>(fneg), nhi = static_cast<uint32_t>(static_cast<unsigned long long>(fneg) >> 32);
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const bool neg = (cur[u].x & kCodeNeg) != 0u;     // a lane's codes share their sign
                    const uint32_t bpl = group_bcast<GW, u>(plo), bph = group_bcast<GW, u>(phi);
                    const uint32_t bnl = group_bcast<GW, u>(nlo), bnh = group_bcast<GW, u>(nhi);
                    const uint32_t lo = neg ? bnl : bpl, hi = neg ? bnh : bph;
                    scatter_row_word<0, SLOTS, true>(cur[u], (static_cast<unsigned long long>(hi) << 32) | lo);
                    more |= cur[u].x;
                });
            }
            // rare: a gene continues in overflow segments (flag in slot 0 of its first lane); one scalar test per batch
            if (__ballot((more & kCodeMore) != 0u) != 0ull) {
                static_for<U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const unsigned long long full = __ballot((cur[u].x & kCodeMore) != 0u);
                    const uint32_t g = group_bcast<GW, u>(gC);
                    const uint32_t bpl = group_bcast<GW, u>(plo), bph = group_bcast<GW, u>(phi);
                    if ((full >> (lane & ~(GW - 1))) & 1ull) {
                        const long long f = static_cast<long long>((static_cast<unsigned long long>(bph) << 32) | bpl);
                        const uint2 oi = P.ovf_slot[g];
                        for (uint32_t sg = 0; sg < oi.y; ++sg) {
                            const Row c2 = load_row_word<SLOTS>(P.ent + (static_cast<size_t>(oi.x) + sg) * SPAN + SLOTS * lg);
                            if constexpr (DUAL) scatter_row_word<0, SLOTS, false>(c2, static_cast<unsigned long long>(f));
                            else scatter_row_word<0, SLOTS, true>(c2, static_cast<unsigned long long>((c2.x & kCodeNeg) ? -f : f));
                        }
                    }
                });
            }
            gC = gN;
            fC = fN;
        };
        auto begin_cell = [&](int it2) __attribute__((always_inline)) {     // the list of cell it2 is complete
            const int b2 = it2 & 1;
            lst = lists + b2 * P.lcap;
            swb = sw + static_cast<size_t>(b2) * P.cap;
            stb = st + static_cast<size_t>(b2) * P.cap;
            ltm = lterms + b2 * P.lcap;
            nnz = __builtin_amdgcn_readfirstlane(static_cast<int>(ctl[b2]));
            nb = (nnz + 63) >> 6;
            if (cw < nb) {
                const uint32_t w0 = load_word(cw), w1 = load_word(cw + NC);
                wR = load_word(cw + 2 * NC);
                decode(cw, w0, gC, fC);
                decode(cw + NC, w1, gL, fL);
                load_lists(gC, cd);
            }
        };
        bool ahead = false;                              // the cell about to be consumed has been set up already
        for (int it = 0; it < nmine; ++it) {
            if (cw == 0 && lane == 0) ctl[2] = static_cast<uint32_t>(P.row_map ? P.row_map[cell_of(it)] : static_cast<int>(cell_of(it)));
            if (!ahead) begin_cell(it);
            if (cw < nb) {
                for (int bt = cw; bt < nb; bt += 2 * NC) {          // two steps per trip: the two row-list sets swap roles
                    step(bt, cd, cdn);
                    if (bt + NC < nb) step(bt + NC, cdn, cd);
                }
            }
            ahead = false;
            if (it + 1 < nmine) {
                const int done = __builtin_amdgcn_readfirstlane(static_cast<int>(ctl[3]));
                if (done >= NP * (it + 2)) { begin_cell(it + 1); ahead = true; }
            }
            cell_end(it);
        }
    } else {
        // ------------------------------------------------------------------ producers
        __builtin_amdgcn_s_setprio(3);
        const T *X = static_cast<const T *>(P.X);
        const int units = (P.m + CP_UNIT - 1) / CP_UNIT;
        const int upp = (units + NP - 1) / NP;            // items (units) per cell and producer; unit = wave + k * NP
        const long long total = static_cast<long long>(nmine) * upp;
        int fj = 0, fk = 0;                               // the item to fetch next
        auto fetch = [&](PcUnit<T> &u) __attribute__((always_inline)) {   // every load unconditional: past the last item, the last unit again
            const int j = fj < nmine ? fj : nmine - 1;
            int uu = wave + fk * NP;
            if (uu >= units) uu = units - 1;
            const T *unit = X + cell_of(j) * P.ld + static_cast<long long>(uu) * CP_UNIT;
            // values of the column (its leading dimension) from the unit's first on: a whole unit of them is fetched whole, gene or padding
            const long long left = P.ld - static_cast<long long>(uu) * CP_UNIT;
            pc_issue<T>(u, unit, static_cast<int>(std::min<long long>(left, CP_UNIT)), lane);
            if (++fk == upp) { fk = 0; ++fj; }
        };
        // `live`: the item exists (past a workgroup's last item, and for the units a producer has fewer of than upp, the fetch repeats the
        // last unit).  Every path through here READS all sixteen values before it returns: a value nobody waited for would still be
        // in flight when the next fetch overwrites its register, and the compiler protects that write with a wait that drains
        // the other units in flight as well.
        bool spilled = false;      // this wave stored words / values of the current cell to the scratch block: they have to land before the cell is handed over
        auto process = [&](PcUnit<T> &u, int j, int k, bool live) __attribute__((always_inline)) {
            pc_wait<(D - 1) * CpLayout<T>::LOADS>(u);      // the unit has arrived; the D - 1 units fetched after it stay in flight
            CpVals<T> b;
#pragma unroll
            for (int q = 0; q < 16; ++q) b.v[q] = pc_val(u, q);
            const int unit = wave + k * NP;
            const int ubase = unit * CP_UNIT;
            const int glim = (live && unit < units) ? P.m - ubase : 0;      // genes of this unit (a cell's last unit is ragged)
            if (glim < CP_UNIT) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (CpLayout<T>::gene_of(lane, q) >= glim) b.v[q] = T(0);
            }
            unsigned long long mk[16];
            int tot = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { mk[q] = __ballot(b.v[q] != T(0)); tot += __popcll(mk[q]); }
            if (tot == 0) return;
            const int b01 = j & 1;
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(&ctl[b01], static_cast<uint32_t>(tot));
            base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
            uint32_t *lst = lists + b01 * P.lcap;
            if constexpr (std::is_same<T, float>::value) {
                if (base + static_cast<uint32_t>(tot) <= static_cast<uint32_t>(P.lcap)) {
                    // the usual path: every value a table value, every entry in LDS
                    uint32_t run = base, bad = 0u;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (b.v[q] != 0.0f) {
                            const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mk[q]), 0u));
                            const uint32_t t = __float_as_uint(b.v[q]) - kPcOne;
                            *(pc_lds_u32 *)(lst + pos) = (t << 4) + static_cast<uint32_t>(ubase + CpLayout<T>::gene_of(lane, q));
                            bad |= t & kPcBadMask;
                        }
                        run += static_cast<uint32_t>(__popcll(mk[q]));
                    }
                    if (__ballot(bad != 0u) == 0ull) return;
                }
            }
            // The general path (a value outside the table somewhere in the unit, an fp64 block, or a list beyond the LDS room): every
            // non-zero again -- table values as above, the others flagged kPcFull with the value's bits as a double parked where the
            // consumer will look for them: the entry's 64-bit LDS slot (fp64 blocks), else the scratch block.  Nothing is read back here.
            uint32_t *swb = sw + static_cast<size_t>(b01) * P.cap;
            long long *stb = st + static_cast<size_t>(b01) * P.cap;
            long long *ltm = lterms + b01 * P.lcap;
            const uint32_t lcap = static_cast<uint32_t>(P.lcap);
            {
                uint32_t run = base;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (b.v[q] != T(0)) {
                        const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mk[q] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mk[q]), 0u));
                        const float xf = static_cast<float>(b.v[q]);
                        const uint32_t t = __float_as_uint(xf) - kPcOne;
                        const bool tab = static_cast<T>(xf) == b.v[q] && (t & kPcBadMask) == 0u;
                        const uint32_t gene = static_cast<uint32_t>(ubase + CpLayout<T>::gene_of(lane, q));
                        const uint32_t word = tab ? (t << 4) + gene : (gene | kPcFull);
                        pc_put_word(lst, swb, lcap, pos, word);
                        if (!tab) {
                            const long long bits = __double_as_longlong(static_cast<double>(b.v[q]));
                            if (WIDE && pos < lcap) *(pc_lds_i64 *)(ltm + pos) = bits;
                            else *(pc_glb_i64 *)(stb + pos) = bits;
                        }
                    }
                    run += static_cast<uint32_t>(__popcll(mk[q]));
                }
            }
            if constexpr (WIDE) {
                if (base + static_cast<uint32_t>(tot) > lcap) spilled = true;          // (wave-uniform) something of this cell is in global memory
            } else {
                // fp32 blocks (rare: a count of 256 or more, a non-integer): this wave turns the values it parked into their terms itself, one lane
                // per entry -- its own stores have to land first, and that wait drains the units it has in flight: twice per such unit
                __builtin_amdgcn_s_waitcnt(0x0070);             // vmcnt(0) lgkmcnt(0)
                asm volatile("" ::: "memory");
                for (uint32_t e = base + static_cast<uint32_t>(lane); e < base + static_cast<uint32_t>(tot); e += 64u) {
                    const uint32_t word = pc_get_word(lst, swb, lcap, e);
                    if (word & kPcFull) {
                        const double x = __longlong_as_double(__builtin_nontemporal_load((const pc_glb_i64 *)(stb + e)));
                        const double f = P.log_flag == 2 ? log10(1.0 + x) : (P.log_flag ? log2(1.0 + x) : x);
                        *(pc_glb_i64 *)(stb + e) = __double2ll_rn(f * P.fix_scale);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): the scratch block is complete before this wave reaches the barrier
                asm volatile("" ::: "memory");                  // (a builtin, not inline assembly: the wait-count pass sees it and forgets the stores)
            }
        };
        PcUnit<T> buf[D];
        static_for<D>([&](auto ic) { fetch(buf[decltype(ic)::value]); });
        int pj = 0, pk = 0;                               // the item being worked on
        for (long long t = 0; t < total; t += D) {
            static_for<D>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const bool live = t + i < total;
                process(buf[i], pj, pk, live);
                fetch(buf[i]);                            // (unconditional: the loads in flight are the same on every path)
                if (live && ++pk == upp) {
                    pk = 0;
                    if (spilled) {                            // once per cell, and only for a cell that spilled: vmcnt(0), the scratch block is complete
                        __builtin_amdgcn_s_waitcnt(0x0f70);   // (a builtin, not inline assembly: the wait-count pass sees it and forgets the stores)
                        asm volatile("" ::: "memory");
                        spilled = false;
                    }
                    if (lane == 0) atomicAdd(&ctl[3], 1u);    // this wave's entries of cell pj are in the list (LDS operations of a wave execute in order)
                    cell_end(pj - 1);
                    ++pj;
                }
            });
        }
        static_for<D>([&](auto ic) { pc_wait<0>(buf[decltype(ic)::value]); });      // (nothing of this wave's is in flight when it ends)
        cell_end(nmine - 1);
    }
}

namespace {
struct PcWs {
    DevBuf<uint32_t> sw;
    DevBuf<long long> st;
    DevBuf<long long> fixtab;
    double fixtab_key = 0.0;
    int fixtab_mode = -1;
};
PcWs &pws() { return per_slot<PcWs>(); }

template <typename T, int GW, int SLOTS, int MODE>
void launch_pc(const ProjectorGroup &g, const Projector &pr, const PcParams &P0, int n, hipStream_t st) {
    Ctx &c = ctx();
    PcWs &W = pws();
    PcParams P = P0;
    // the workgroup shape (above rp_pc_kernel): B for the narrow row lists (K = 5), A for 16 lanes x 4 slots (K = 15) and fp64 blocks
    // (an fp64 unit is 32 registers per lane); SHARP_RP_PC_SHAPE=a / b forces one for fp32 blocks
    const bool wide_rows = g.gw == 16 && g.slots == 4;
    const bool shape_b = !std::is_same<T, double>::value && (knobs().rp_pc_shape == 2 || (knobs().rp_pc_shape == 0 && !wide_rows));
#ifndef SHARP_PC_B_THREADS      // (tools/build_variant.sh: other splits of shape B)
#define SHARP_PC_B_THREADS 768
#define SHARP_PC_B_NP 4
#endif
#ifndef SHARP_PC_B_D
#define SHARP_PC_B_D 2
#endif
#ifndef SHARP_PC_A_NP                     // shape A (512 threads): producer waves and ring depth (variant builds sweep them)
#define SHARP_PC_A_NP 2
#define SHARP_PC_A_D 3
#endif
    const int threads = shape_b ? SHARP_PC_B_THREADS : 512;
    const void *kern = nullptr;
    if constexpr (std::is_same<T, double>::value) kern = reinterpret_cast<const void *>(rp_pc_kernel<T, GW, SLOTS, MODE, 512, 2, 2>);
    else kern = shape_b ? reinterpret_cast<const void *>(rp_pc_kernel<T, GW, SLOTS, MODE, SHARP_PC_B_THREADS, SHARP_PC_B_NP, SHARP_PC_B_D>)
                        : reinterpret_cast<const void *>(rp_pc_kernel<T, GW, SLOTS, MODE, 512, SHARP_PC_A_NP, SHARP_PC_A_D>);
    // LDS per workgroup: accumulators + dump slots, eight control words, two entry lists; two workgroups per CU
    const size_t acc_bytes = static_cast<size_t>(g.acc_slots() + kDumpSlots) * 8 + 32;
    const size_t budget = 80 * 1024;
    SHARP_REQUIRE(acc_bytes + 2 * 64 * 12 <= budget, "rp_pc_kernel: the accumulators leave no LDS for the entry lists");
    constexpr bool wide = std::is_same<T, double>::value;     // fp64 blocks: a word and a 64-bit term slot per list entry (rp_pc_kernel: WIDE)
    const size_t per_entry = wide ? 12 : 4;
    int lcap = static_cast<int>((budget - acc_bytes) / (2 * per_entry) / 64 * 64);
    lcap = std::min(lcap, (P.m + 63) / 64 * 64);
    P.lcap = lcap;
    const size_t lds = acc_bytes + static_cast<size_t>(lcap) * 2 * per_entry;
    SHARP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    {   // scatter_row_word<0, ...>: the accumulators sit at LDS address 0, i.e. the kernel must not have static LDS in front of the dynamic block
        hipFuncAttributes fa;
        SHARP_HIP_CHECK(hipFuncGetAttributes(&fa, kern));
        SHARP_REQUIRE(fa.sharedSizeBytes == 0, "rp_pc_kernel: static LDS in front of the accumulators");
    }
    int per_cu = 1;
    SHARP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
    per_cu = std::max(1, std::min(per_cu, 2));
    per_cu = std::min(per_cu, std::max(1, knobs().rp_pc_wgs));
    if (c.rp_wgs_cap > 0) per_cu = std::min(per_cu, c.rp_wgs_cap);
    const int grid = c.num_cu * per_cu;
    P.cap = (P.m + 3) / 4 * 4;
    W.sw.ensure(static_cast<size_t>(c.num_cu) * 2 * 2 * P.cap);       // per workgroup: [2 buffers][cap] words, [2][cap] terms
    W.st.ensure(static_cast<size_t>(c.num_cu) * 2 * 2 * P.cap);
    P.sw = W.sw.p;
    P.st = W.st.p;
    // One launch for the whole block -- except for a block prepared under another block's tail (Ctx::polite, SHARP_unlimited block
    // after block): a persistent grid that holds every CU's LDS for milliseconds keeps the tail's whole-CU workgroups waiting whatever
    // the stream priorities, so there it goes out in slices of sixteen cells per workgroup and the queue drains between them.
    const long long slice = c.polite ? static_cast<long long>(grid) * 16 : static_cast<long long>(n);
    for (long long c0 = 0; c0 < n; c0 += slice) {
        P.cell0 = c0;
        P.ncell = static_cast<int>(std::min<long long>(slice, n - c0));
        const int blocks = std::min(P.ncell, grid);
        void *args[] = {&P};
        SHARP_HIP_CHECK(hipLaunchKernel(kern, dim3(static_cast<unsigned>(blocks)), dim3(static_cast<unsigned>(threads)), args, lds, st));
        launch_check("rp_pc_kernel");
    }
}
}  // namespace

// The default form of the RP matmul (SHARP_RP_KERNEL unset or "pc") wherever X can be read with 16-byte loads and a gene index fits
// the entry word; "fused" / "dense" name the other forms (lab builds: "split", the two-kernel form of tools/lab/rp2.hip).
bool rp_pc_eligible(XRef X, int m, long long ld) {
    const bool vec = (ld % (X.f64 ? 2 : 4) == 0) && ((reinterpret_cast<uintptr_t>(X.p) & 15u) == 0);
    const int k = knobs().rp_kernel;
    return (k == 0 || k == 3 || k == 4 || (X.f64 && k != 2)) && vec && m > 16 && m <= (1 << 20);   // (an fp64 block has no other sparse form)
}

// One projector group per call; X 16-byte aligned with ld % 4 == 0 (fp32) / ld % 2 == 0 (fp64).
void project_dev_pc(const Projector &pr, const ProjectorGroup &g, XRef dX, int m, int n, long long ld, int log_flag, int fix_bits,
                    double *dE, long long ldE, const int *d_row_map) {
    Ctx &c = ctx();
    PcWs &W = pws();
    const double fix_scale = std::ldexp(1.0, fix_bits), inv_fix = std::ldexp(1.0, -fix_bits);
    if (W.fixtab.n == 0 || W.fixtab_key != fix_scale || W.fixtab_mode != log_flag) {
        W.fixtab.ensure(PC_TAB);
        hipLaunchKernelGGL(rp_pc_fixtab_kernel, dim3(PC_TAB / 256), dim3(256), 0, c.stream, fix_scale, log_flag, W.fixtab.p);
        launch_check("rp_pc_fixtab_kernel");
        W.fixtab_key = fix_scale;
        W.fixtab_mode = log_flag;
    }
    PcParams P;
    P.X = dX.p; P.m = m; P.ld = ld; P.cell0 = 0; P.ncell = n; P.log_flag = log_flag; P.fix_scale = fix_scale; P.fixtab = W.fixtab.p;
    P.ent = g.ent.p; P.dummy_seg = static_cast<unsigned int>(g.nseg); P.ovf_slot = g.ovf_slot.p; P.ncomp = g.ncomp; P.neg_base = g.neg_base;
    P.inv_fix = inv_fix; P.val = pr.val; P.out_scale = 1.0 / std::sqrt(static_cast<double>(pr.p));
    P.E = dE; P.ldE = ldE; P.comp0 = g.k0 * pr.p; P.row_map = d_row_map;
    P.lcap = 0; P.cap = 0; P.sw = nullptr; P.st = nullptr;
    KernelTimer t("rp_stage");
    KernelTimer t2("rp_pc");
    const int mode = g.neg_base > 0 ? 1 : 0;
#define SHARP_PCL(TT, GWV, SL, MD) launch_pc<TT, GWV, SL, MD>(g, pr, P, n, c.stream)
#define SHARP_PCM(TT, GWV, SL) { if (mode == 1) SHARP_PCL(TT, GWV, SL, 1); else SHARP_PCL(TT, GWV, SL, 0); }
#define SHARP_PCT(TT)                                        \
    if (g.gw == 16 && g.slots == 4) SHARP_PCM(TT, 16, 4)     \
    else if (g.gw == 16) SHARP_PCM(TT, 16, 2)                \
    else if (g.gw == 8) SHARP_PCM(TT, 8, 4)                  \
    else SHARP_PCM(TT, 4, 4)
    if (dX.f64) { SHARP_PCT(double) } else { SHARP_PCT(float) }
#undef SHARP_PCM
#undef SHARP_PCT
#undef SHARP_PCL
}

void rp_pc_trim() {
    PcWs &W = pws();
    W.sw.release();
    W.st.release();
}

}  // namespace sharp
