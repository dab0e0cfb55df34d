// rp_shared.hpp -- device helpers shared by the RP matmul kernels (rp2.hip: compaction + apply as two kernels; rp3.hip: the same
// work as one persistent producer / consumer kernel): how a wave loads a 1024-gene unit of a cell, compile-time loops, and the
// DPP row broadcast that hands entry u of a lane group to the group's lanes.
#pragma once
#include <type_traits>

#include "projector.hpp"

namespace sharp {

constexpr int CP_UNIT = 1024;           // genes per wave unit = 64 lanes x 16 values

// One unit = 1024 genes = 16 values per lane, fetched with 16-byte loads: fp32 blocks as 4 x float4 (lane l, load j: genes
// 4 (l + 64 j) ..), fp64 blocks as 8 x double2 (genes 2 (l + 64 j) ..).  gene_of(q) is the gene of a lane's q-th value.
template <typename T> struct CpVals { T v[16]; };
template <typename T> struct CpLayout;
template <> struct CpLayout<float> {
    static constexpr int VEC = 4, LOADS = 4;
    __device__ static __forceinline__ int gene_of(int lane, int q) { return 4 * lane + 256 * (q >> 2) + (q & 3); }
};
template <> struct CpLayout<double> {
    static constexpr int VEC = 2, LOADS = 8;
    __device__ static __forceinline__ int gene_of(int lane, int q) { return 2 * lane + 128 * (q >> 1) + (q & 1); }
};

// `unit` points at the unit's first gene.  CLAMP (only a cell's last, ragged unit): a load that would run past the column's `lim`
// values (the leading dimension, a multiple of VEC) reads the unit's first values instead -- every load is unconditional -- and the
// caller zeroes what lies beyond the last gene.  The other units need no address arithmetic at all: one lane offset, the load's
// immediate offset, a scalar base.
template <typename T, bool CLAMP>
__device__ __forceinline__ CpVals<T> cp_load_unit(const T *unit, int lim, int lane) {
    typedef T tv __attribute__((ext_vector_type(CpLayout<T>::VEC)));
    constexpr int V = CpLayout<T>::VEC;
    CpVals<T> r;
#pragma unroll
    for (int j = 0; j < CpLayout<T>::LOADS; ++j) {
        int g = V * (lane + 64 * j);
        if (CLAMP) g = g + V - 1 < lim ? g : 0;
        const tv t = __builtin_nontemporal_load(reinterpret_cast<const tv *>(unit + g));
#pragma unroll
        for (int e = 0; e < V; ++e) r.v[V * j + e] = t[e];
    }
    return r;
}

// compile-time loop: body(std::integral_constant<int, u>) for u = 0 .. N-1 (a DPP control word must be a constant expression)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&body) {
    if constexpr (I < N) {
        body(std::integral_constant<int, I>{});
        static_for<N, I + 1>(body);
    }
}

// Entry u of a lane group's own GW entries, broadcast to the group's GW lanes without touching the LDS: the group's entries sit in
// the group's own lanes, a group is (part of) one DPP row of 16 lanes, and `row_newbcast:n` copies lane n of every row to the row's
// lanes; groups narrower than a row take their own part through the bank mask (one bank = 4 lanes).
template <int GW, int U_>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x) {
    static_assert(GW == 16 || GW == 8 || GW == 4, "a lane group is 4, 8 or 16 lanes");
    // (the first move leaves the lanes outside its bank mask undefined -- no zero-initialised destination register -- and the
    // following moves complete them: together the bank masks cover the row)
    int v = __builtin_amdgcn_mov_dpp(static_cast<int>(x), 0x150 + U_, 0xf, GW == 16 ? 0xf : (GW == 8 ? 0x3 : 0x1), true);
    if constexpr (GW == 8) {
        v = __builtin_amdgcn_update_dpp(v, static_cast<int>(x), 0x150 + 8 + U_, 0xf, 0xc, true);
    } else if constexpr (GW == 4) {
        v = __builtin_amdgcn_update_dpp(v, static_cast<int>(x), 0x150 + 4 + U_, 0xf, 0x2, true);
        v = __builtin_amdgcn_update_dpp(v, static_cast<int>(x), 0x150 + 8 + U_, 0xf, 0x4, true);
        v = __builtin_amdgcn_update_dpp(v, static_cast<int>(x), 0x150 + 12 + U_, 0xf, 0x8, true);
    }
    return static_cast<uint32_t>(v);
}

}  // namespace sharp
