// Micro-benchmark: how many cycles does one wave64 vector-ALU instruction hold its SIMD on gfx950?  A block of W waves per SIMD runs an
// unrolled stream of independent instructions of one kind; the per-SIMD cost per instruction is (cycles x SIMD-share) / instructions.
// Kinds: 0 v_and_b32, 1 v_mov_b32_dpp row_newbcast, 2 v_cndmask_b32, 3 v_cmp (to SGPR), 4 v_add_u32, 5 v_lshl_add_u32, 6 v_cvt_u32_f32,
//        7 v_mbcnt_lo, 8 mixed and+dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int KIND>
__global__ void k(unsigned *out, long long *cyc, int reps) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#define REP8(S) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc", "s12", "s13");
#define I0(n) "v_and_b32 %" #n ", 0xfffff8, %" #n "\n"
#define I1(n) "v_mov_b32_dpp %" #n ", %" #n " row_newbcast:3 row_mask:0xf bank_mask:0x3\n"
#define I2(n) "v_cndmask_b32 %" #n ", %" #n ", %" #n ", vcc\n"
#define I3(n) "v_cmp_gt_u32_e64 s[12:13], 7, %" #n "\n"
#define I4(n) "v_add_u32 %" #n ", 5, %" #n "\n"
#define I5(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %" #n "\n"
#define I6(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I7(n) "v_mbcnt_lo_u32_b32 %" #n ", -1, %" #n "\n"
#define I9(n) "v_cmp_ne_u32 vcc, 0, %" #n "\n v_cndmask_b32 %" #n ", %" #n ", %" #n ", vcc\n"
#define I10(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %" #n ", s[12:13]\n"
#define I11(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 8\n"
#define I12(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define I13(n) "v_and_b32_sdwa %" #n ", %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I14(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " row_newbcast:3 row_mask:0xf bank_mask:0x3\n"
#define I15(n) "v_cmp_ne_u32 vcc, 0, %" #n "\n"
#define I16(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I17(n) "v_bfe_u32 %" #n ", %" #n ", 3, 8\n"
#define I18(n) "v_or3_b32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I19(n) "v_mbcnt_hi_u32_b32 %" #n ", -1, %" #n "\n"
#define I20(n) "v_cmpx_ne_u32 0, %" #n "\n s_mov_b64 exec, -1\n"
#define I21(n) "v_addc_co_u32 %" #n ", vcc, 0, %" #n ", vcc\n"
#define I22(n) "v_cmp_neq_f32 vcc, 0, %" #n "\n"
#define I23(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %" #n "\n"
#define I24(n) "v_xor_b32 %" #n ", 0xfffff8, %" #n "\n"
#define I25(n) "v_sub_co_u32 %" #n ", vcc, 0, %" #n "\n"
        if constexpr (KIND == 0) { REP8(I0) REP8(I0) REP8(I0) REP8(I0) }
        if constexpr (KIND == 1) { REP8(I1) REP8(I1) REP8(I1) REP8(I1) }
        if constexpr (KIND == 2) { REP8(I2) REP8(I2) REP8(I2) REP8(I2) }
        if constexpr (KIND == 3) { REP8(I3) REP8(I3) REP8(I3) REP8(I3) }
        if constexpr (KIND == 4) { REP8(I4) REP8(I4) REP8(I4) REP8(I4) }
        if constexpr (KIND == 5) { REP8(I5) REP8(I5) REP8(I5) REP8(I5) }
        if constexpr (KIND == 6) { REP8(I6) REP8(I6) REP8(I6) REP8(I6) }
        if constexpr (KIND == 7) { REP8(I7) REP8(I7) REP8(I7) REP8(I7) }
        if constexpr (KIND == 9) { REP8(I9) REP8(I9) REP8(I9) REP8(I9) }
        if constexpr (KIND == 10) { REP8(I10) REP8(I10) REP8(I10) REP8(I10) }
        if constexpr (KIND == 11) { REP8(I11) REP8(I11) REP8(I11) REP8(I11) }
        if constexpr (KIND == 12) { REP8(I12) REP8(I12) REP8(I12) REP8(I12) }
        if constexpr (KIND == 13) { REP8(I13) REP8(I13) REP8(I13) REP8(I13) }
        if constexpr (KIND == 14) { REP8(I14) REP8(I14) REP8(I14) REP8(I14) }
        if constexpr (KIND == 15) { REP8(I15) REP8(I15) REP8(I15) REP8(I15) }
        if constexpr (KIND == 16) { REP8(I16) REP8(I16) REP8(I16) REP8(I16) }
        if constexpr (KIND == 17) { REP8(I17) REP8(I17) REP8(I17) REP8(I17) }
        if constexpr (KIND == 18) { REP8(I18) REP8(I18) REP8(I18) REP8(I18) }
        if constexpr (KIND == 19) { REP8(I19) REP8(I19) REP8(I19) REP8(I19) }
        if constexpr (KIND == 20) { REP8(I20) REP8(I20) REP8(I20) REP8(I20) }
        if constexpr (KIND == 21) { REP8(I21) REP8(I21) REP8(I21) REP8(I21) }
        if constexpr (KIND == 22) { REP8(I22) REP8(I22) REP8(I22) REP8(I22) }
        if constexpr (KIND == 23) { REP8(I23) REP8(I23) REP8(I23) REP8(I23) }
        if constexpr (KIND == 24) { REP8(I24) REP8(I24) REP8(I24) REP8(I24) }
        if constexpr (KIND == 25) { REP8(I25) REP8(I25) REP8(I25) REP8(I25) }
        if constexpr (KIND == 8) { REP8(I0) REP8(I1) REP8(I0) REP8(I1) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    unsigned *out; long long *cyc;
    CK(hipMalloc(&out, 1024 * 1024 * 4)); CK(hipMalloc(&cyc, 1024 * 16 * 8));
    const int reps = 2000;
    for (int wps : {1, 2, 4}) {            // waves per SIMD: a block of 256 * wps threads, one block per CU
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipDeviceSynchronize());
        long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        // s_memtime ticks at 100 MHz on this part?  report raw ticks per instruction per wave, and per SIMD (divide by waves/SIMD)
        printf("%-14s waves/SIMD %d: %.3f ticks per instr per wave, %.3f per instr per SIMD\n", name, wps, mx / (reps * 32.0), mx / (reps * 32.0) / wps);
    }
    CK(hipFree(out)); CK(hipFree(cyc));
}

template <int KIND>
__global__ void k64(unsigned long long *out, long long *cyc, int reps) {
    unsigned long long a0 = threadIdx.x * 0x9e3779b97f4a7c15ull, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    double d0 = threadIdx.x * 1.5, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#define R4(S) asm volatile(S(0) S(1) S(2) S(3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc", "s12", "s13");
#define D4(S) asm volatile(S(0) S(1) S(2) S(3) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) :: "vcc", "s12", "s13");
#define J0(n) "v_cmp_lt_u64 vcc, %" #n ", %" #n "\n"
#define J1(n) "v_cmp_lt_f64 vcc, %" #n ", %" #n "\n"
#define J2(n) "v_max_f64 %" #n ", %" #n ", %" #n "\n"
#define J3(n) "v_min_f64 %" #n ", %" #n ", %" #n "\n"
#define J4(n) "v_mul_f64 %" #n ", %" #n ", %" #n "\n"
#define J5(n) "v_lshl_add_u64 %" #n ", %" #n ", 3, %" #n "\n"
#define J6(n) "v_cmp_eq_u64 vcc, %" #n ", %" #n "\n"
        if constexpr (KIND == 0) { R4(J0) R4(J0) R4(J0) R4(J0) R4(J0) R4(J0) R4(J0) R4(J0) }
        if constexpr (KIND == 1) { D4(J1) D4(J1) D4(J1) D4(J1) D4(J1) D4(J1) D4(J1) D4(J1) }
        if constexpr (KIND == 2) { D4(J2) D4(J2) D4(J2) D4(J2) D4(J2) D4(J2) D4(J2) D4(J2) }
        if constexpr (KIND == 3) { D4(J3) D4(J3) D4(J3) D4(J3) D4(J3) D4(J3) D4(J3) D4(J3) }
        if constexpr (KIND == 4) { D4(J4) D4(J4) D4(J4) D4(J4) D4(J4) D4(J4) D4(J4) D4(J4) }
        if constexpr (KIND == 5) { R4(J5) R4(J5) R4(J5) R4(J5) R4(J5) R4(J5) R4(J5) R4(J5) }
        if constexpr (KIND == 6) { R4(J6) R4(J6) R4(J6) R4(J6) R4(J6) R4(J6) R4(J6) R4(J6) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned long long)(d0 + d1 + d2 + d3);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND>
void run64(const char *name) {
    unsigned long long *out; long long *cyc;
    CK(hipMalloc(&out, 1024 * 1024 * 8)); CK(hipMalloc(&cyc, 1024 * 16 * 8));
    const int reps = 2000;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k64<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k64<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipDeviceSynchronize());
        long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-14s waves/SIMD %d: %.3f ticks per instr per wave, %.3f per instr per SIMD\n", name, wps, mx / (reps * 32.0), mx / (reps * 32.0) / wps);
    }
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    run64<0>("v_cmp_lt_u64"); run64<1>("v_cmp_lt_f64"); run64<2>("v_max_f64"); run64<3>("v_min_f64"); run64<4>("v_mul_f64"); run64<5>("v_lshl_add_u64"); run64<6>("v_cmp_eq_u64");
    run<0>("v_and"); run<1>("v_mov_dpp"); run<2>("v_cndmask"); run<3>("v_cmp->sgpr"); run<4>("v_add_u32"); run<5>("v_lshl_add");
    run<6>("v_cvt_u32_f32"); run<7>("v_mbcnt_lo"); run<8>("and+dpp"); run<9>("cmp+cndmask(pair)"); run<10>("cndmask_e64 sgpr"); run<11>("v_alignbit"); run<12>("cvt_f32_ubyte0"); run<13>("v_and_sdwa"); run<14>("v_add_u32_dpp"); run<15>("v_cmp_ne->vcc"); run<16>("v_lshrrev"); run<17>("v_bfe_u32"); run<18>("v_or3"); run<19>("v_mbcnt_hi"); run<20>("cmpx+s_mov exec"); run<21>("v_addc vcc"); run<22>("v_cmp_neq_f32"); run<23>("v_lshl_or"); run<24>("v_xor lit"); run<25>("v_sub_co");
    return 0;
}
