/* Entry points that exist in LAB builds of the library only (make -C sharp_amd/csrc LAB=1 -> sharp_amd/variants/libsharp_hip_lab.so): hooks of
 * kernels that were measured and not adopted (LAB_NOTES.md).  The product library (include/sharp_hip.h) exports none of them. */
#ifndef SHARP_LAB_H
#define SHARP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif
/* D = 1 - U U^T (n x n row-major) of n unit rows U (n x p row-major) through the sliced-integer distance GEMM (tools/lab/gemm_i8.hip: seven
 * 7-bit digits per entry, exact int8 products on the matrix cores; R/get_opt_hclust.R:66-74 is what it serves). */
int sharp_dist_i8(const double *U, int n, int p, double *D);
/* `count` tasks of n x p unit rows; ms[0] rows -> digits, ms[1] digits -> D, ms[2] the fp64 MFMA kernel on the same tasks. */
int sharp_dist_i8_bench(int n, int p, int count, int reps, double *ms);
#ifdef __cplusplus
}
#endif
#endif
