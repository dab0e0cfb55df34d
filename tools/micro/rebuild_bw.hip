// Micro-benchmark: which ingredient of hclust_rnn_kernel's rebuild pass costs it the bandwidth a bare copy of the same shape has.
// W workgroups of 1024 threads, one task (2000 x 2048 doubles in, compacted rows out) each; a wave takes NR rows at a time, U loads per
// row and lane in flight; variants add the ingredients one at a time:
//   V0 bare copy   V1 + column map in LDS (every column kept)   V2 + 10 % of the columns dropped (compacted stores, the dropped entries parked in LDS)
//   V3 + the running (min, second min, arg min) per row   V4 + ~40 extra f64 instructions per entry pair (the Lance-Williams share)
// Build: hipcc --offload-arch=gfx950 -O3 rebuild_bw.hip -o rebuild_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef __attribute__((address_space(1))) const double *gcd;
typedef __attribute__((address_space(1))) double *gd;
template <int V, int NR, int U>
__global__ __launch_bounds__(1024) void rebuild(const double *src, double *dst, int rows, int ld, int passes, double *sink) {
  __shared__ uint16_t colmap[2048];
  __shared__ double stage[16][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // column map: V >= 2 drops every 10th column (flag 0x8000 | slot), the others keep their order
  for (int c = threadIdx.x; c < 2048; c += 1024) {
    if (V >= 2) { const int d = c / 10; colmap[c] = (c % 10 == 9) ? (uint16_t)(0x8000u | (d & 255)) : (uint16_t)(c - d); }
    else colmap[c] = (uint16_t)c;
  }
  __syncthreads();
  double acc = 0.0;
  for (int p = 0; p < passes; ++p) {
    gcd s = (gcd)src + (size_t)blockIdx.x * rows * ld;
    gd d = (gd)dst + (size_t)blockIdx.x * rows * ld;
    for (int r0 = wave * NR; r0 + NR <= rows; r0 += 16 * NR) {
      gcd sr[NR]; gd dr[NR];
      double mn[NR], sc[NR]; int ix[NR];
#pragma unroll
      for (int t = 0; t < NR; ++t) { sr[t] = s + (size_t)(r0 + t) * ld; dr[t] = d + (size_t)(r0 + t) * ld; mn[t] = 1e300; sc[t] = 1e300; ix[t] = 0; }
      for (int j = lane; j < 2000; j += 64 * U) {
        unsigned cm[U];
        double x[NR][U];
        if (V >= 1) {
#pragma unroll
          for (int u = 0; u < U; ++u) cm[u] = colmap[(j + 64 * u) & 2047];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int t = 0; t < NR; ++t) x[t][u] = sr[t][j + 64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (V == 0) {
#pragma unroll
            for (int t = 0; t < NR; ++t) dr[t][j + 64 * u] = x[t][u];
          } else if (cm[u] & 0x8000u) {
#pragma unroll
            for (int t = 0; t < NR; ++t) stage[wave][t][cm[u] & 255] = x[t][u];
          } else {
#pragma unroll
            for (int t = 0; t < NR; ++t) {
              double v = x[t][u];
              if (V >= 4) { double w = v; for (int q = 0; q < 10; ++q) w = w * 1.0000001 + 0.5 / (w + 2.0); v = w > 1e300 ? w : v; }
              dr[t][cm[u]] = v;
              if (V >= 3) { sc[t] = fmin(sc[t], fmax(mn[t], v)); if (v < mn[t]) { mn[t] = v; ix[t] = (int)cm[u]; } }
            }
          }
        }
      }
      if (V >= 3) {
#pragma unroll
        for (int t = 0; t < NR; ++t) acc += mn[t] + sc[t] + ix[t];
      }
    }
  }
  if (acc == 12345.678) sink[0] = acc + stage[wave][0][lane];
}
template <int V, int NR, int U> void run(int wgs, double *a, double *b, double *sink) {
  const int ld = 2048, rows = 2000, passes = 4;
  auto k = rebuild<V, NR, U>;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, a, b, rows, ld, 1, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, a, b, rows, ld, passes, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double wfrac = V >= 2 ? 0.9 : 1.0;
  const double bytes = (double)wgs * rows * 2000 * 8.0 * passes * (1.0 + wfrac);
  printf("V%d NR=%d U=%-2d wgs=%-4d %8.3f ms  %6.2f TB/s (read+write)  %6.1f GB/s per workgroup\n", V, NR, U, wgs, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
}
int main() {
  const size_t n = (size_t)256 * 2000 * 2048;
  double *a, *b, *sink; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  for (int wgs : {150, 188}) {
    run<0, 1, 8>(wgs, a, b, sink);
    run<0, 2, 8>(wgs, a, b, sink);
    run<0, 2, 4>(wgs, a, b, sink);
    run<1, 2, 8>(wgs, a, b, sink);
    run<2, 2, 8>(wgs, a, b, sink);
    run<3, 2, 8>(wgs, a, b, sink);
    run<4, 2, 8>(wgs, a, b, sink);
    run<3, 2, 4>(wgs, a, b, sink);
    run<3, 1, 8>(wgs, a, b, sink);
    run<3, 1, 16>(wgs, a, b, sink);
  }
  return 0;
}
