// Micro-benchmark: what the memory system gives a kernel shaped like the agglomeration -- W workgroups of 1024 threads, each streaming
// its OWN region (one task's matrices) row by row, one wave per 16 KB row, reading and writing in equal parts -- against plain
// read-only / write-only / copy streams over the whole chip.  Build: hipcc --offload-arch=gfx950 -O3 copy_bw.hip -o copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef __attribute__((address_space(1))) const double *gcd;
typedef __attribute__((address_space(1))) double *gd;
// MODE 0 read, 1 write, 2 copy.  Region of workgroup b: rows [b*rows, (b+1)*rows) of `ld` doubles; wave w takes rows w, w+16, ...
template <int MODE, int U>
__global__ __launch_bounds__(1024) void stream_rows(const double *src, double *dst, int rows, int ld, int passes, double *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  for (int p = 0; p < passes; ++p) {
    gcd s = (gcd)src + (size_t)blockIdx.x * rows * ld;
    gd d = (gd)dst + (size_t)blockIdx.x * rows * ld;
    for (int r = wave; r < rows; r += 16) {
      gcd sr = s + (size_t)r * ld; gd dr = d + (size_t)r * ld;
      for (int j = lane; j < ld; j += 64 * U) {
        double x[U];
        if (MODE != 1) {
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = sr[j + 64 * u];
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = (double)(j + u);
        }
        if (MODE != 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) dr[j + 64 * u] = x[u];
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) acc += x[u];
        }
      }
    }
  }
  if (MODE == 0 && acc == 12345.678) sink[0] = acc;
}
template <int MODE, int U> void run(const char *name, int wgs, double *a, double *b, double *sink) {
  const int ld = 2048, rows = 2000, passes = 4;       // 32.8 MB per workgroup and pass
  auto k = stream_rows<MODE, U>;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, a, b, rows, ld, 1, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, a, b, rows, ld, passes, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * rows * ld * 8.0 * passes * (MODE == 2 ? 2 : 1);
  printf("%-22s U=%-2d wgs=%-4d %8.3f ms  %7.2f TB/s (read+write)  %6.1f GB/s per workgroup\n", name, U, wgs, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
}
int main() {
  const size_t n = (size_t)512 * 2000 * 2048;         // 16.8 GB each
  double *a, *b, *sink; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  for (int wgs : {25, 64, 128, 188, 256, 512}) {
    run<0, 8>("read", wgs, a, b, sink);
    run<1, 8>("write", wgs, a, b, sink);
    run<2, 8>("copy", wgs, a, b, sink);
  }
  run<2, 4>("copy", 188, a, b, sink);
  run<2, 16>("copy", 188, a, b, sink);
  run<2, 16>("copy", 512, a, b, sink);
  return 0;
}
