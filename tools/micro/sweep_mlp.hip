// Micro-benchmark: the agglomeration's row sweep (read an old row pair, per entry one LDS column-map read, a running (min, second min,
// arg min), a store to the mapped column of the new row) in two shapes:
//   A: 1024 threads, 16 loads in flight per lane, issued and then worked through (what hclust_rnn_kernel does: 128 VGPRs)
//   B:  512 threads, 32 loads per batch and the NEXT batch's loads issued before the current one is worked through (256 VGPRs)
// W workgroups, each on its own 2000 x 2048 matrix pair.  Build: hipcc --offload-arch=gfx950 -O3 sweep_mlp.hip -o sweep_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef __attribute__((address_space(1))) const double *gcd;
typedef __attribute__((address_space(1))) double *gd;
constexpr int N = 2000, LD = 2048;
__device__ __forceinline__ void upd(double &m, double &s, int &i, double v, int B) { s = fmin(s, fmax(m, v)); if (v < m) { m = v; i = B; } }
__device__ __forceinline__ double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
template <int U>
__global__ __launch_bounds__(1024) void sweep_a(const double *src, double *dst, int passes, double *sink) {
  __shared__ unsigned short colmap[LD];
  __shared__ double res[N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < LD; i += 1024) colmap[i] = (unsigned short)(i < N ? i : N - 1);
  __syncthreads();
  gcd s = (gcd)src + (size_t)blockIdx.x * N * LD;
  gd d = (gd)dst + (size_t)blockIdx.x * N * LD;
  for (int p = 0; p < passes; ++p)
    for (int a0 = wave * 2; a0 < N; a0 += 32) {
      gcd r0 = s + (size_t)a0 * LD, r1 = r0 + LD;
      gd w0 = d + (size_t)a0 * LD, w1 = w0 + LD;
      double m0 = 1e300, s0 = 1e300, m1 = 1e300, s1 = 1e300; int i0 = 0, i1 = 0;
      for (int j0 = lane; j0 + 64 * (U - 1) < N; j0 += 64 * U) {
        unsigned cm[U]; double x0[U], x1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cm[u] = colmap[j0 + 64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) { x0[u] = r0[j0 + 64 * u]; x1[u] = r1[j0 + 64 * u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int B = cm[u]; w0[B] = x0[u]; w1[B] = x1[u]; upd(m0, s0, i0, x0[u], B); upd(m1, s1, i1, x1[u], B); }
      }
      m0 = wave_min(m0); m1 = wave_min(m1);
      if (lane == 0) { res[a0] = m0 + s0 + i0; res[a0 + 1] = m1 + s1 + i1; }
    }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = res[0] + res[N - 1];
}
// B: batches of (row pair, 16 x 64 columns), flattened; the next batch's loads go out before the current batch is worked through
template <int U>
__global__ __launch_bounds__(512) void sweep_b(const double *src, double *dst, int passes, double *sink) {
  __shared__ unsigned short colmap[LD];
  __shared__ double res[N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < LD; i += 512) colmap[i] = (unsigned short)(i < N ? i : N - 1);
  __syncthreads();
  gcd s = (gcd)src + (size_t)blockIdx.x * N * LD;
  gd d = (gd)dst + (size_t)blockIdx.x * N * LD;
  constexpr int CB = (LD + 64 * U - 1) / (64 * U);          // column blocks per row (the last one clamped)
  const int npair = N / 2, nbatch = (npair / 8) * CB * passes;   // per wave: pairs wave, wave + 8, ...
  double xc0[U], xc1[U], xn0[U], xn1[U];
  auto fetch = [&](int b, double (&y0)[U], double (&y1)[U]) {
    const int pr = (b / CB) % (npair / 8), cb = b % CB;
    const int a0 = (pr * 8 + wave) * 2;
    gcd r0 = s + (size_t)a0 * LD, r1 = r0 + LD;
#pragma unroll
    for (int u = 0; u < U; ++u) { int j = cb * 64 * U + 64 * u + lane; j = j < N ? j : N - 1; y0[u] = r0[j]; y1[u] = r1[j]; }
  };
  fetch(0, xc0, xc1);
  double m0 = 1e300, s0 = 1e300, m1 = 1e300, s1 = 1e300; int i0 = 0, i1 = 0;
  for (int b = 0; b < nbatch; ++b) {
    if (b + 1 < nbatch) fetch(b + 1, xn0, xn1);
    const int pr = (b / CB) % (npair / 8), cb = b % CB;
    const int a0 = (pr * 8 + wave) * 2;
    gd w0 = d + (size_t)a0 * LD, w1 = w0 + LD;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = cb * 64 * U + 64 * u + lane;
      if (j < N) { const int B = colmap[j]; w0[B] = xc0[u]; w1[B] = xc1[u]; upd(m0, s0, i0, xc0[u], B); upd(m1, s1, i1, xc1[u], B); }
    }
    if (cb == CB - 1) {
      m0 = wave_min(m0); m1 = wave_min(m1);
      if (lane == 0) { res[a0] = m0 + s0 + i0; res[a0 + 1] = m1 + s1 + i1; }
      m0 = s0 = m1 = s1 = 1e300; i0 = i1 = 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { xc0[u] = xn0[u]; xc1[u] = xn1[u]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = res[0] + res[N - 1];
}
template <typename K> void run(const char *name, K k, int threads, int wgs, double *a, double *b, double *sink) {
  const int passes = 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, a, b, 1, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, a, b, passes, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * N * N * 8.0 * passes * 2;
  printf("%-28s wgs=%-4d %8.3f ms  %6.2f TB/s (read+write)  %6.1f GB/s per workgroup\n", name, wgs, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
}
int main() {
  const size_t n = (size_t)256 * N * LD;
  double *a, *b, *sink; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 8 * 256));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  for (int wgs : {25, 125, 188, 256}) {
    run("A 1024 thr, 16 in flight", sweep_a<8>, 1024, wgs, a, b, sink);
    run("B 512 thr, 16+16 pipelined", sweep_b<8>, 512, wgs, a, b, sink);
    run("B 512 thr, 32+32 pipelined", sweep_b<16>, 512, wgs, a, b, sink);
  }
  return 0;
}
