// Micro-benchmark: (1) what FETCH_SIZE / WRITE_SIZE report for 4-, 8- and 16-byte-per-lane streaming (run under rocprofv3 --pmc);
// (2) how many GB/s ONE workgroup of 1024 threads (one CU) streams, and what G workgroups reach together, for 8 / 16 B per lane
// and 4 / 8 / 16 loads in flight.   usage: fetch_calib [MiB per test]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <typename T, int U>
__global__ __launch_bounds__(1024) void stream_read(const T* __restrict__ src, size_t n_per_block, double* out) {
  const T* p = src + (size_t)blockIdx.x * n_per_block;
  double acc = 0;
  for (size_t i = threadIdx.x; i + (size_t)(U - 1) * blockDim.x < n_per_block; i += (size_t)U * blockDim.x) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
    for (int u = 0; u < U; u++) { const double* d = (const double*)&v[u]; for (unsigned q = 0; q < sizeof(T) / 8; q++) acc += d[q]; }
  }
  if (acc == 1.2345e300) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void stream_read4(const float* __restrict__ src, size_t n_per_block, float* out) {
  const float* p = src + (size_t)blockIdx.x * n_per_block;
  float acc = 0;
  for (size_t i = threadIdx.x; i + 7 * (size_t)blockDim.x < n_per_block; i += 8 * (size_t)blockDim.x) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (acc == 1.2345e30f) out[blockIdx.x] = acc;
}
template <typename T>
__global__ __launch_bounds__(1024) void stream_write(T* __restrict__ dst, size_t n_per_block, T val) {
  T* p = dst + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) p[i] = val;
}
template <typename T>
__global__ __launch_bounds__(1024) void stream_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n_per_block) {
  const T* p = src + (size_t)blockIdx.x * n_per_block; T* q = dst + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i + 7 * (size_t)blockDim.x < n_per_block; i += 8 * (size_t)blockDim.x) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
    for (int u = 0; u < 8; u++) q[i + (size_t)u * blockDim.x] = v[u];
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t bytes = mib << 20;
  char *src, *dst; double* out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(src, 0, bytes)); CK(hipMemset(dst, 0, bytes));
  // (1) counter calibration: whole buffer, 2048 workgroups
  {
    const int G = 2048; const size_t per = bytes / G;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(stream_read4, dim3(G), dim3(1024), 0, 0, (const float*)src, per / 4, (float*)out); });
    printf("read  4 B/lane  %zu MiB  %.3f ms  %.2f TB/s\n", mib, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((stream_read<double, 8>), dim3(G), dim3(1024), 0, 0, (const double*)src, per / 8, out); });
    printf("read  8 B/lane  %zu MiB  %.3f ms  %.2f TB/s\n", mib, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((stream_read<double2, 8>), dim3(G), dim3(1024), 0, 0, (const double2*)src, per / 16, out); });
    printf("read 16 B/lane  %zu MiB  %.3f ms  %.2f TB/s\n", mib, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((stream_write<double>), dim3(G), dim3(1024), 0, 0, (double*)dst, per / 8, 1.0); });
    printf("write 8 B/lane  %zu MiB  %.3f ms  %.2f TB/s\n", mib, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((stream_write<double2>), dim3(G), dim3(1024), 0, 0, (double2*)dst, per / 16, double2{1.0, 2.0}); });
    printf("write 16 B/lane %zu MiB  %.3f ms  %.2f TB/s\n", mib, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((stream_copy<double>), dim3(G), dim3(1024), 0, 0, (const double*)src, (double*)dst, per / 8); });
    printf("copy  8 B/lane  %zu MiB  %.3f ms  %.2f TB/s (read + write)\n", mib, t, 2.0 * bytes / t / 1e9);
  }
  // (2) per-workgroup streaming rate: G workgroups, each its own 64 MiB
  for (int G : {1, 8, 64, 128, 188, 256, 512}) {
    const size_t per = (size_t)64 << 20;
    if ((size_t)G * per > bytes) break;
    float t8_4 = timeit([&] { hipLaunchKernelGGL((stream_read<double, 4>), dim3(G), dim3(1024), 0, 0, (const double*)src, per / 8, out); });
    float t8_8 = timeit([&] { hipLaunchKernelGGL((stream_read<double, 8>), dim3(G), dim3(1024), 0, 0, (const double*)src, per / 8, out); });
    float t8_16 = timeit([&] { hipLaunchKernelGGL((stream_read<double, 16>), dim3(G), dim3(1024), 0, 0, (const double*)src, per / 8, out); });
    float t16_8 = timeit([&] { hipLaunchKernelGGL((stream_read<double2, 8>), dim3(G), dim3(1024), 0, 0, (const double2*)src, per / 16, out); });
    float tc = timeit([&] { hipLaunchKernelGGL((stream_copy<double>), dim3(G), dim3(1024), 0, 0, (const double*)src, (double*)dst, per / 8); });
    printf("G=%3d workgroups x 64 MiB: GB/s per workgroup  8B x4 %.1f  8B x8 %.1f  8B x16 %.1f  16B x8 %.1f  copy(8B x8, r+w) %.1f   chip total (8B x8) %.2f TB/s\n",
           G, per / t8_4 / 1e6, per / t8_8 / 1e6, per / t8_16 / 1e6, per / t16_8 / 1e6, 2.0 * per / tc / 1e6, G * (double)per / t8_8 / 1e9);
  }
  return 0;
}
