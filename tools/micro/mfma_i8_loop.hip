// Micro-benchmark for the sliced-integer form of the distance GEMM: (a) the A/B lane maps of v_mfma_i32_32x32x32_i8 checked
// with exact integer data, (b) the sustained rate of the 28 slice products of one k step (7 slices, levels s + t <= 6) with
// operands in registers and with operands read from LDS, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// one wave: C = A (32 x 32 int8, row-major [i][k]) * B^T (B as [j][k]); hypothesis: lane l holds A[l & 31][16 (l >> 5) + 0..15]
__global__ void layout_kernel(const int8_t *A, const int8_t *B, int *C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v4i a = *reinterpret_cast<const v4i *>(A + r * 32 + 16 * h);
  v4i b = *reinterpret_cast<const v4i *>(B + r * 32 + 16 * h);
  v16i c; for (int q = 0; q < 16; ++q) c[q] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int q = 0; q < 16; ++q) { const int row = (q & 3) + 8 * (q >> 2) + 4 * h, col = r; C[row * 32 + col] = c[q]; }
}

template <int NS, int LDSR>
__global__ __launch_bounds__(512) void rate_kernel(int *out, int ksteps) {
  __shared__ v4i sm[2 * NS * 2 * 64 * 2];        // per wave pair ... (only the access pattern matters: 16 B per lane per fragment)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * NS * 2 * 64 * 2; i += 512) sm[i] = (v4i){(int)(i * 2654435761u), (int)(i * 40503u), i, ~i};
  __syncthreads();
  v4i a[NS], b[NS];
  for (int s = 0; s < NS; ++s) { a[s] = sm[(s * 2) * 64 + lane]; b[s] = sm[(s * 2 + 1) * 64 + lane]; }
  v16i acc[NS];
  for (int s = 0; s < NS; ++s) for (int q = 0; q < 16; ++q) acc[s][q] = 0;
  for (int it = 0; it < ksteps; ++it) {
    if (LDSR) {
      const int base = ((it & 1) * NS * 2) * 64 + (wave & 1) * 0;
#pragma unroll
      for (int s = 0; s < NS; ++s) { a[s] = sm[base + (s * 2) * 64 + lane]; b[s] = sm[base + (s * 2 + 1) * 64 + lane]; }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t + s < NS; ++t) acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[t], acc[s + t], 0, 0, 0);
  }
  int r = 0; for (int s = 0; s < NS; ++s) for (int q = 0; q < 16; ++q) r += acc[s][q];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int NS, int LDSR> void run(const char *name, int *out, int blocks, int ksteps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((rate_kernel<NS, LDSR>), dim3(blocks), dim3(512), 0, 0, out, 2); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); hipLaunchKernelGGL((rate_kernel<NS, LDSR>), dim3(blocks), dim3(512), 0, 0, out, ksteps); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double prods = NS * (NS + 1) / 2.0;
  const double ops = (double)blocks * 8 * ksteps * prods * 65536.0;
  printf("%-40s blocks=%d ksteps=%d: %.2f ms, %.2f POP/s (%.0f products per k step)\n", name, blocks, ksteps, ms, ops / (ms * 1e-3) / 1e15, prods);
}
int main() {
  std::vector<int8_t> A(32 * 32), B(32 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) { A[i * 32 + k] = (int8_t)((i * 7 + k * 3) % 23 - 11); B[i * 32 + k] = (int8_t)((i * 5 + k * 11 + 1) % 19 - 9); }
  int8_t *dA, *dB; int *dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
  CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
  std::vector<int> C(1024); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[j * 32 + k]; bad += s != C[i * 32 + j]; }
  printf("layout check (lane l: row l&31, k = 16 (l>>5) + j; C row = (q&3) + 8 (q>>2) + 4 (l>>5), col = l&31): %d of 1024 wrong\n", bad);
  int *out; CK(hipMalloc(&out, (size_t)20000 * 512 * 4));
  run<7, 0>("7 slices, operands in registers", out, 5120, 200);
  run<7, 1>("7 slices, operands from LDS", out, 5120, 200);
  run<8, 0>("8 slices, operands in registers", out, 5120, 200);
  run<8, 1>("8 slices, operands from LDS", out, 5120, 200);
  run<7, 1>("7 slices, LDS, short blocks", out, 20000, 15);
  return 0;
}
