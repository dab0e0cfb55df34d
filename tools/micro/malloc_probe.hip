// Micro-benchmark: cost of hipMalloc and of the first touch (memset) of successive 4 GiB allocations, up to 96 GiB in use.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} }while(0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t chunk = (size_t)4 << 30;
  std::vector<void*> ps;
  CK(hipFree(0));
  for (int i = 0; i < 24; i++) {
    void* p; double t0 = now();
    CK(hipMalloc(&p, chunk)); double t1 = now();
    CK(hipMemset(p, 0, chunk)); CK(hipDeviceSynchronize()); double t2 = now();
    CK(hipMemset(p, 1, chunk)); CK(hipDeviceSynchronize()); double t3 = now();
    printf("chunk %2d (%3zu GiB in use): hipMalloc %.1f ms, first memset %.1f ms, second memset %.1f ms\n", i, (ps.size() + 1) * 4, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    ps.push_back(p);
  }
  double t0 = now();
  for (void* p : ps) CK(hipFree(p));
  printf("hipFree of all: %.1f ms\n", (now() - t0) * 1e3);
  void* p; t0 = now(); CK(hipMalloc(&p, (size_t)24 << 30)); printf("one hipMalloc of 24 GiB after the frees: %.1f ms\n", (now() - t0) * 1e3);
  return 0;
}
