// Micro-benchmark: LDS atomic add throughput on gfx950 (u64 / f64 / u32, random vs conflict-free,
// full vs partial lane occupancy).  Build: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ inline unsigned mix(unsigned x){x^=x>>16;x*=0x7feb352du;x^=x>>15;x*=0x846ca68bu;x^=x>>16;return x;}
template<typename T, int MODE, int ACTIVE>
__global__ __launch_bounds__(512) void k(T* out, int reps, int ncomp){
  extern __shared__ unsigned char sm[];
  T* acc=(T*)sm;
  for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) acc[i]=T(0);
  __syncthreads();
  const int lane=threadIdx.x&63;
  unsigned idx[16];
  for(int j=0;j<16;j++){
    if(MODE==0) idx[j]=mix(threadIdx.x*977u+j*131u+blockIdx.x*7919u)%ncomp;      // random
    else idx[j]=(threadIdx.x+j*512)%ncomp;                                         // conflict-free, coalesced
  }
  T v=T(threadIdx.x+1);
  if(lane<ACTIVE){
    for(int r=0;r<reps;r++){
#pragma unroll
      for(int j=0;j<16;j++) atomicAdd(&acc[idx[j]], v);
    }
  }
  __syncthreads();
  T s=T(0); for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<typename T,int MODE,int ACTIVE> void run(const char* name,int blocks_per_cu){
  int reps=2000, ncomp=5865; size_t lds=ncomp*sizeof(T);
  int blocks=256*blocks_per_cu; T* out; CK(hipMalloc(&out,blocks*512*sizeof(T)));
  auto kern=k<T,MODE,ACTIVE>;
  CK(hipFuncSetAttribute((const void*)kern,hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern,dim3(blocks),dim3(512),lds,0,out,10,ncomp); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); hipLaunchKernelGGL(kern,dim3(blocks),dim3(512),lds,0,out,reps,ncomp); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  double ops=(double)blocks*8*ACTIVE*16.0*reps; // lane-ops
  double per_cu_clk=ops/(ms*1e-3)/256/2.4e9;
  printf("%-34s blocks/CU=%d  %.3f ms  %.2f Tops/s  %.2f lane-atomics/clk/CU (@2.4GHz)\n",name,blocks_per_cu,ms,ops/(ms*1e-3)/1e12,per_cu_clk);
  CK(hipFree(out));
}
int main(){
  run<unsigned long long,0,64>("u64 random 64 lanes",2);
  run<unsigned long long,0,64>("u64 random 64 lanes",4);
  run<unsigned long long,0,44>("u64 random 44 lanes",2);
  run<unsigned long long,0,16>("u64 random 16 lanes",2);
  run<unsigned long long,1,64>("u64 conflict-free 64 lanes",2);
  run<double,0,64>("f64 random 64 lanes",2);
  run<double,1,64>("f64 conflict-free 64 lanes",2);
  run<unsigned,0,64>("u32 random 64 lanes",2);
  run<unsigned,1,64>("u32 conflict-free 64 lanes",2);
  run<float,0,64>("f32 random 64 lanes",2);
  return 0;
}
