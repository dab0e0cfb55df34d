// Micro-benchmark: do L2 row-list gathers (8 B/lane, 128-B segments at random) and LDS ds_add_u64 overlap when
// issued by DIFFERENT waves of the same CU?  Modes: gather-only, atomic-only, both (half the waves each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ inline unsigned mix(unsigned x){x^=x>>16;x*=0x7feb352du;x^=x>>15;x*=0x846ca68bu;x^=x>>16;return x;}
// mode bit0: gather waves active, bit1: atomic waves active. wsplit: number of gather waves out of 8
__global__ __launch_bounds__(512,4) void k(const uint2* __restrict__ tab, int nseg, unsigned long long* out, int reps, int ncomp, int mode, int wsplit){
  extern __shared__ unsigned char sm[];
  unsigned long long* acc=(unsigned long long*)sm;
  for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) acc[i]=0; __syncthreads();
  const int wave=threadIdx.x>>6, lane=threadIdx.x&63;
  unsigned long long x=0;
  if(wave<wsplit){
    if(mode&1){
      unsigned h=mix(blockIdx.x*977u+wave*131u+1u);
      for(int r=0;r<reps;r++){
        uint2 c[8];
#pragma unroll
        for(int u=0;u<8;u++){ h=mix(h+u+(lane>>4)*7919u*(u+1)); unsigned seg=(mix(h ^ ((lane>>4)*0x9e3779b9u)))%nseg; c[u]=tab[(size_t)seg*16+(lane&15)]; }
#pragma unroll
        for(int u=0;u<8;u++) x^=((unsigned long long)c[u].x<<32)|c[u].y;
      }
    }
  } else {
    if(mode&2){
      unsigned idx[16];
      for(int j=0;j<16;j++) idx[j]=mix(threadIdx.x*977u+j*131u+blockIdx.x*7919u)%ncomp;
      for(int r=0;r<reps;r++){
#pragma unroll
        for(int j=0;j<16;j++) if(lane<44) atomicAdd(&acc[idx[j]], (unsigned long long)(lane+1));
      }
    }
  }
  __syncthreads();
  unsigned long long s=x; for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

// mode 4: every wave software-pipelines its own gathers (next batch) behind its atomics (current batch), like rp_apply_kernel
template<int U>
__global__ __launch_bounds__(512,4) void k2(const uint2* __restrict__ tab, int nseg, unsigned long long* out, int reps, int ncomp, int what){
  extern __shared__ unsigned char sm[];
  unsigned long long* acc=(unsigned long long*)sm;
  for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) acc[i]=0; __syncthreads();
  const int wave=threadIdx.x>>6, lane=threadIdx.x&63;
  unsigned h=mix(blockIdx.x*977u+wave*131u+1u);
  uint2 c[U], cn[U];
  auto gather=[&](uint2* d){
#pragma unroll
    for(int u=0;u<U;u++){ h=mix(h+u+(lane>>4)*7919u*(u+1)); unsigned seg=(mix(h ^ ((lane>>4)*0x9e3779b9u)))%nseg; d[u]=tab[(size_t)seg*16+(lane&15)]; }
  };
  gather(c);
  unsigned long long x=0;
  for(int r=0;r<reps;r++){
    if(what&1) gather(cn);
    if(what&2){
#pragma unroll
      for(int u=0;u<U;u++){
        const unsigned w[4]={c[u].x&0xfffu,(c[u].x>>12)&0xfffu,(c[u].x>>20)&0xfffu,c[u].y&0xfffu};
#pragma unroll
        for(int q=0;q<4;q++) if(lane<44) atomicAdd(&acc[w[q]], (unsigned long long)(lane+1));
      }
    } else {
#pragma unroll
      for(int u=0;u<U;u++) x^=((unsigned long long)c[u].x<<32)|c[u].y;
    }
    if(what&1){
#pragma unroll
      for(int u=0;u<U;u++) c[u]=cn[u];
    }
  }
  __syncthreads();
  unsigned long long s=x; for(int i=threadIdx.x;i<ncomp;i+=blockDim.x) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){
  int nseg=20000, ncomp=5865; size_t lds=ncomp*8;
  std::vector<uint2> h((size_t)nseg*16); for(size_t i=0;i<h.size();i++){h[i].x=(unsigned)i*2654435761u; h[i].y=(unsigned)i;}
  uint2* tab; CK(hipMalloc(&tab,h.size()*8)); CK(hipMemcpy(tab,h.data(),h.size()*8,hipMemcpyHostToDevice));
  int blocks=512; unsigned long long* out; CK(hipMalloc(&out,blocks*512*8));
  CK(hipFuncSetAttribute((const void*)k,hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int wsplit: {4,2,6}) for(int mode: {1,2,3}){
    int reps=400;
    hipLaunchKernelGGL(k,dim3(blocks),dim3(512),lds,0,tab,nseg,out,10,ncomp,mode,wsplit); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k,dim3(blocks),dim3(512),lds,0,tab,nseg,out,reps,ncomp,mode,wsplit); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,a,b));
    double gbytes=(mode&1)? (double)blocks*wsplit*4.0*8*reps*128 : 0;   // 4 groups x 8 segs x 128 B per wave-rep
    double atoms=(mode&2)? (double)blocks*(8-wsplit)*44.0*16*reps : 0;
    printf("gather_waves=%d mode=%d  %.3f ms  gather %.2f TB/s  atomics %.2f T/s (%.2f /clk/CU)\n",wsplit,mode,ms,gbytes/(ms*1e-3)/1e12,atoms/(ms*1e-3)/1e12,atoms/(ms*1e-3)/256/2.4e9);
  }
  for(int what: {1,2,3}){
    int reps=100; const int U=16;
    CK(hipFuncSetAttribute((const void*)k2<U>,hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds));
    hipLaunchKernelGGL(k2<U>,dim3(blocks),dim3(512),lds,0,tab,nseg,out,10,ncomp,what); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k2<U>,dim3(blocks),dim3(512),lds,0,tab,nseg,out,reps,ncomp,what); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,a,b));
    double gbytes=(what&1)? (double)blocks*8*4.0*U*reps*128 : 0;
    double atoms=(what&2)? (double)blocks*8*44.0*U*4*reps : 0;
    printf("same-wave pipelined what=%d  %.3f ms  gather %.2f TB/s  atomics %.2f T/s (%.2f /clk/CU)\n",what,ms,gbytes/(ms*1e-3)/1e12,atoms/(ms*1e-3)/1e12,atoms/(ms*1e-3)/256/2.4e9);
  }
  return 0;
}
