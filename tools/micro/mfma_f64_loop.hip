// Micro-benchmark: the MFMA skeleton of gemm_tn_f64_fast_kernel (512 threads, 8 accumulator tiles per wave, 32 MFMAs per
// k-step) with and without the two workgroup barriers per k-step, and with a share of blocks that exit immediately.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int BAR, int LDSR>
__global__ __launch_bounds__(512) void k(double* out, int ksteps, double seed, int dead_mod){
  __shared__ double As[16][144]; __shared__ double Bs[16][144];
  if(dead_mod && (blockIdx.x % dead_mod) >= dead_mod/2 + 1) return;
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6; const int wr=(wave>>2)*64, wc=(wave&3)*32;
  for(int i=threadIdx.x;i<16*144;i+=512){ As[0][i]=seed+i*1e-6; Bs[0][i]=seed*0.5+i*2e-6; }
  __syncthreads();
  v4f64 acc[4][2]; for(int i=0;i<4;i++) for(int j=0;j<2;j++) acc[i][j]=(v4f64){0,0,0,0};
  double a[4], b[2];
  for(int i=0;i<4;i++) a[i]=As[lane>>4][wr+i*16+(lane&15)];
  for(int j=0;j<2;j++) b[j]=Bs[lane>>4][wc+j*16+(lane&15)];
  for(int it=0;it<ksteps;it++){
    if(BAR){ __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
#pragma unroll
    for(int kk=0;kk<16;kk+=4){
      if(LDSR){ const int kr=kk+(lane>>4);
#pragma unroll
        for(int i=0;i<4;i++) a[i]=As[kr][wr+i*16+(lane&15)];
#pragma unroll
        for(int j=0;j<2;j++) b[j]=Bs[kr][wc+j*16+(lane&15)]; }
#pragma unroll
      for(int i=0;i<4;i++)
#pragma unroll
        for(int j=0;j<2;j++) acc[i][j]=__builtin_amdgcn_mfma_f64_16x16x4f64(a[i],b[j],acc[i][j],0,0,0);
    }
    if(BAR){ __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
  }
  double s=0; for(int i=0;i<4;i++) for(int j=0;j<2;j++) for(int r=0;r<4;r++) s+=acc[i][j][r];
  out[blockIdx.x*512+threadIdx.x]=s;
}
template<int BAR,int LDSR> void run(const char* name, double* out, int blocks, int ksteps, int dead_mod){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<BAR,LDSR>),dim3(blocks),dim3(512),0,0,out,2,1.0,dead_mod); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); hipLaunchKernelGGL((k<BAR,LDSR>),dim3(blocks),dim3(512),0,0,out,ksteps,1.0,dead_mod); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  double live = dead_mod ? (double)blocks*(dead_mod/2+1)/dead_mod : blocks;
  double flops=live*8*ksteps*32.0*2048.0;
  printf("%-34s blocks=%d ksteps=%d: %.2f ms, %.1f TFLOP/s\n",name,blocks,ksteps,ms,flops/(ms*1e-3)/1e12);
}
int main(){
  double* out; CK(hipMalloc(&out,(size_t)100000*512*8));
  run<0,0>("no barrier, no LDS reads",out,51000,25,0);
  run<1,0>("2 barriers/kstep, no LDS reads",out,51000,25,0);
  run<1,1>("2 barriers/kstep, LDS reads",out,51000,25,0);
  run<0,1>("no barrier, LDS reads",out,51000,25,0);
  run<1,1>("same, 47% dead blocks interleaved",out,96000,25,16);
  run<1,1>("long blocks (250 ksteps)",out,5100,250,0);
  run<0,0>("long blocks no barrier no LDS",out,5100,250,0);
  return 0;
}
