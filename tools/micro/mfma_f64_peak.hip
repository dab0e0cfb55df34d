// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (register operands, 4 independent accumulators per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed){
  v4f64 acc[4]; for(int i=0;i<4;i++) acc[i]=(v4f64){0,0,0,0};
  double a=seed+threadIdx.x*1e-3, b=seed*0.5+threadIdx.x*2e-3;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int r=0;r<8;r++){
#pragma unroll
      for(int i=0;i<4;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
    }
  }
  double s=0; for(int i=0;i<4;i++) for(int j=0;j<4;j++) s+=acc[i][j];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){
  double* out; int blocks=256*4; CK(hipMalloc(&out,blocks*256*8));
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int rep=0;rep<3;rep++){
    int iters=20000;
    hipLaunchKernelGGL(k,dim3(blocks),dim3(256),0,0,out,100,1.0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k,dim3(blocks),dim3(256),0,0,out,iters,1.0); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,a,b));
    double flops=(double)blocks*4*iters*32.0*2048.0;
    printf("f64 MFMA 16x16x4: %.1f ms, %.1f TFLOP/s\n",ms,flops/(ms*1e-3)/1e12);
  }
  return 0;
}
