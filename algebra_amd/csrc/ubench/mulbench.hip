#include "lazy.cuh"
using namespace arkhip;
template<class P> __global__ void __launch_bounds__(256) k_lazy(u32* out, int iters){
  typedef FpLazy<P> L; L a,b; u32 tid=blockIdx.x*blockDim.x+threadIdx.x;
  for(int i=0;i<L::L;i++){a.l[i]=(tid*2654435761u*(i+1))&L::MASK; b.l[i]=((tid^77)*40503u*(i+3))&L::MASK;}
  for(int k=0;k<iters;k++) a=L::mul(a,b);
  u32 r=0; for(int i=0;i<L::L;i++) r^=a.l[i]; out[tid]=r;
}
template<class P> __global__ void __launch_bounds__(256) k_sat(u32* out, int iters){
  typedef Fp<P> F; F a,b; u32 tid=blockIdx.x*blockDim.x+threadIdx.x;
  for(int i=0;i<F::N;i++){a.l[i]=(tid*2654435761u*(i+1)); b.l[i]=((tid^77)*40503u*(i+3));} a.l[F::N-1]&=0xfffffff; b.l[F::N-1]&=0xfffffff;
  for(int k=0;k<iters;k++) a=F::mul(a,b);
  u32 r=0; for(int i=0;i<F::N;i++) r^=a.l[i]; out[tid]=r;
}
template __global__ void k_lazy<BLS12_381_FQ>(u32*,int);
template __global__ void k_sat<BLS12_381_FQ>(u32*,int);
#include <stdio.h>
int main(){
  u32* out; hipMalloc(&out, 256*8*256*4*4); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters=2000;
  for(int w: {1,2,4,8}){
    for(int v=0; v<2; v++){
      int b=256*w;
      if(v==0) hipLaunchKernelGGL((k_lazy<BLS12_381_FQ>), dim3(b), dim3(256),0,0,out,10); else hipLaunchKernelGGL((k_sat<BLS12_381_FQ>), dim3(b), dim3(256),0,0,out,10);
      hipDeviceSynchronize(); hipEventRecord(e0);
      if(v==0) hipLaunchKernelGGL((k_lazy<BLS12_381_FQ>), dim3(b), dim3(256),0,0,out,iters); else hipLaunchKernelGGL((k_sat<BLS12_381_FQ>), dim3(b), dim3(256),0,0,out,iters);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
      printf("%s blocks/CU=%d  %8.3f ms  %8.2f Gmul/s\n", v==0?"lazy28":"sat32 ", w, ms, (double)b*256*iters/(ms*1e-3)*1e-9);
    }
  }
  return 0;
}
