#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// wave 0 of every SIMD (waves 0..3 of the block) runs a dependent DPP chain; the other waves run a "noise" workload
// NOISE: 0 idle(exit), 1 v_add_f32 stream, 2 f64 fma stream, 3 rsq_f64 stream, 4 LDS read stream, 5 SALU stream, 6 dependent DPP chains too, 7 s_sleep polling loop w/ ds_read
template<int NOISE> __global__ __launch_bounds__(1024) void k(float*out,long long*cyc,int iters,int nw){
 __shared__ float lds[4096];
 const int wave=threadIdx.x>>6; float a=threadIdx.x*1e-3f,b=1.f,c=2.f,x=1e-4f; double d0=1.0+threadIdx.x,dx=1.0000001;
 lds[threadIdx.x]=a; __syncthreads();
 if(wave>=nw) return;
 long long t0=clock64();
 if(wave<4){
  for(int it=0;it<iters;++it){ asm volatile(REP64("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a) : "v"(x)); }
 } else {
  for(int it=0;it<iters;++it){
   if(NOISE==1){ asm volatile(REP64("v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_add_f32 %2, %2, %3\n") : "+v"(a),"+v"(b),"+v"(c) : "v"(x)); }
   if(NOISE==2){ asm volatile(REP64("v_fma_f64 %0, %0, %1, %0\n") : "+v"(d0) : "v"(dx)); }
   if(NOISE==3){ asm volatile(REP64("v_rsq_f64 %0, %0\n") : "+v"(d0)); }
   if(NOISE==4){ float t; asm volatile(REP64("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=v"(t) : "v"((int)(threadIdx.x*4)) : "memory"); a+=t; }
   if(NOISE==5){ asm volatile(REP64("s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 7\n") ::: "s20","s21","scc"); }
   if(NOISE==6){ asm volatile(REP64("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a) : "v"(x)); }
   if(NOISE==7){ float t; asm volatile(REP8("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s20, %0\n s_cmp_gt_i32 s20, 5\n s_sleep 1\n") : "=v"(t) : "v"(0) : "memory","s20","scc"); a+=t; }
  }
 }
 long long t1=clock64();
 out[blockIdx.x*1024+threadIdx.x]=a+b+c+(float)d0; if((threadIdx.x&63)==0)cyc[blockIdx.x*16+wave]=t1-t0;
}
int main(){float*out;long long*cyc;hipMalloc(&out,256*1024*4);hipMalloc(&cyc,256*16*8);
 const char*names[8]={"none","v_add_f32 x3 streams","v_fma_f64","v_rsq_f64","ds_read+wait","SALU","DPP chain","poll loop (sleep)"};
 int iters=200;
 for(int nw : {4,8,16}){ printf("--- %d waves per block (chain waves 0-3; %d noise waves per SIMD)\n",nw,(nw-4)/4);
 for(int m=0;m<8;++m){ if(nw==4 && m>0) break; hipMemset(cyc,0,256*16*8);
  for(int rep=0;rep<2;++rep){
  switch(m){case 0:hipLaunchKernelGGL(k<0>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 1:hipLaunchKernelGGL(k<1>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 2:hipLaunchKernelGGL(k<2>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 3:hipLaunchKernelGGL(k<3>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 4:hipLaunchKernelGGL(k<4>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 5:hipLaunchKernelGGL(k<5>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 6:hipLaunchKernelGGL(k<6>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;case 7:hipLaunchKernelGGL(k<7>,dim3(256),dim3(1024),0,0,out,cyc,iters,nw);break;}
  hipDeviceSynchronize();}
  std::vector<long long>c(256*16);hipMemcpy(c.data(),cyc,256*16*8,hipMemcpyDeviceToHost);
  printf("noise=%-22s chain wave: %.2f cycles per dependent DPP step   (noise wave total %.0f cycles)\n",names[m],(double)c[0]/iters/64,(double)c[4]);}}
 return 0;}
