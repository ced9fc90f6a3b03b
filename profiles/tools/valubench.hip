#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// throughput of independent VALU ops: 8 independent accumulators per wave, N waves per SIMD
template<int MODE> __global__ __launch_bounds__(1024) void k(float*out,long long*cyc,int iters){
 float a0=threadIdx.x*1e-3f,a1=1.0f,a2=2.0f,a3=3.f,a4=4.f,a5=5.f,a6=6.f,a7=7.f,x=1e-4f;
 double d0=1.0+threadIdx.x,d1=2.0,d2=3.0,d3=4.0,dx=1.0000001;
 long long t0=clock64();
 for(int it=0;it<iters;++it){
  if(MODE==0){ asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n") : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x)); }
  if(MODE==1){ asm volatile(REP8("v_add_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x)); }
  if(MODE==2){ asm volatile(REP8("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n") : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3) : "v"(dx)); }
  if(MODE==3){ asm volatile(REP8("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n") : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3)); }
  if(MODE==4){ asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3) : "v"(dx)); }
  if(MODE==5){ asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n") : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)); }
  if(MODE==6){ asm volatile(REP8("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4\n") : "+v"(d0),"+v"(d1),"+v"(d2),"+v"(d3) : "v"(x)); }
  if(MODE==7){ asm volatile(REP8("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s22, %6\n v_readfirstlane_b32 s23, %7\n") :: "v"(a0),"v"(a1),"v"(a2),"v"(a3),"v"(a4),"v"(a5),"v"(a6),"v"(a7) : "s20","s21","s22","s23"); }
 }
 long long t1=clock64();
 out[blockIdx.x*256+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7+(float)(d0+d1+d2+d3); if(threadIdx.x==0)cyc[blockIdx.x]=t1-t0;
}
int main(){float*out;long long*cyc;hipMalloc(&out,4096*256*4);hipMalloc(&cyc,4096*8);
 const char*names[8]={"v_add_f32","v_add_f32_dpp wave_shr","v_fma_f64","v_rsq_f64","v_pk_add_f32","v_rcp_f32","v_cvt_f64_f32","v_readfirstlane"};
 for(int wps : {1,2,4}){ printf("--- %d waves per SIMD (block = %d threads, 256 blocks)\n",wps,wps*256);
 for(int m=0;m<8;++m){int iters=300;hipEvent_t a,b;hipEventCreate(&a);hipEventCreate(&b);
  for(int rep=0;rep<2;++rep){hipEventRecord(a);
  dim3 g(256),bl(256*wps);
  switch(m){case 0:hipLaunchKernelGGL(k<0>,g,bl,0,0,out,cyc,iters);break;case 1:hipLaunchKernelGGL(k<1>,g,bl,0,0,out,cyc,iters);break;case 2:hipLaunchKernelGGL(k<2>,g,bl,0,0,out,cyc,iters);break;case 3:hipLaunchKernelGGL(k<3>,g,bl,0,0,out,cyc,iters);break;case 4:hipLaunchKernelGGL(k<4>,g,bl,0,0,out,cyc,iters);break;case 5:hipLaunchKernelGGL(k<5>,g,bl,0,0,out,cyc,iters);break;case 6:hipLaunchKernelGGL(k<6>,g,bl,0,0,out,cyc,iters);break;case 7:hipLaunchKernelGGL(k<7>,g,bl,0,0,out,cyc,iters);break;}
  hipEventRecord(b);hipEventSynchronize(b);}
  std::vector<long long>c(256);hipMemcpy(c.data(),cyc,256*8,hipMemcpyDeviceToHost);
  // cycles per instruction per SIMD = wave cycles / (instr per wave) / ... each wave issues 64*iters instrs; wps waves share a SIMD
  printf("%-26s wave-cycles/instr %.2f  -> SIMD cycles/instr %.2f\n",names[m],(double)c[0]/iters/64,(double)c[0]/iters/64/wps);}}
 return 0;}
