// valu_probe.hip -- issue cost (shader cycles per wave-instruction) of the VALU instructions the attention softmax is made
// of, one wave per SIMD and four: v_exp_f32, v_rcp_f32, v_fma_f32, v_pk_fma_f32, v_cvt_pk_bf16_f32, v_max3_f32, v_ldexp_f32.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long long* clk, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    // 8 independent chains x 16 = 128 instructions per iteration
    if (OP == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 1) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 3) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 4) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 5) { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %0\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
      a0 = p0[0] + p1[1]; a1 = p2[0] + p3[1]; }
    if (OP == 6) { REP16(asm volatile("v_ldexp_f32 %0, %0, 1\n v_ldexp_f32 %1, %1, 1\n v_ldexp_f32 %2, %2, 1\n v_ldexp_f32 %3, %3, 1\n v_ldexp_f32 %4, %4, 1\n v_ldexp_f32 %5, %5, 1\n v_ldexp_f32 %6, %6, 1\n v_ldexp_f32 %7, %7, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
  }
  const long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
template <int OP>
static void run(const char* name, float* out, long long* clk) {
  for (int waves : {4, 16}) {  // per CU: 1 and 4 waves per SIMD
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, out, clk, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), clk, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double c = 0;
    for (auto x : h) c += x;
    c /= 256;
    const double per_wave_instr = c / (iters * 128.0);          // cycles per instruction as seen by one wave
    const double per_simd = per_wave_instr / (waves / 4.0);      // SIMD cycles per wave-instruction at this occupancy
    printf("%-20s %2d waves/CU: %6.2f cycles per instruction per wave, %6.2f SIMD cycles per wave-instruction\n", name, waves, per_wave_instr, per_simd);
  }
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&clk, 256 * sizeof(long long));
  run<0>("v_exp_f32", out, clk); run<1>("v_rcp_f32", out, clk); run<2>("v_fma_f32", out, clk); run<3>("v_max3_f32", out, clk);
  run<4>("v_cvt_pk_bf16_f32", out, clk); run<5>("v_pk_fma_f32", out, clk); run<6>("v_ldexp_f32", out, clk);
  return 0;
}
