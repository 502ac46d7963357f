// Developer micro-benchmark (GPU box): issue rate of plain wave64 VALU instructions on gfx950, per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                   "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == 1) {  // v_add_u32
      asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n"
                   "v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n"
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    } else if (KIND == 2) {  // v_cndmask
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                   "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : : "vcc");
    } else if (KIND == 3) {  // v_cmp + nothing else
      asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                   "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n"
                   : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
    } else if (KIND == 4) {  // v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                   "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
    } else if (KIND == 5) {  // transcendental
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == 6) {  // fma + exec-mask churn: v_cmp -> s_and_saveexec -> s_or exec (a divergent-branch skeleton)
      asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %2, %2, %2, %2\n s_or_b64 exec, exec, s[20:21]\n"
                   "v_cmp_lt_f32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %3, %3, %3, %3\n s_or_b64 exec, exec, s[20:21]\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc", "s20", "s21");
    } else if (KIND == 7) {  // v_mad_u64_u32 / 64-bit address maths
      asm volatile("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %0\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %2\n"
                   "v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %0\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %2\n"
                   : "+v"(*(long long*)&b0), "+v"(*(long long*)&b2), "+v"(*(long long*)&b4), "+v"(*(long long*)&b6));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}
template <int KIND> void run(const char* name, float* out, int wavesPerSimd) {
  const int iters = 20000, blocks = 256 * wavesPerSimd;   // 256-thread block = 1 wave per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, 100);
  hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 8 * wavesPerSimd;
  printf("%-34s waves/SIMD %d: %.3f ms -> %.2f cycles @2.4GHz per wave-instruction per SIMD\n", name, wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", out, w); run<1>("v_add_u32", out, w); run<2>("v_cndmask_b32", out, w); run<3>("v_cmp_lt_f32", out, w);
    run<4>("v_pk_fma_f32", out, w); run<5>("v_exp_f32", out, w); run<6>("cmp+saveexec+fma+restore (x2, as 8)", out, w); run<7>("v_lshl_add_u64", out, w);
  }
  return 0;
}
