// Developer micro-benchmark (GPU box): issue cost of the instruction kinds the rasteriser's inner loops are made of, per SIMD at
// 8 waves/SIMD (wave64, gfx950).  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate2.hip -o /tmp/valu_rate2 && /tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8(a, b, c, d, e, f, g, h) a "\n" b "\n" c "\n" d "\n" e "\n" f "\n" g "\n" h "\n"
#define OPS8F(fmt) I8(fmt("%0", "%1"), fmt("%1", "%2"), fmt("%2", "%3"), fmt("%3", "%4"), fmt("%4", "%5"), fmt("%5", "%6"), fmt("%6", "%7"), fmt("%7", "%0"))
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7;
#define FREGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define IREGS : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
  asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s20", "s21", "vcc");
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
#define F(d, s) "v_fma_f32 " d ", " d ", " s ", " d
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 1) {   // cndmask e32, vcc as mask, independent destinations
#define F(d, s) "v_cndmask_b32 " d ", " d ", " s ", vcc"
      asm volatile(OPS8F(F) IREGS : : "vcc");
#undef F
    } else if (KIND == 2) {   // cndmask e64, SGPR pair as mask
#define F(d, s) "v_cndmask_b32_e64 " d ", " d ", " s ", s[20:21]"
      asm volatile(OPS8F(F) IREGS : : "s20", "s21");
#undef F
    } else if (KIND == 3) {   // cndmask with an inline constant operand
#define F(d, s) "v_cndmask_b32_e64 " d ", 0, " s ", s[20:21]"
      asm volatile(OPS8F(F) IREGS : : "s20", "s21");
#undef F
    } else if (KIND == 4) {
#define F(d, s) "v_min_f32 " d ", " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 5) {
#define F(d, s) "v_med3_f32 " d ", " d ", " s ", 1.0"
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 6) {
#define F(d, s) "v_min3_f32 " d ", " d ", " s ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 7) {
#define F(d, s) "v_cvt_f32_i32 " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 8) {
#define F(d, s) "v_mad_u32_u24 " d ", " d ", " s ", " d
      asm volatile(OPS8F(F) IREGS);
#undef F
    } else if (KIND == 9) {
#define F(d, s) "v_fma_f32 " d ", " d ", " s ", " d " clamp"
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 10) {  // compare into an SGPR pair (VOP3 form)
#define F(d, s) "v_cmp_lt_f32_e64 s[20:21], " d ", " s
      asm volatile(OPS8F(F) : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21");
#undef F
    } else if (KIND == 11) {
#define F(d, s) "v_rcp_f32 " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 12) {
#define F(d, s) "v_log_f32 " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 13) {
#define F(d, s) "v_mul_lo_u32 " d ", " d ", " s
      asm volatile(OPS8F(F) IREGS);
#undef F
    } else if (KIND == 14) {
#define F(d, s) "v_bfi_b32 " d ", " d ", " s ", " d
      asm volatile(OPS8F(F) IREGS);
#undef F
    } else if (KIND == 15) {  // compare + dependent cndmask pairs (what a select compiles to)
      asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n"
                   "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                   FREGS : : "vcc");
    } else if (KIND == 16) {
#define F(d, s) "v_sub_f32 " d ", " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 17) {
#define F(d, s) "v_cvt_i32_f32 " d ", " s
      asm volatile(OPS8F(F) IREGS);
#undef F
    } else if (KIND == 18) {
#define F(d, s) "v_lshl_add_u32 " d ", " d ", 3, " s
      asm volatile(OPS8F(F) IREGS);
#undef F
    } else if (KIND == 19) {
#define F(d, s) "v_exp_f32 " d ", " s
      asm volatile(OPS8F(F) FREGS);
#undef F
    } else if (KIND == 20) {  // fma with an SGPR operand
#define F(d, s) "v_fma_f32 " d ", " d ", s20, " s
      asm volatile(OPS8F(F) FREGS : : "s20");
#undef F
    } else if (KIND == 21) {  // mul with negation / abs modifiers (VOP3)
#define F(d, s) "v_mul_f32_e64 " d ", -" d ", |" s "|"
      asm volatile(OPS8F(F) FREGS);
#undef F
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}
template <int KIND> void run(const char* name, float* out, int wavesPerSimd) {
  const int iters = 20000, blocks = 256 * wavesPerSimd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, 100);
  hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 8 * wavesPerSimd;
  printf("%-44s waves/SIMD %d: %.2f cycles @2.4GHz per wave-instruction per SIMD\n", name, wavesPerSimd, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w : {8, 2}) {
    run<0>("v_fma_f32", out, w); run<1>("v_cndmask_b32 (vcc)", out, w); run<2>("v_cndmask_b32_e64 (sgpr pair)", out, w);
    run<3>("v_cndmask_b32_e64 0, v, sgpr", out, w); run<4>("v_min_f32", out, w); run<5>("v_med3_f32", out, w); run<6>("v_min3_f32", out, w);
    run<7>("v_cvt_f32_i32", out, w); run<8>("v_mad_u32_u24", out, w); run<9>("v_fma_f32 clamp", out, w); run<10>("v_cmp_lt_f32_e64 -> sgpr", out, w);
    run<11>("v_rcp_f32", out, w); run<12>("v_log_f32", out, w); run<13>("v_mul_lo_u32", out, w); run<14>("v_bfi_b32", out, w);
    run<15>("v_cmp + dependent v_cndmask (pairs)", out, w); run<16>("v_sub_f32", out, w); run<17>("v_cvt_i32_f32", out, w);
    run<18>("v_lshl_add_u32", out, w); run<19>("v_exp_f32", out, w); run<20>("v_fma_f32 with an SGPR operand", out, w); run<21>("v_mul_f32_e64 neg/abs", out, w);
  }
  return 0;
}
