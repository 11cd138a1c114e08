// Issue cost of v_exp_f32 on gfx950, alone and mixed (what bounds the D = 64 forward).  One wave per SIMD, cycles
// from s_memtime, independent destination registers (no dependent-issue stalls):
//   exp        N x v_exp_f32
//   fma        N x v_fma_f32
//   exp+fma    N x (v_exp_f32, v_fma_f32)           -- sum of the two => one issue pipe; max => the transcendental unit runs beside
//   exp+3fma   N x (v_exp_f32, 3 x v_fma_f32)
//   mfma       N x v_mfma_f32_32x32x16_bf16
//   mfma+2exp  N x (v_mfma, 2 x v_exp_f32)          -- the forward's ratio at D = 64: 16 exps per 8 MFMAs
//   mfma+2exp+3 N x (v_mfma, 2 x v_exp_f32, 3 x v_fma_f32)   -- plus its add / pack work
//   add, pk_add, exp+pk_add                          -- is v_pk_add_f32 (two row-sum adds in one instruction) full rate?
//   mfma+2exp+2add+cvt / mfma+2exp+pk_add+cvt        -- the forward's per-MFMA VALU today / with packed row sums
//   mfma+pk_add, mfma+add, mfma+pk_mul               -- does a packed-fp32 op run beside an MFMA in flight?
// (the `it` loop's taken branch costs ~4.5 cycles per slot in the rows without an MFMA: subtract it)
// Diagnostic only, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x;
  float x[8], y[8], z[8];
  for (int i = 0; i < 8; ++i) { x[i] = -in[(tid + 64 * i) & 1023]; y[i] = 0.f; z[i] = in[(tid + 7 * i) & 1023]; }
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 xx[4], zz[4];
  unsigned pk[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { xx[i] = f32x2{x[i], x[i + 4]}; zz[i] = f32x2{0.f, 0.f}; }
  u32x4 a = {__float_as_uint(in[tid & 1023]), __float_as_uint(in[(tid + 1) & 1023]), 0x3f803f80u, 0x3f803f80u};
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if constexpr (MODE >= 4 && MODE != 7 && MODE != 8 && MODE != 9) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(c[s & 3]) : "v"(a));
      if constexpr (MODE == 0 || MODE == 2 || MODE == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(y[s]) : "v"(x[s]));
      if constexpr (MODE == 1 || MODE == 2 || MODE == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[s]) : "v"(x[s]));
      if constexpr (MODE == 3) {
        asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[(s + 3) & 7]) : "v"(x[s]));
        asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[(s + 5) & 7]) : "v"(x[s]));
      }
      if constexpr (MODE == 5 || MODE == 6 || MODE >= 10) {
        asm volatile("v_exp_f32 %0, %1" : "=v"(y[s]) : "v"(x[s]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y[(s + 4) & 7]) : "v"(x[(s + 4) & 7]));
      }
      if constexpr (MODE == 7) asm volatile("v_add_f32 %0, %1, %0" : "+v"(z[s]) : "v"(x[s]));
      if constexpr (MODE == 8 || MODE == 9) {
        if constexpr (MODE == 9) asm volatile("v_exp_f32 %0, %1" : "=v"(y[s]) : "v"(x[s]));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(zz[s & 3]) : "v"(xx[s & 3]));
      }
      if constexpr (MODE == 10) {   // the forward's VALU per MFMA today: 2 exps, 2 row-sum adds, 1 pack
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(z[s]) : "v"(y[(s + 2) & 7]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(z[(s + 4) & 7]) : "v"(y[(s + 6) & 7]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[s & 3]) : "v"(y[(s + 2) & 7]), "v"(y[(s + 6) & 7]));
      }
      if constexpr (MODE == 11) {   // with the two adds as one v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(zz[s & 3]) : "v"(xx[(s + 1) & 3]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[s & 3]) : "v"(y[(s + 2) & 7]), "v"(y[(s + 6) & 7]));
      }
      if constexpr (MODE == 12) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(zz[s & 3]) : "v"(xx[(s + 1) & 3]));
      if constexpr (MODE == 13) asm volatile("v_add_f32 %0, %1, %0" : "+v"(z[s]) : "v"(x[s]));
      if constexpr (MODE == 14) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(zz[s & 3]) : "v"(xx[(s + 1) & 3]));
      if constexpr (MODE == 6) {
        asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[s]) : "v"(x[s]));
        asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[(s + 3) & 7]) : "v"(x[s]));
        asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(z[(s + 5) & 7]) : "v"(x[s]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += y[i] + z[i];
  for (int i = 0; i < 4; ++i) acc += zz[i][0] + zz[i][1] + __uint_as_float(pk[i]);
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc += c[j][i];
  out[blockIdx.x * 256 + tid] = acc;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, const float* in, float* out, unsigned long long* clk) {
  const int iters = 4096, blocks = 256;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("%-12s %6.2f cycles per group of %d instruction(s)\n", name, s / blocks / (iters * 8.0), per_iter);
}

int main() {
  float *in, *out;
  unsigned long long* clk;
  hipMalloc(&in, 4096);
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&clk, 256 * 8);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 32768.f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  run<0>("exp", 1, in, out, clk);
  run<1>("fma", 1, in, out, clk);
  run<2>("exp+fma", 2, in, out, clk);
  run<3>("exp+3fma", 4, in, out, clk);
  run<4>("mfma", 1, in, out, clk);
  run<5>("mfma+2exp", 3, in, out, clk);
  run<6>("mfma+2exp+3", 6, in, out, clk);
  run<7>("add", 1, in, out, clk);
  run<8>("pk_add", 1, in, out, clk);
  run<9>("exp+pk_add", 2, in, out, clk);
  run<10>("mfma+2exp+2add+cvt", 6, in, out, clk);
  run<11>("mfma+2exp+pk_add+cvt", 5, in, out, clk);
  run<12>("mfma+pk_add", 2, in, out, clk);
  run<13>("mfma+add", 2, in, out, clk);
  run<14>("mfma+pk_mul", 2, in, out, clk);
  return 0;
}
