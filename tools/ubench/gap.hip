// How much VALU issues "for free" in the shadow of one v_mfma_f32_32x32x16_bf16?  Exact instruction
// streams (inline asm, nothing for the compiler to reorder): per gap 1 MFMA + NF v_fma_f32 + NE v_exp_f32,
// 1 / 2 / 3 waves per SIMD.  Prints cycles per gap per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define REP8(x) x x x x x x x x
#define MFMA "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
#define FMA(i) "v_fma_f32 %1, %1, %4, %5\n\t"

template <int NF, int NE, bool INDEP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc = {0}, acc2 = {0};
  f32x4 fa = {1e-3f * threadIdx.x, 2e-3f, 3e-3f, 4e-3f}, fb = {1e-3f, 2e-3f, 3e-3f, 5e-3f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 1e-4f * threadIdx.x;
  const float c = 0.999f, d = 1e-4f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (INDEP && (g & 1))
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(fa), "v"(fb));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c), "v"(d));
#pragma unroll
      for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(NF + i) & 7]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  __shared__ unsigned long long tmax;
  if (threadIdx.x == 0) tmax = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&tmax, t1);   // finish time of the LAST wave (arbitration favours old waves)
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = tmax - t0;
}

template <int NF, int NE, bool INDEP>
void run() {
  static float* out = nullptr; static unsigned long long* cyc = nullptr;
  if (!out) { hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long)); }
  const int iters = 2000;
  printf("fma %d exp %d %s:", NF, NE, INDEP ? "2 accs " : "1 acc  ");
  for (int wps = 1; wps <= 3; ++wps) {
    hipMemset(cyc, 0, 256 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NF, NE, INDEP>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) printf(" LAUNCH FAILED");
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += x; mean /= 256;
    printf("  %dw/SIMD %.1f cyc/gap/SIMD", wps, mean / iters / 8 / wps);
  }
  printf("\n");
}

int main() {
  run<0, 0, false>(); run<0, 0, true>();
  run<2, 0, false>(); run<4, 0, false>(); run<5, 0, false>(); run<6, 0, false>(); run<8, 0, false>(); run<12, 0, false>();
  run<0, 1, false>(); run<0, 2, false>(); run<0, 3, false>(); run<0, 4, false>();
  run<2, 1, false>(); run<4, 1, false>(); run<4, 2, false>(); run<6, 2, false>(); run<8, 2, false>();
  run<6, 2, true>();
  return 0;
}
