// Issue cost of packed-f32 VALU ops vs scalar ones, alone and in the shadow of an MFMA (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND, int N, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc = {0};
  f32x4 fa = {1e-3f * threadIdx.x, 2e-3f, 3e-3f, 4e-3f}, fb = {1e-3f, 2e-3f, 3e-3f, 5e-3f};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{0.5f + 0.01f * i, 0.25f + 1e-4f * threadIdx.x};
  f32x2 c = {0.999f, 1.001f}, d = {1e-4f, 2e-4f};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (WITH_MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7].x) : "v"(d.x));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(d));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7].x) : "v"(c.x), "v"(d.x));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c), "v"(d));
        if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(c));
        if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[i & 7].x) : "v"(v[(i + 1) & 7].y), "v"(d.x));
        if (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i & 7].x) : "v"(c.x), "v"(d.x));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  __shared__ unsigned long long tmax;
  if (threadIdx.x == 0) tmax = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&tmax, t1);
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = tmax - t0;
}

template <int KIND, int N, bool M>
void run(const char* name) {
  static float* out = nullptr; static unsigned long long* cyc = nullptr;
  if (!out) { hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long)); }
  const int iters = 2000;
  hipMemset(cyc, 0, 256 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, N, M>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto x : h) mean += x; mean /= 256;
  double per_gap = mean / iters / 8 / 2;  // 2 waves per SIMD
  if (M) printf("%-18s x%-2d + MFMA: %.1f cyc/gap/SIMD\n", name, N, per_gap);
  else printf("%-18s x%-2d alone : %.2f cyc per instr per SIMD\n", name, N, per_gap / N);
}

int main() {
  run<0, 8, false>("v_add_f32"); run<1, 8, false>("v_pk_add_f32"); run<2, 8, false>("v_fma_f32"); run<3, 8, false>("v_pk_fma_f32");
  run<4, 8, false>("v_pk_mul_f32"); run<5, 8, false>("v_cvt_pk_bf16_f32"); run<6, 8, false>("v_max3_f32");
  run<0, 8, true>("v_add_f32"); run<1, 4, true>("v_pk_add_f32"); run<1, 8, true>("v_pk_add_f32");
  run<2, 8, true>("v_fma_f32"); run<3, 4, true>("v_pk_fma_f32"); run<3, 8, true>("v_pk_fma_f32");
  run<5, 8, true>("v_cvt_pk_bf16_f32"); run<6, 8, true>("v_max3_f32"); run<4, 4, true>("v_pk_mul_f32");
  return 0;
}
