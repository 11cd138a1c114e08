// Same question as gap.hip for v_mfma_f32_16x16x32_bf16: per "gap" TWO 16x16x32 MFMAs (the FLOPs of one 32x32x16,
// 2 x 16 cycles) with NF v_fma_f32 + NE v_exp_f32 split between them; 1 / 2 / 3 waves per SIMD; cycles per gap per SIMD.
// Tells whether halving the MFMA length doubles its share of the vector issue port.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NF, int NE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4] = {{0}, {0}, {0}, {0}};
  f32x4 fa = {1e-3f * threadIdx.x, 2e-3f, 3e-3f, 4e-3f}, fb = {1e-3f, 2e-3f, 3e-3f, 5e-3f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 1e-4f * threadIdx.x;
  const float c = 0.999f, d = 1e-4f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(2 * g) & 3]) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = 0; i < NF / 2; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c), "v"(d));
#pragma unroll
      for (int i = 0; i < (NE + 1) / 2; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(NF + i) & 7]));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(2 * g + 1) & 3]) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = NF / 2; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(c), "v"(d));
#pragma unroll
      for (int i = (NE + 1) / 2; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(NF + i) & 7]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  __shared__ unsigned long long tmax;
  if (threadIdx.x == 0) tmax = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicMax(&tmax, t1);
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = tmax - t0;
}

template <int NF, int NE>
void run() {
  static float* out = nullptr; static unsigned long long* cyc = nullptr;
  if (!out) { hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long)); }
  const int iters = 2000;
  printf("2 x 16x16x32 + fma %d exp %d:", NF, NE);
  for (int wps = 1; wps <= 3; ++wps) {
    hipMemset(cyc, 0, 256 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NF, NE>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) printf(" LAUNCH FAILED");
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += x; mean /= 256;
    printf("  %dw/SIMD %.1f cyc/gap/SIMD", wps, mean / iters / 8 / wps);
  }
  printf("\n");
}

int main() {
  run<0, 0>(); run<2, 0>(); run<4, 0>(); run<6, 0>(); run<8, 0>();
  run<0, 2>(); run<2, 1>(); run<4, 1>(); run<4, 2>(); run<6, 2>(); run<8, 2>();
  return 0;
}
