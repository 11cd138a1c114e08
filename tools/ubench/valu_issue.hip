// Micro-benchmark: VALU / transcendental / MFMA issue throughput per SIMD as a function of the number
// of waves resident on the SIMD.  Answers: is a wave64 v_fma 2 or 4 cycles of SIMD time when several
// waves issue?  How much VALU hides under back-to-back MFMAs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  f32x16 acc = {0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(threadIdx.x * 0.01f); fb[i] = (__bf16)(i * 0.1f); }
  const float c = 1.0001f, d = 0.0003f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 64 independent fma per iteration
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], c, d);
    } else if (MODE == 1) {  // 64 exp
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if (MODE == 2) {  // 32 fma + 32 exp interleaved
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[i], c, d));
    } else if (MODE == 3) {  // 8 MFMA only
#pragma unroll
      for (int r = 0; r < 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    } else if (MODE == 4) {  // 8 MFMA + 40 fma (5 per MFMA gap)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[(r * 5 + i) & 15] = __builtin_fmaf(a[(r * 5 + i) & 15], c, d);
      }
    } else if (MODE == 6) {  // waves 0-3: 8 MFMA; waves 4-7: 96 fma (role split inside one block)
      if (threadIdx.x < 256) {
#pragma unroll
        for (int r = 0; r < 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], c, d);
      }
    } else if (MODE == 7) {  // waves 0-3: 8 MFMA; waves 4-7: 24 exp
      if (threadIdx.x < 256) {
#pragma unroll
        for (int r = 0; r < 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 24; ++r) a[r & 15] = __builtin_amdgcn_exp2f(a[r & 15]);
      }
    } else if (MODE == 8) {  // 8 MFMA on two independent accumulators, 5 fma in each gap, order pinned
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[(r * 5 + i) & 15] = __builtin_fmaf(a[(r * 5 + i) & 15], c, d);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 5) {  // 8 MFMA + 8*(2 exp + 4 fma)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        a[(2 * r) & 15] = __builtin_amdgcn_exp2f(a[(2 * r) & 15]);
        a[(2 * r + 1) & 15] = __builtin_amdgcn_exp2f(a[(2 * r + 1) & 15]);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[(r * 4 + i + 7) & 15] = __builtin_fmaf(a[(r * 4 + i + 7) & 15], c, d);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == blockDim.x - 64) cyc[blockIdx.x] = t1 - t0;
  if (threadIdx.x == 0) cyc[2048 + blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter_valu, int per_iter_mfma) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 8 * sizeof(float));
  hipMalloc(&cyc, 4096 * sizeof(unsigned long long));
  const int iters = 2000;
  for (int wps = (MODE >= 6 && MODE <= 7 ? 2 : 1); wps <= (MODE >= 6 && MODE <= 7 ? 2 : 4); ++wps) {          // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    int blocks = 256;                            // one block per CU; 4*wps waves per block = wps per SIMD
    hipMemset(cyc, 0, 4096 * sizeof(unsigned long long));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(e));
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    std::vector<unsigned long long> h0(blocks);
    hipMemcpy(h0.data(), cyc + 2048, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean0 = 0; for (auto v : h0) mean0 += v; mean0 /= blocks;
    if (MODE >= 6 && MODE <= 7) printf("   first wave (MFMA role) %.0f cyc/iter, last wave (VALU role) %.0f cyc/iter\n", mean0 / iters, mean / iters);
    // per SIMD: wps waves each did iters * per_iter instrs during `mean` cycles
    printf("%-34s waves/SIMD %d: %.0f cyc/iter/wave -> per SIMD %.2f cyc per VALU instr, %.2f cyc per MFMA\n", name, wps,
           mean / iters, per_iter_valu ? mean / iters / (per_iter_valu * wps) : 0.0,
           per_iter_mfma ? mean / iters / (per_iter_mfma * wps) : 0.0);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("64 fma", 64, 0);
  run<1>("64 exp", 64, 0);
  run<2>("32 fma + 32 exp", 64, 0);
  run<3>("8 mfma", 0, 8);
  run<4>("8 mfma + 40 fma", 40, 8);
  run<5>("8 mfma + 16 exp + 32 fma", 48, 8);
  run<8>("8 mfma + 40 fma, order pinned", 40, 8);
  printf("role split (2 waves/SIMD: one MFMA wave + one VALU wave):\n");
  run<6>("[8 mfma | 96 fma]", 96, 8);
  run<7>("[8 mfma | 24 exp]", 24, 8);
  return 0;
}
