// Sustained MFMA rate of an MI355X under its power management (diagnostic, not part of the product).
// Every wave keeps its operands in registers and issues back-to-back v_mfma_f32_32x32x16_bf16 (or 16x16x32) on four
// independent accumulators; operands are random bf16 (the realistic case) or zeros (no toggling).  Prints TFLOP/s
// from hipEvent wall time and the shader clock from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const unsigned* in, float* out, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  unsigned w[8];
  for (int i = 0; i < 8; ++i) w[i] = in[(tid * 8 + i) & 0xFFFF];
  bf16x8 a = __builtin_bit_cast(bf16x8, *(uint4*)&w[0]);
  bf16x8 b = __builtin_bit_cast(bf16x8, *(uint4*)&w[4]);
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  float acc_sum = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc_sum += c[j][i];
  } else {
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) c[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc_sum += c[j][i];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  out[tid] = acc_sum;
  if ((threadIdx.x & 63) == 0) {
    clk[2 * (tid >> 6)] = c1 - c0;
    clk[2 * (tid >> 6) + 1] = r1 - r0;
  }
}

int main() {
  const int CUS = 256;
  unsigned* in; float* out; unsigned long long* clk;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, CUS * 512 * 4 * 4); hipMalloc(&clk, CUS * 8 * 2 * 8 * 4);
  std::vector<unsigned> h(65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {32, 16})
  for (int zero = 0; zero < 2; ++zero)
  for (int wps : {1, 2}) {
    srand(1);
    for (auto& x : h) {  // random bf16 pairs in [-2, 2): sign, exponent 125..128, random mantissa
      auto one = [&]() -> unsigned { return zero ? 0u : ((rand() & 1) << 15) | ((125 + (rand() & 3)) << 7) | (rand() & 127); };
      x = one() | (one() << 16);
    }
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int threads = 256 * wps, iters = 40000;
    auto launch = [&]() { if (shape == 32) k<32><<<CUS, threads>>>(in, out, clk, iters); else k<16><<<CUS, threads>>>(in, out, clk, iters); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int waves = CUS * threads / 64;
    std::vector<unsigned long long> hc(2 * waves);
    hipMemcpy(hc.data(), clk, 2 * waves * 8, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0; for (int i = 0; i < waves; ++i) { cs += hc[2 * i]; rs += hc[2 * i + 1]; }
    const double flops = (double)waves * iters * 4 * 32768.0;  // 4 x 32x32x16 == 8 x 16x16x32 per iteration
    printf("mfma %s  %-6s  %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s (%.1f%% of 2516.6)  shader clock %.3f GHz\n",
           shape == 32 ? "32x32x16" : "16x16x32", zero ? "zeros" : "random", wps, ms, flops / (ms * 1e-3) / 1e12,
           100 * flops / (ms * 1e-3) / 1e12 / 2516.6, 0.1 * cs / rs);
  }
  return 0;
}
