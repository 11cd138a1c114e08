// VERDICT r2 item 5, "by wall": the per-slot instruction mix of the one-wave-per-SIMD dK/dV kernel (fa_bwd_dkv_v3.hip:
// per 32x32x16 MFMA one v_exp_f32, one v_mul_f32, one v_cvt_pk_bf16_f32 and one 16-byte LDS fragment read feeding the MFMA)
// with the matrix work issued either as ONE v_mfma_f32_32x32x16_bf16 or as TWO v_mfma_f32_16x16x32_bf16 (same FLOPs, same
// output footprint) -- on RANDOM data, one wave per SIMD on every CU, >= 0.3 s of back-to-back launches.  Prints TFLOP/s
// from hipEvent wall time, the shader clock (s_memtime / s_memrealtime) and cycles per slot: the 16x16x32 form costs more
// issue cycles per slot and is granted a higher clock; the product decides.  Diagnostic only, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int SHAPE, int VALU>   // VALU: 1 = the kernel's mix, 0 = MFMA + LDS read only
__global__ __launch_bounds__(256, 1) void k(const unsigned* in, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 256) lds[i] = in[(i + 977 * blockIdx.x) & 0xFFFF];   // 64 KiB of random bf16
  __syncthreads();
  float v[16], w[16];
  for (int i = 0; i < 16; ++i) {
    const unsigned x = in[(tid * 16 + i + 31 * blockIdx.x) & 0xFFFF];
    v[i] = -(float)(x & 0xFFFF) / 32768.f;             // exponent arguments in (-2, 0]
    w[i] = (float)((x >> 16) & 0xFFFF) / 32768.f - 1.f;  // dP - delta stand-ins in [-1, 1)
  }
  u32x4 b = {in[tid & 0xFFFF], in[(tid + 256) & 0xFFFF], in[(tid + 512) & 0xFFFF], in[(tid + 768) & 0xFFFF]};
  u32x4 a[8];
  for (int j = 0; j < 8; ++j) a[j] = *(const u32x4*)&lds[(tid * 4 + 1024 * j) & 16383];
  f32x16 c32[4];
  f32x4 c16[16];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c32[j][i] = 0.f;
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) c16[j][i] = 0.f;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  unsigned off = (tid * 16) & 0x7FFF;
  float e_prev = 0.5f, m_prev = 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {   // 16 slots = one 32x32 block of the kernel
      if constexpr (SHAPE == 32) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c32[s & 3]) : "v"(a[s & 7]), "v"(b));
      } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[(2 * s) & 15]) : "v"(a[s & 7]), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c16[(2 * s + 1) & 15]) : "v"(a[(s + 1) & 7]), "v"(b));
      }
      // the fragment read that feeds an MFMA eight slots on (an 8-deep ring), base register + immediate as in the kernel
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[s & 7]) : "v"(off), "i"(2048 * (s & 15) + 16 * (s & 3)));
      if constexpr (VALU) {   // software-pipelined as in the kernel: every op consumes the PREVIOUS slot's result
        float e_new, m_new;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e_new) : "v"(v[s]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_new) : "v"(e_prev), "v"(w[s]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[s & 3]) : "v"(m_prev), "v"(e_prev));
        e_prev = e_new;
        m_prev = m_new;
      }
      asm volatile("s_waitcnt lgkmcnt(7)");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += c32[j][i];
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) sum += c16[j][i];
  out[blockIdx.x * 256 + tid] = sum + b[0] + e_prev + m_prev;
  if ((tid & 63) == 0) {
    clk[2 * (blockIdx.x * 4 + (tid >> 6))] = t1 - t0;
    clk[2 * (blockIdx.x * 4 + (tid >> 6)) + 1] = r1 - r0;
  }
}

template <int SHAPE, int VALU>
void run(const unsigned* in, float* out, unsigned long long* clk) {
  const int CUS = 256, iters = 4000;
  hipFuncSetAttribute((const void*)k<SHAPE, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() { hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(CUS), dim3(256), 100 * 1024, 0, in, out, clk, iters); };
  for (int i = 0; i < 60; ++i) launch();   // clock settle
  hipDeviceSynchronize();
  const int reps = 100;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(CUS * 4 * 2);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (size_t i = 0; i < h.size(); i += 2) { cyc += h[i]; rt += h[i + 1]; }
  const double flop = 2.0 * 32 * 32 * 16 * 16 * iters * 4.0 * CUS * reps;   // 16 slots x iters per wave, 4 waves per CU
  printf("%s, %-22s  %7.1f TFLOP/s   shader clock %.3f GHz   %.1f cycles per slot\n",
         SHAPE == 32 ? "1 x 32x32x16" : "2 x 16x16x32", VALU ? "exp + mul + cvt + LDS" : "LDS read only",
         flop / (ms * 1e-3) / 1e12, 0.1 * cyc / rt, cyc / (h.size() / 2) / (16.0 * iters));
}

int main() {
  unsigned* in; float* out; unsigned long long* clk;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 4 * 2 * 8);
  std::vector<unsigned> h(65536);
  srand(1);
  for (auto& x : h) {   // random bf16 pairs in [-2, 2)
    auto one = [&]() -> unsigned { return ((rand() & 1) << 15) | ((125 + (rand() & 3)) << 7) | (rand() & 127); };
    x = one() | (one() << 16);
  }
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {   // interleaved twice: the second round is the one to read
    run<32, 1>(in, out, clk);
    run<16, 1>(in, out, clk);
    run<32, 0>(in, out, clk);
    run<16, 0>(in, out, clk);
  }
  return 0;
}
