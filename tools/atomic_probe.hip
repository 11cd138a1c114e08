// Probe: fp32 atomic-add throughput in the pattern a fused backward would need for dQ
// (each kv-tile workgroup adds a 128x64 fp32 tile per q-tile it visits; causal, B*H=128, S=4096).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* __restrict__ acc, int S, int D, int nt) {
  // blockIdx.x: xcd-affine head mapping like the kernels: head = (id & 7) * (BH/8) + (id >> 3) / nt ...
  const int id = blockIdx.x;
  const int BH = gridDim.x / nt;
  const int x = id & 7, j = id >> 3;
  const int per = BH / 8;
  const int head = x * per + j / nt;
  const int kt = nt - 1 - (j % nt);          // heavy first
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* base = acc + (size_t)head * S * D;
  for (int qt = kt; qt < nt; ++qt) {
    // wave w owns rows [32w, 32w+32) of the 128-row q-tile; 32x32 accumulator layout x 2 column blocks
    float* tile = base + (size_t)(qt * 128 + w * 32) * D;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        float* p = tile + row * D + db * 32 + (lane & 31);
        const float v = 1.0f;
        if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else *p = v;                        // plain stores, for reference
      }
  }
}

int main() {
  const int BH = 128, S = 4096, D = 64, nt = S / 128;
  float* acc; CK(hipMalloc(&acc, (size_t)BH * S * D * 4));
  CK(hipMemset(acc, 0, (size_t)BH * S * D * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)BH * (nt * (nt + 1) / 2) * 128 * D * 4;
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0));
      if (mode == 0) probe<0><<<BH * nt, 256>>>(acc, S, D, nt);
      if (mode == 1) probe<1><<<BH * nt, 256>>>(acc, S, D, nt);
      if (mode == 2) probe<2><<<BH * nt, 256>>>(acc, S, D, nt);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    printf("mode %d (%s): %.3f ms  %.2f TB/s of fp32 adds (%.2f GB)\n", mode,
           mode == 0 ? "atomic agent" : mode == 1 ? "atomic wg-scope" : "plain store", best, bytes / best / 1e9, bytes / 1e9);
  }
  std::vector<float> h(64);
  CK(hipMemcpy(h.data(), acc, 256, hipMemcpyDeviceToHost));
  printf("acc[0]=%g\n", h[0]);
  return 0;
}
