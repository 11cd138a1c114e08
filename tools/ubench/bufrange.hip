// Does the raw-buffer range check of gfx950 include the SCALAR offset?  The kernels address tile t of a (batch, head)
// slice as voffset (row-in-tile * stride + chunk) + soffset (t * tile_bytes) and rely on rows past the slice reading 0.
// Buffer of N bytes inside a 2N-byte allocation filled with 1.0: any 1.0 read at an offset >= N means "not checked".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* base, float* out, int n_bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n_bytes, 0x00020000);
  const int lane = threadIdx.x, voff = lane * 4;
  auto ld = [&](int v, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0)); };
  out[0 * 64 + lane] = ld(voff, 0);                  // in range
  out[1 * 64 + lane] = ld(voff, n_bytes);            // past the end through soffset only
  out[2 * 64 + lane] = ld(voff + n_bytes, 0);        // past the end through voffset
  out[3 * 64 + lane] = ld(voff, n_bytes - 128);      // straddles the end through soffset: lanes 0..31 in range
  out[4 * 64 + lane] = ld(voff + n_bytes - 128, 0);  // straddles the end through voffset
}
int main() {
  const int N = 4096;
  float *buf, *out;
  hipMalloc(&buf, 2 * N); hipMalloc(&out, 5 * 64 * 4);
  std::vector<float> ones(2 * N / 4, 1.0f);
  hipMemcpy(buf, ones.data(), 2 * N, hipMemcpyHostToDevice);
  k<<<1, 64>>>(buf, out, N);
  std::vector<float> h(5 * 64);
  hipMemcpy(h.data(), out, 5 * 64 * 4, hipMemcpyDeviceToHost);
  const char* names[5] = {"in range (expect 64 ones)", "soffset past end (0 ones if checked)", "voffset past end (expect 0 ones)",
                          "soffset straddles (32 ones if checked)", "voffset straddles (expect 32 ones)"};
  for (int c = 0; c < 5; ++c) {
    int n = 0; for (int l = 0; l < 64; ++l) n += h[c * 64 + l] == 1.0f;
    printf("%-42s: %d lanes read 1.0\n", names[c], n);
  }
  return 0;
}
