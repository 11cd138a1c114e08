// FlashAttention backward dQ (+ delta), head dim 64, second-generation schedule for gfx950.
//
// Same maths and rounding points as fa_bwd_dq.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:165-258).  Mapping as in fa_fwd_v2.hip:
//   * workgroup = 4 waves = 256 query rows, every wave owns two 32-row query blocks whose Q^T / dO^T
//     fragments and dQ^T accumulators stay in registers; each K / V fragment read from LDS feeds the
//     MFMAs of BOTH blocks (S^T, dP^T by rows; dQ^T += K^T dS^T by transposed reads);
//   * K/V tiles (64 keys) arrive by LDS-DMA into a 3-deep ring, two tiles ahead, one raw s_barrier
//     per tile and a counted vmcnt -- no staging registers, no ds_write pass.
#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dq2Cfg {
  static constexpr int D = 64;
  static constexpr int BM = 256, BN = 64, NT = 256, NW = 4;
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BN * ROWB;
  static constexpr int RING = 3;
  static constexpr int V_BASE = RING * TILE_BYTES;
  static constexpr int LDS_BYTES = 2 * RING * TILE_BYTES;  // 48 KiB
  static constexpr int DMA_PER_MAT = TILE_BYTES / (NW * 1024);
};

// vmcnt(n) AND lgkmcnt(0): LDS reads issued on the current tile must have returned before the barrier (see fa_fwd.hip)
#define FA_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0070 | (n))

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void fa_bwd_dq2_kernel(BwdParams p) {
  using C = Dq2Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // causal: each workgroup takes the query-tile pair (nq-1-i, i) -> equal work everywhere (fa_fwd_v2.hip)
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.n_tiles + 1) / 2 : p.n_tiles;
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const int npass = (paired && idx != p.n_tiles - 1 - idx) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  // lane coordinates re-derived per pass (fa_common.h lane_id_now): nothing lane-dependent stays live across passes
  const int lane = lane_id_now(), tid = wave * 64 + lane, r = lane & 31, h = lane >> 5;
  const int qt = paired ? (pass == 0 ? p.n_tiles - 1 - idx : idx) : (CAUSAL ? p.n_tiles - 1 - idx : idx);  // heavy first
  const int q0_wg = qt * C::BM;
  const int qw0 = q0_wg + wave * 64;
  if (pass) __syncthreads();  // the previous pass staged its dQ tile in the ring

  const size_t qoff = (size_t)bh * p.Sq * C::ROWB, koff = (size_t)bh * p.Sk * C::ROWB;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc((const char*)p.q + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc((const char*)p.dout + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t ro = make_rsrc((const char*)p.o + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t rdq = make_rsrc((char*)p.dq + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t rk = make_rsrc((const char*)p.k + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc((const char*)p.v + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rl = make_rsrc(p.lse + (size_t)bh * p.Sq, (unsigned)p.Sq * 4);
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.delta + (size_t)bh * p.Sq, (unsigned)p.Sq * 4);

  const int kv_end = CAUSAL ? min(p.Sk, q0_wg + C::BM) : p.Sk;
  const int ntiles = (kv_end + C::BN - 1) / C::BN;
  int n_mine = CAUSAL ? min(ntiles, qw0 / C::BN + 1) : ntiles;
  if (qw0 >= p.Sq) n_mine = 0;
  const int nfull = min(n_mine, CAUSAL ? min(p.Sk / C::BN, qw0 / C::BN) : p.Sk / C::BN);

  if (p.Sk % C::BN != 0) {  // see fa_fwd_v2.hip: never expose uninitialised LDS behind a ragged last tile
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }

  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + 8 * i + (lane >> 3);
    dma_src[i] = row * C::ROWB + swz_chunk<D>(row, lane & 7) * 16;
  }
  auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int soff = t * C::TILE_BYTES;
#pragma unroll
    for (int i = 0; i < C::DMA_PER_MAT; ++i) {
      const int dst = buf * C::TILE_BYTES + (16 * wave + 8 * i) * C::ROWB;
      dma16(rk, lds_addr_of(smem + dst), dma_src[i], soff);
      dma16(rv, lds_addr_of(smem + C::V_BASE + dst), dma_src[i], soff);
    }
  };
  constexpr int DMA_PER_TILE = 2 * C::DMA_PER_MAT;

  dma_tile(0, 0);
  if (ntiles > 1) dma_tile(1, 1);

  // ---- resident B operands of both query blocks: Q^T, dO^T; delta = rowsum(dO * O); -LSE*log2(e) ----
  vec8 qf[2][C::KS], dof[2][C::KS];
  float delta[2], nl[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int off = (qw0 + 32 * j + r) * C::ROWB + (2 * ks + h) * 16;
      qf[j][ks] = as_vec8<T>(buf_load16(rq, off));
      dof[j][ks] = as_vec8<T>(buf_load16(rdo, off));
      const vec8 of = as_vec8<T>(buf_load16(ro, off));
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = __builtin_fmaf((float)dof[j][ks][e], (float)of[e], dsum);
    }
    delta[j] = half_sum(dsum);
    nl[j] = -buf_load_f32(rl, (qw0 + 32 * j + r) * 4) * kLog2e;
    if (h == 0) buf_store_f32(rd, (qw0 + 32 * j + r) * 4, delta[j]);
  }

  int row_off[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
  int tr_off[2][C::DB];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) tr_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);

  const float c2 = p.scale * kLog2e;
  // FOLD (bf16, fa_common.h / fa_bwd_dq.hip): Q carries c2, the same rounded operand the folded forward kernels used for
  // LSE, so the recomputed P is consistent with it.  Unlike fa_bwd_dq.hip the chains still start from zero (two query
  // blocks per wave: starting them from -LSE*log2e / -delta keeps 64 more registers live and spills several hundred).
  constexpr bool FOLD = T::kFoldScale;
  if constexpr (FOLD) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) qf[j][ks] = scale_frag<T>(qf[j][ks], c2);
  }
  if constexpr (FOLD) {
    if (p.qs) {  // workspace for the dK/dV launch (fa_kernels.h BwdParams::qs); this family runs on contiguous tensors only
      const __amdgpu_buffer_rsrc_t rqs = make_rsrc((char*)p.qs + qoff, (unsigned)p.Sq * C::ROWB);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
          buf_store16(rqs, (qw0 + 32 * j + r) * C::ROWB + (2 * ks + h) * 16, __builtin_bit_cast(u32x4, qf[j][ks]));
    }
  }
  f32x16 dqacc[2][C::DB];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dqacc[j][db][i] = 0.f;

  // one 32-key block against both query blocks; USE0/USE1: which query blocks take part;
  // MASKED adds the per-element causal / key-tail test.
  auto key_block = [&](const FA_LDS char* kbp, const FA_LDS char* vbp, int key0, auto masked_tag, bool use0,
                       bool use1) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[j][i] = 0.f;
        dp[j][i] = 0.f;
      }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      vec8 a = as_vec8<T>(lds_read16(kbp + row_off[ks]));
      if (!MASKED || use0) s[0] = T::mfma(a, qf[0][ks], s[0]);
      if (!MASKED || use1) s[1] = T::mfma(a, qf[1][ks], s[1]);
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      vec8 a = as_vec8<T>(lds_read16(vbp + row_off[ks]));
      if (!MASKED || use0) dp[0] = T::mfma(a, dof[0][ks], dp[0]);
      if (!MASKED || use1) dp[1] = T::mfma(a, dof[1][ks], dp[1]);
    }
    vec8 dsf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? s[j][i] + nl[j] : __builtin_fmaf(s[j][i], c2, nl[j]);
        if constexpr (MASKED) {
          const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool dead = (CAUSAL && key > qw0 + 32 * j + r) || key >= p.Sk;
          x = dead ? -INFINITY : x;
        }
        s[j][i] = __builtin_amdgcn_exp2f(x) * (dp[j][i] - delta[j]);  // dS^T = P^T o (dP^T - delta)
      }
      dsf[j][0] = pack8<T, 0>(s[j]);
      dsf[j][1] = pack8<T, 1>(s[j]);
    }
#pragma unroll
    for (int db = 0; db < C::DB; ++db) {
      vec8 a0 = lds_read_tr_frag<T>(kbp + tr_off[0][db], kbp + tr_off[1][db]);
      if (!MASKED || use0) dqacc[0][db] = T::mfma(a0, dsf[0][0], dqacc[0][db]);
      if (!MASKED || use1) dqacc[1][db] = T::mfma(a0, dsf[1][0], dqacc[1][db]);
      vec8 a1 = lds_read_tr_frag<T>(kbp + 16 * C::ROWB + tr_off[0][db], kbp + 16 * C::ROWB + tr_off[1][db]);
      if (!MASKED || use0) dqacc[0][db] = T::mfma(a1, dsf[0][1], dqacc[0][db]);
      if (!MASKED || use1) dqacc[1][db] = T::mfma(a1, dsf[1][1], dqacc[1][db]);
    }
  };

  auto tile_full = [&](int buf) __attribute__((always_inline)) {
    const FA_LDS char* kt = smem + buf * C::TILE_BYTES;
    const FA_LDS char* vt = smem + C::V_BASE + buf * C::TILE_BYTES;
#pragma unroll
    for (int b = 0; b < 2; ++b)
      key_block(kt + b * 32 * C::ROWB, vt + b * 32 * C::ROWB, 0, std::false_type{}, true, true);
  };
  auto tile_masked = [&](int t, int buf) __attribute__((always_inline)) {
    const FA_LDS char* kt = smem + buf * C::TILE_BYTES;
    const FA_LDS char* vt = smem + C::V_BASE + buf * C::TILE_BYTES;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int key0 = t * C::BN + 32 * b;
      if (key0 >= p.Sk) continue;
      const bool use0 = !CAUSAL || key0 <= qw0;
      const bool use1 = !CAUSAL || key0 <= qw0 + 32;
      if (!use0 && !use1) continue;
      key_block(kt + b * 32 * C::ROWB, vt + b * 32 * C::ROWB, key0, std::true_type{}, use0, use1);
    }
  };

  auto ring_sync = [&](bool fetched) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    if (fetched) FA_WAIT_VMCNT(DMA_PER_TILE); else FA_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto step_full = [&](int t, auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    const bool fetch = t + 2 < ntiles;
    if (fetch) dma_tile(t + 2, (BUF + 2) % 3);
    tile_full(BUF);
    ring_sync(fetch);
  };
  auto step_tail = [&](int t) __attribute__((always_inline)) {
    const bool fetch = t + 2 < ntiles;
    if (fetch) dma_tile(t + 2, (t + 2) % 3);
    if (t < n_mine) tile_masked(t, t % 3);
    ring_sync(fetch);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;

  asm volatile("" ::: "memory");
  FA_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int t = 0;
  for (; t + 3 <= nfull; t += 3) {
    step_full(t, B0{});
    step_full(t + 1, B1{});
    step_full(t + 2, B2{});
  }
  if (t < nfull) {
    step_full(t, B0{});
    ++t;
    if (t < nfull) {
      step_full(t, B1{});
      ++t;
    }
  }
  for (; t < ntiles; ++t) step_tail(t);

#pragma unroll
  for (int j = 0; j < 2; ++j)
    store_tile_rows<D, T>(dqacc[j], p.scale, smem + (wave * 2 + j) * 32 * C::ROWB, rdq, (qw0 + 32 * j) * C::ROWB, lane);
  }  // pass
}

template <typename T, bool CAUSAL>
static hipError_t launch2(const BwdParams& p, hipStream_t s) {
  using C = Dq2Cfg;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  auto kern = fa_bwd_dq2_kernel<T, CAUSAL>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dq_v2(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sq + Dq2Cfg::BM - 1) / Dq2Cfg::BM;
  p.pair = causal != 0;
  if (dtype == 1) return causal ? launch2<BF16, true>(p, s) : launch2<BF16, false>(p, s);
  return causal ? launch2<FP16, true>(p, s) : launch2<FP16, false>(p, s);
}

}  // namespace fa
