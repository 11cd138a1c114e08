// FlashAttention backward dK / dV, head dim 64, third-generation schedule: one wave per SIMD.
//
// Same maths and rounding points as fa_bwd_dkv.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:292-386).  Geometry:
//   * workgroup = 4 waves = 256 keys, ONE workgroup per CU: a wave owns the whole 512-entry register
//     file of its SIMD.  It keeps K^T / V^T fragments and the dK^T / dV^T accumulators of TWO 32-key
//     blocks (192 registers of stationary state, all of it MFMA-only operands);
//   * every fragment of the streamed Q / dO tile is read from LDS once and feeds the MFMAs of both key
//     blocks (3/8 of the LDS reads per MFMA of the 32-key-per-wave kernel), and a tile costs half the
//     LDS-DMA issue per MFMA;
//   * the two query blocks of a tile and the two key blocks give four independent
//     (S, dP) -> (P, dS) -> (dV, dK) chains in ONE basic block: latency is hidden by instruction-level
//     parallelism inside the wave instead of by a second wave that would halve the register budget.
// Q/dO tiles: 64 rows, LDS-DMA double buffer + pre-scaled row constants, as in fa_bwd_dkv_v2.hip.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dkv3Cfg {
  static constexpr int D = 64;
  static constexpr int BK = 256, BQ = 64, NT = 256, NW = 4;
  static constexpr int ROWB = D * 2, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BQ * ROWB;             // 8 KiB per matrix
  static constexpr int DO_BASE = 2 * TILE_BYTES;           // Q[2], then dO[2]
  static constexpr int ROWC_OFF = 4 * TILE_BYTES;          // then row constants: nl[64], nd[64] per buffer
  static constexpr int ROWC_BYTES = 2 * BQ * 4;
  static constexpr int TILES_BYTES = 4 * TILE_BYTES + 2 * ROWC_BYTES;  // 33 KiB
  // epilogue staging needs 4 waves x 2 key blocks x 4 KiB = 32 KiB (fits in the tile buffers)
  static constexpr int LDS_BYTES = TILES_BYTES;
  static constexpr int DMA_PER_MAT = TILE_BYTES / (NW * 1024);
};

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void fa_bwd_dkv3_kernel(BwdParams p) {
  using C = Dkv3Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.n_tiles + 1) / 2 : p.n_tiles;
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const int npass = (paired && idx != p.n_tiles - 1 - idx) ? 2 : 1;

  const size_t qoff = (size_t)bh * p.Sq * C::ROWB, koff = (size_t)bh * p.Sk * C::ROWB;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc((const char*)p.q + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc((const char*)p.dout + qoff, (unsigned)p.Sq * C::ROWB);
  const __amdgpu_buffer_rsrc_t rk = make_rsrc((const char*)p.k + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc((const char*)p.v + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rdk = make_rsrc((char*)p.dk + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rdv = make_rsrc((char*)p.dv + koff, (unsigned)p.Sk * C::ROWB);
  const __amdgpu_buffer_rsrc_t rl = make_rsrc(p.lse + (size_t)bh * p.Sq, (unsigned)p.Sq * 4);
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.delta + (size_t)bh * p.Sq, (unsigned)p.Sq * 4);

  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + 8 * i + (lane >> 3);
    dma_src[i] = row * C::ROWB + swz_chunk<D>(row, lane & 7) * 16;
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
  const int ntiles = (p.Sq + C::BQ - 1) / C::BQ;

  if (p.Sq % C::BQ != 0) {  // ragged last query tile: its tail rows come from an out-of-range DMA
    for (int i = tid * 16; i < C::LDS_BYTES; i += C::NT * 16) lds_write16(smem + i, u32x4{0, 0, 0, 0});
    __syncthreads();
  }

  for (int pass = 0; pass < npass; ++pass) {
    const int kt_idx = paired ? (pass == 0 ? idx : p.n_tiles - 1 - idx) : idx;  // low key tiles are the heavy ones
    const int k0_wg = kt_idx * C::BK;
    const int kw0 = k0_wg + wave * 64;
    if (pass) __syncthreads();  // previous pass staged dK / dV in the tile buffers

    const int t_start = CAUSAL ? k0_wg / C::BQ : 0;
    const int t_full = CAUSAL ? kw0 / C::BQ + 1 : 0;  // tiles >= t_full lie entirely below the diagonal for this wave

    float rc = 0.f;
    auto fetch_tile = [&](int t, int buf) __attribute__((always_inline)) {
      const int soff = t * C::TILE_BYTES;
#pragma unroll
      for (int i = 0; i < C::DMA_PER_MAT; ++i) {
        const int dst = buf * C::TILE_BYTES + (16 * wave + 8 * i) * C::ROWB;
        dma16(rq, lds_addr_of(smem + dst), dma_src[i], soff);
        dma16(rdo, lds_addr_of(smem + C::DO_BASE + dst), dma_src[i], soff);
      }
      if (wave == 0) rc = buf_load_f32(rl, (t * C::BQ + lane) * 4);
      else if (wave == 1) rc = buf_load_f32(rd, (t * C::BQ + lane) * 4);
    };
    auto commit_tile = [&](int t, int buf, bool fetched) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): DMA + row constants landed
      if (fetched) {
        FA_LDS float* rcp = (FA_LDS float*)(smem + C::ROWC_OFF + buf * C::ROWC_BYTES);
        if (wave == 0) rcp[lane] = (t * C::BQ + lane < p.Sq) ? -rc * kLog2e : -INFINITY;  // rows past S_q: P = 0 (K:355-356)
        else if (wave == 1) rcp[64 + lane] = -rc;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };

    if (t_start < ntiles) fetch_tile(t_start, t_start & 1);

    // ---- stationary: K^T / V^T fragments and dK^T / dV^T accumulators of the wave's two key blocks ----
    vec8 kf[2][C::KS], vf[2][C::KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (kw0 + 32 * kb + r) * C::ROWB + (2 * ks + h) * 16;
        kf[kb][ks] = as_vec8<T>(buf_load16(rk, off));
        vf[kb][ks] = as_vec8<T>(buf_load16(rv, off));
      }
    f32x16 dkacc[2][C::DB], dvacc[2][C::DB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dkacc[kb][db][i] = 0.f;
          dvacc[kb][db][i] = 0.f;
        }

    // one 32-row query block against both key blocks
    auto q_block = [&](int buf, int b, int qb0, auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      const FA_LDS char* qbp = smem + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* dbp = smem + C::DO_BASE + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
      f32x16 nl, sacc[2], pacc[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // per-register row constants: reg i <-> row (i&3) + 8(i>>2) + 4h
        const f32x4 a = *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * g + 4 * h) * 4);
        const f32x4 d = *(const FA_LDS f32x4*)(rcp + (64 + 32 * b + 8 * g + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nl[4 * g + j] = a[j];
          pacc[0][4 * g + j] = d[j];
          pacc[1][4 * g + j] = d[j];
        }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(qbp + row_off[ks]));
        sacc[0] = T::mfma(a, kf[0][ks], sacc[0]);
        sacc[1] = T::mfma(a, kf[1][ks], sacc[1]);
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(dbp + row_off[ks]));
        pacc[0] = T::mfma(a, vf[0][ks], pacc[0]);
        pacc[1] = T::mfma(a, vf[1][ks], pacc[1]);
      }
      vec8 pf[2][2], sf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float x = __builtin_fmaf(sacc[kb][i], c2, nl[i]);
          if constexpr (MASKED) {
            const int qrow = qb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            x = (kw0 + 32 * kb + r > qrow) ? -INFINITY : x;
          }
          const float pe = __builtin_amdgcn_exp2f(x);
          sacc[kb][i] = pe;                   // P
          pacc[kb][i] = pe * pacc[kb][i];     // dS = P o (dP - delta)
        }
        pf[kb][0] = pack8<T, 0>(sacc[kb]);
        pf[kb][1] = pack8<T, 1>(sacc[kb]);
        sf[kb][0] = pack8<T, 0>(pacc[kb]);
        sf[kb][1] = pack8<T, 1>(pacc[kb]);
      }
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        vec8 a0 = lds_read_tr_frag<T>(dbp + tr_off[0][db], dbp + tr_off[1][db]);
        dvacc[0][db] = T::mfma(a0, pf[0][0], dvacc[0][db]);
        dvacc[1][db] = T::mfma(a0, pf[1][0], dvacc[1][db]);
        vec8 a1 = lds_read_tr_frag<T>(dbp + 16 * C::ROWB + tr_off[0][db], dbp + 16 * C::ROWB + tr_off[1][db]);
        dvacc[0][db] = T::mfma(a1, pf[0][1], dvacc[0][db]);
        dvacc[1][db] = T::mfma(a1, pf[1][1], dvacc[1][db]);
      }
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        vec8 a0 = lds_read_tr_frag<T>(qbp + tr_off[0][db], qbp + tr_off[1][db]);
        dkacc[0][db] = T::mfma(a0, sf[0][0], dkacc[0][db]);
        dkacc[1][db] = T::mfma(a0, sf[1][0], dkacc[1][db]);
        vec8 a1 = lds_read_tr_frag<T>(qbp + 16 * C::ROWB + tr_off[0][db], qbp + 16 * C::ROWB + tr_off[1][db]);
        dkacc[0][db] = T::mfma(a1, sf[0][1], dkacc[0][db]);
        dkacc[1][db] = T::mfma(a1, sf[1][1], dkacc[1][db]);
      }
    };

    auto step_full = [&](int t, auto buf_tag) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_tag)::value;
      const bool more = t + 1 < ntiles;
      if (more) fetch_tile(t + 1, BUF ^ 1);
      q_block(BUF, 0, 0, std::false_type{});
      q_block(BUF, 1, 0, std::false_type{});
      commit_tile(t + 1, BUF ^ 1, more);
    };
    auto step_masked = [&](int t) __attribute__((always_inline)) {
      const int buf = t & 1;
      const bool more = t + 1 < ntiles;
      if (more) fetch_tile(t + 1, buf ^ 1);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int qb0 = t * C::BQ + 32 * b;
        if (qb0 < kw0) continue;  // every row of the block is above the diagonal of both key blocks
        q_block(buf, b, qb0, std::true_type{});
      }
      commit_tile(t + 1, buf ^ 1, more);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    commit_tile(t_start, t_start & 1, t_start < ntiles);

    int t = t_start;
    const int t_masked_end = min(ntiles, t_full);
    for (; t < t_masked_end; ++t) step_masked(t);
    if (t < ntiles && (t & 1)) {
      step_full(t, B1{});
      ++t;
    }
    for (; t + 2 <= ntiles; t += 2) {
      step_full(t, B0{});
      step_full(t + 1, B1{});
    }
    if (t < ntiles) step_full(t, B0{});

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      FA_LDS char* stage = smem + (wave * 2 + kb) * 32 * C::ROWB;
      store_tile_rows<D, T>(dkacc[kb], p.scale, stage, rdk, (kw0 + 32 * kb) * C::ROWB, lane);
      store_tile_rows<D, T>(dvacc[kb], 1.0f, stage, rdv, (kw0 + 32 * kb) * C::ROWB, lane);
    }
  }  // pass
}

template <typename T, bool CAUSAL>
static hipError_t launch3(const BwdParams& p, hipStream_t s) {
  using C = Dkv3Cfg;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  auto kern = fa_bwd_dkv3_kernel<T, CAUSAL>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dkv_v3(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sk + Dkv3Cfg::BK - 1) / Dkv3Cfg::BK;
  p.pair = causal != 0;
  if (dtype == 1) return causal ? launch3<BF16, true>(p, s) : launch3<BF16, false>(p, s);
  return causal ? launch3<FP16, true>(p, s) : launch3<FP16, false>(p, s);
}

}  // namespace fa
