// FlashAttention backward, key-tile-stationary half: dK and dV.
//
// Replaces the reference's flash_attention_dKV_kernel
// (code/_flash_attention_kernel_optimized.py:292-386).  Semantics kept: runs AFTER the dQ
// kernel and loads the delta it stored (K:376, launch order M:111-126); P recomputed from
// LSE (K:367); P^T and dS^T rounded to the input dtype before their matmuls (K:370, K:382);
// padded query rows contribute nothing (K:355-356); causal loop starts at the key tile
// (K:341); dK, dV cast on store (K:385-386).  (scale applied once to the fp32 dK.)
//
// Decomposition: workgroup = 4 waves = 128 keys of one (batch, head); wave = 32 keys whose
// K and V fragments stay in registers as MFMA B operands, and whose dK^T / dV^T tiles
// ([d][key], key on the lane) stay in fp32 accumulators for the whole kernel -- no
// cross-workgroup reduction.  Q and dO stream through LDS in 64-row tiles (one swizzled
// image each, read by rows for S / dP and transposed for dV^T / dK^T), with
// -LSE*log2(e) and -delta staged beside them.  With the key on the lane,
//     S  = Q K^T             (A = Q rows,  B = K^T resident)
//     dP = dO V^T - delta    (A = dO rows, B = V^T resident; -delta[q] preloaded as C)
// have the query index in the accumulator REGISTER, so P and dS = P o dP are, after
// rounding, directly the B operands of dV^T += dO^T P and dK^T += Q^T dS.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

template <int D>
struct DkvCfg {
  static constexpr int BK = 128;  // keys per workgroup
  static constexpr int BQ = 64;   // query rows per LDS tile
  static constexpr int NT = 256;
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BQ * ROWB;
  static constexpr int DMA_PER_MAT = TILE_BYTES / (4 * 1024);  // 1-KiB LDS-DMA instructions per wave per matrix
  static constexpr int ROWC_OFF = 4 * TILE_BYTES;           // row constants after Q[2], dO[2]
  static constexpr int ROWC_BYTES = 2 * BQ * 4;             // nl[64], nd[64] per buffer
  static constexpr int LDS_BYTES = 4 * TILE_BYTES + 2 * ROWC_BYTES;
};

// DROP: attention dropout (fa_common.h `Dropout`): dV uses the masked, rescaled P; dP = mask / (1 - p) o (dO V^T), so the
// dP chain starts from zero and -delta is added per element.
template <int D, typename T, bool CAUSAL, bool DROP = false>
__global__ __launch_bounds__(256, (D == 64 ? 2 : 1)) void fa_bwd_dkv_kernel(BwdParams p) {
  using C = DkvCfg<D>;
  using vec8 = typename T::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // causal: key tile i meets the query tiles from i on, so a workgroup takes the PAIR (i, nk-1-i)
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.n_tiles + 1) / 2 : p.n_tiles;
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const BatchHead ix = batch_head(bh, p.B, p.H, p.vl.cu_q != nullptr);
  const int b_ = ix.b, h_ = ix.h;
  // variable-length launch (fa_kernels.h VarLen): this sequence's rows and lengths; surplus workgroups exit
  const SeqInfo si = seq_info(p.vl, b_, p.Sq, p.Sk);
  const int Sq = si.Sq, Sk = si.Sk;
  const int nk = (Sk + C::BK - 1) / C::BK;
  if (idx >= (paired ? (nk + 1) / 2 : nk)) return;
  const int npass = (paired && idx != nk - 1 - idx) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int kt_idx = paired ? (pass == 0 ? idx : nk - 1 - idx) : idx;  // low key tiles are the heavy ones
  const int k0_wg = kt_idx * C::BK;
  const int kw0 = k0_wg + wave * 32;
  if (pass) __syncthreads();  // the previous pass staged dK / dV in the tile buffers

  // Q, K, V, dO may be strided views with a contiguous head dim (fa_fwd.hip); dK and dV carry their own layouts
  // (contiguous for the reference's launch, packed rows for varlen); LSE / delta rows of one (batch, head) are contiguous
  const int q_rs = p.lq.rs, do_rs = p.ldo.rs, kv_rs = p.lk.rs, dk_rs = p.ldk.rs, dv_rs = p.ldv.rs;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(
      (const char*)p.q + b_ * p.lq.sb + h_ * p.lq.sh + (long long)si.q0 * q_rs, (unsigned)(Sq - 1) * q_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(
      (const char*)p.dout + b_ * p.ldo.sb + h_ * p.ldo.sh + (long long)si.q0 * do_rs, (unsigned)(Sq - 1) * do_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(
      (const char*)p.k + b_ * p.lk.sb + h_ * p.lk.sh + (long long)si.k0 * kv_rs, (unsigned)(Sk - 1) * kv_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(
      (const char*)p.v + b_ * p.lv.sb + h_ * p.lv.sh + (long long)si.k0 * kv_rs, (unsigned)(Sk - 1) * kv_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rdk = make_rsrc(
      (char*)p.dk + b_ * p.ldk.sb + h_ * p.ldk.sh + (long long)si.k0 * dk_rs, (unsigned)(Sk - 1) * dk_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rdv = make_rsrc(
      (char*)p.dv + b_ * p.ldv.sb + h_ * p.ldv.sh + (long long)si.k0 * dv_rs, (unsigned)(Sk - 1) * dv_rs + C::ROWB);
  const long long rowc_off = b_ * p.lse_sb + h_ * p.lse_sh + si.q0;
  // Row constants of a query tile: wave 0 loads its LSE rows, wave 1 its delta rows, through ONE wave-uniform
  // descriptor and an unconditional load (a divergent `if` around the load makes hipcc wait vmcnt(0) at the merge,
  // which also waits for the tile DMA issued just before: the double buffer then hides nothing).
  const __amdgpu_buffer_rsrc_t rrc =
      make_rsrc((wave == 0 ? p.lse : p.delta) + rowc_off, wave < 2 ? (unsigned)Sq * 4 : 0u);

  const float c2 = p.scale * kLog2e;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h: the score chain starts from -LSE*log2e and K carries c2
  // ---- resident B operands: K^T and V^T of this wave's 32 keys ----
  vec8 kf[C::KS], vf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int off = (kw0 + r) * kv_rs + (2 * ks + h) * 16;
    kf[ks] = as_vec8<T>(buf_load16(rk, off));
    if (FOLD && !p.q_prescaled) kf[ks] = scale_frag<T>(kf[ks], c2);  // K * softmax_scale * log2(e)
    vf[ks] = as_vec8<T>(buf_load16(rv, off));
  }

  const int ntiles = (Sq + C::BQ - 1) / C::BQ;
  const int t_start = CAUSAL ? k0_wg / C::BQ : 0;
  // tiles t >= t_full are entirely below the diagonal for this wave's keys
  const int t_full = CAUSAL ? kw0 / C::BQ + 1 : 0;

  // LDS-DMA source offsets (see fa_fwd.hip): wave w fills rows [16w, 16w+16) of each tile
  constexpr int RPI = 1024 / C::ROWB;
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + RPI * i + lane / C::CPR;
    dma_src[i] = row * q_rs + swz_chunk<D>(row, lane % C::CPR) * 16;
  }
  // the dO tile has the same lane -> (row, chunk) map; only its row stride may differ (the difference can be
  // negative: it is added in the VGPR offset, whose sum row*do_rs + chunk is not; the scalar offset is unsigned)
  const int do_delta = (16 * wave + lane / C::CPR) * (do_rs - q_rs);
  int row_off[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
  int tr_off[2][C::DB];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) tr_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);

  f32x16 dkacc[C::DB], dvacc[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dkacc[db][i] = 0.f;
      dvacc[db][i] = 0.f;
    }

  float cst = 0.f;  // threads 0..63: LSE row, 64..127: delta row
  auto stage_load = [&](int t) __attribute__((always_inline)) {
    const int soff_q = t * C::BQ * q_rs, soff_do = t * C::BQ * do_rs;
    const int buf = t & 1;
#pragma unroll
    for (int i = 0; i < C::DMA_PER_MAT; ++i) {
      const int dst = buf * C::TILE_BYTES + (16 * wave + RPI * i) * C::ROWB;
      dma16(rq, lds_addr_of(smem + dst), dma_src[i], soff_q);
      dma16(rdo, lds_addr_of(smem + 2 * C::TILE_BYTES + dst), dma_src[i] + do_delta + RPI * i * (do_rs - q_rs), soff_do);
    }
    cst = buf_load_f32(rrc, (t * C::BQ + lane) * 4);
  };
  // tile t (fetched during the previous step) has landed: publish its pre-scaled row constants, then meet
  auto stage_write = [&](int t) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    FA_LDS float* rc = (FA_LDS float*)(smem + C::ROWC_OFF + (t & 1) * C::ROWC_BYTES);
    // rows past S_q must give P = 0 (K:355-356): exp2(-inf) = 0
    const float lse_c = (t * C::BQ + lane < Sq) ? -cst * kLog2e : -INFINITY;
    if (wave < 2) rc[tid] = wave == 0 ? lse_c : -cst;  // rc[row] = -LSE*log2e, rc[64 + row] = -delta
  };

  auto tile = [&](int t, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int buf = t & 1;
    const FA_LDS char* qt = smem + buf * C::TILE_BYTES;
    const FA_LDS char* dt = smem + (2 + buf) * C::TILE_BYTES;
    const FA_LDS char* rc = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
    const int q0 = t * C::BQ;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int qb0 = q0 + 32 * b;
      if constexpr (MASKED) {
        if (qb0 < kw0) continue;  // every row of the block is above the diagonal
      }
      const FA_LDS char* qbp = qt + b * 32 * C::ROWB;
      const FA_LDS char* dbp = dt + b * 32 * C::ROWB;
      // per-register row constants: reg i <-> row (i&3) + 8(i>>2) + 4h
      // both MFMA chains START from them: with K pre-scaled the first delivers s*c2 - LSE*log2e, the second dP - delta
      f32x16 nl, nd, pacc, sacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = *(const FA_LDS f32x4*)(rc + (32 * b + 8 * g + 4 * h) * 4);
        const f32x4 d = *(const FA_LDS f32x4*)(rc + (64 + 32 * b + 8 * g + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nl[4 * g + j] = a[j];
          nd[4 * g + j] = d[j];
          sacc[4 * g + j] = FOLD ? a[j] : 0.f;
          pacc[4 * g + j] = DROP ? 0.f : d[j];
        }
      }
      // DROP: one Philox call per lane and block -- registers 4g..4g+3 are query rows qb0 + 8g + 4h + 0..3 of key kw0 + r,
      // i.e. byte (key & 3) of the four words of patch g, and the quad's four lanes (four consecutive keys) need the same
      // four patches: lane j generates patch g = j (fa_common.h quad_bcast).  Issued here, beside the MFMA chains.
      u32x4 mine = {0, 0, 0, 0};
      if constexpr (DROP) {
        const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
        mine = dropout_patch(dr, ((qb0 + 4 * h) >> 2) + 2 * (r & 3), (kw0 + r) >> 2, b_ * p.H + h_);
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(qbp + row_off[ks]));
        sacc = T::mfma(a, kf[ks], sacc);
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(dbp + row_off[ks]));
        pacc = T::mfma(a, vf[ks], pacc);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? sacc[i] : __builtin_fmaf(sacc[i], c2, nl[i]);
        if constexpr (MASKED) {
          const int qrow = qb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          x = (kw0 + r > qrow) ? -INFINITY : x;
        }
        const float pe = __builtin_amdgcn_exp2f(x);
        sacc[i] = pe;             // P
        if constexpr (!DROP) pacc[i] = pe * pacc[i];   // dS = P o (dP - delta)
      }
      if constexpr (DROP) {
        const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
        const int key = kw0 + r;
        auto apply = [&](auto g_tag) __attribute__((always_inline)) {
          constexpr int g = decltype(g_tag)::value;
          const u32x4 patch = quad_bcast4<g>(mine);
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // register 4g + j <-> query 4*qg + j: word j of the patch, byte key & 3
            const int i = 4 * g + j;
            const bool keep = ((patch[j] >> (8 * (key & 3))) & 255u) >= dr.thresh;
            const float pe = sacc[i];
            // dS = P o (dP - delta) with dP = mask / (1 - p) o (dO V^T): one fma, one select, one multiply
            const float t = __builtin_fmaf(pacc[i], dr.rp, nd[i]);
            pacc[i] = pe * (keep ? t : nd[i]);
            sacc[i] = keep ? pe : 0.f;                         // dropped P for dV (its 1 / (1 - p) is applied once, to dV)
          }
        };
        apply(std::integral_constant<int, 0>{});
        apply(std::integral_constant<int, 1>{});
        apply(std::integral_constant<int, 2>{});
        apply(std::integral_constant<int, 3>{});
      }
      const vec8 p0 = pack8<T, 0>(sacc), p1 = pack8<T, 1>(sacc);
      const vec8 s0 = pack8<T, 0>(pacc), s1 = pack8<T, 1>(pacc);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        vec8 a0 = lds_read_tr_frag<T>(dbp + tr_off[0][db], dbp + tr_off[1][db]);
        dvacc[db] = T::mfma(a0, p0, dvacc[db]);
        vec8 a1 = lds_read_tr_frag<T>(dbp + 16 * C::ROWB + tr_off[0][db], dbp + 16 * C::ROWB + tr_off[1][db]);
        dvacc[db] = T::mfma(a1, p1, dvacc[db]);
      }
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        vec8 a0 = lds_read_tr_frag<T>(qbp + tr_off[0][db], qbp + tr_off[1][db]);
        dkacc[db] = T::mfma(a0, s0, dkacc[db]);
        vec8 a1 = lds_read_tr_frag<T>(qbp + 16 * C::ROWB + tr_off[0][db], qbp + 16 * C::ROWB + tr_off[1][db]);
        dkacc[db] = T::mfma(a1, s1, dkacc[db]);
      }
    }
  };

  if (Sq % C::BQ != 0) {  // a ragged last query tile must not expose uninitialised LDS
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }
  if (t_start < ntiles) {
    stage_load(t_start);
    stage_write(t_start);
  }
  __syncthreads();
  int t = t_start;
  const int t_masked_end = min(ntiles, t_full);
  for (; t < t_masked_end; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) stage_load(t + 1);
    tile(t, std::true_type{});
    if (more) stage_write(t + 1);
    __syncthreads();
  }
  for (; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) stage_load(t + 1);
    tile(t, std::false_type{});
    if (more) stage_write(t + 1);
    __syncthreads();
  }

  FA_LDS char* stage = smem + wave * 32 * C::ROWB;
  // dK = dS^T Q * scale; with the pre-scaled Q (= Q * scale * log2e) in LDS that is dS^T Q' * ln 2
    store_tile_rows<D, T>(dkacc, (FOLD && p.q_prescaled) ? kLn2 : p.scale, stage, rdk, kw0 * dk_rs, lane, dk_rs);
  store_tile_rows<D, T>(dvacc, DROP ? p.drop.rp : 1.0f, stage, rdv, kw0 * dv_rs, lane, dv_rs);
  }  // pass
}

template <int D, typename T, bool CAUSAL, bool DROP = false>
static hipError_t launch(const BwdParams& p, hipStream_t s) {
  using C = DkvCfg<D>;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  auto kern = fa_bwd_dkv_kernel<D, T, CAUSAL, DROP>;
  if (C::LDS_BYTES > 48 * 1024) {
    static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
    if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dkv_v2(BwdParams p, int D, int dtype, int causal, hipStream_t s);  // fa_bwd_dkv_v2.hip
hipError_t launch_bwd_dkv_v3(BwdParams p, int dtype, int causal, hipStream_t s);         // fa_bwd_dkv_v3.hip
hipError_t launch_bwd_dkv_v4(BwdParams p, int dtype, int causal, hipStream_t s);         // fa_bwd_dkv_v4.hip

hipError_t launch_bwd_dkv(BwdParams p, int D, int dtype, int causal, hipStream_t s) {
  const int impl = p.drop.thresh ? 1 : pick_dkv_impl(g_force_dkv, D, dtype, p.B, p.H, p.Sq, p.Sk, causal != 0, !p.vl.cu_q);
  if (impl == 4) return launch_bwd_dkv_v4(p, dtype, causal, s);
  if (impl == 2) return launch_bwd_dkv_v2(p, D, dtype, causal, s);
  if (impl == 3) return launch_bwd_dkv_v3(p, dtype, causal, s);
  p.n_tiles = (p.Sk + 127) / 128;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
#define FA_GO(DD, TT)                                                                           \
  (p.drop.thresh ? (causal ? launch<DD, TT, true, true>(p, s) : launch<DD, TT, false, true>(p, s)) \
                 : (causal ? launch<DD, TT, true>(p, s) : launch<DD, TT, false>(p, s)))
  if (D == 64) return dtype == 1 ? FA_GO(64, BF16) : FA_GO(64, FP16);
  if (D == 128) return dtype == 1 ? FA_GO(128, BF16) : FA_GO(128, FP16);
#undef FA_GO
  return hipErrorInvalidValue;
}

}  // namespace fa
