// FlashAttention backward, query-tile-stationary half: dQ and delta = rowsum(dO * O).
//
// Replaces the reference's flash_attention_dQ_kernel
// (code/_flash_attention_kernel_optimized.py:165-258).  Semantics kept: delta from the
// ROUNDED 16-bit O in fp32 (K:210-211) and stored for the dK/dV kernel (K:258);
// P recomputed from the forward's LSE (K:244); dS rounded to the input dtype before
// dS @ K (K:253); dQ cast on store (K:256).  (scale is applied once to the fp32
// accumulator instead of to every partial product.)
//
// Decomposition: workgroup = 4 waves = 128 query rows, wave = 32 rows; K and V stream
// through LDS in 64-key tiles exactly as in the forward.  Both score-shaped products are
// computed transposed with the query on the lane,
//     S^T  = K Q^T          (A = K rows from LDS,  B = Q^T  resident in registers)
//     dP^T = V dO^T - delta (A = V rows from LDS,  B = dO^T resident; -delta[q] is the
//                            accumulator's initial value: q is the lane, so it is one
//                            broadcast register tuple and costs no VALU per tile)
// so LSE / delta are per-lane scalars and dS^T = P^T o dP^T, rounded, is directly the
// B operand of dQ^T += K^T dS^T (K^T fetched from the same LDS image by
// ds_read_b64_tr_b16).
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

template <int D>
struct DqCfg {
  static constexpr int BM = 128, BN = 64, NT = 256;
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BN * ROWB;
  static constexpr int DMA_PER_MAT = TILE_BYTES / (4 * 1024);  // 1-KiB LDS-DMA instructions per wave per matrix
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;
};

// OCC = workgroups per CU the register allocation is held to (2: 256 VGPRs, 3: 168).
// DROP: attention dropout (fa_common.h `Dropout`): dP = mask / (1 - p) o (dO V^T), so the dP chain starts from zero and
// the mask, the rescale and -delta are applied per element before dS = P o (dP - delta).
template <int D, typename T, bool CAUSAL, int OCC, bool DROP = false>
__global__ __launch_bounds__(256, OCC) void fa_bwd_dq_kernel(BwdParams p) {
  using C = DqCfg<D>;
  using vec8 = typename T::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // causal: a workgroup takes the query-tile pair (nq-1-i, i) -> equal work everywhere (see fa_fwd.hip)
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
  const int nq = (Sq + C::BM - 1) / C::BM;
  if (idx >= (paired ? (nq + 1) / 2 : nq)) return;
  const int npass = (paired && idx != nq - 1 - idx) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  // lane coordinates re-derived per pass (fa_common.h lane_id_now): nothing lane-dependent stays live across passes
  const int lane = lane_id_now(), tid = wave * 64 + lane, r = lane & 31, h = lane >> 5;
  const int qt = paired ? (pass == 0 ? nq - 1 - idx : idx) : (CAUSAL ? nq - 1 - idx : idx);  // heavy first
  const int q0_wg = qt * C::BM;
  const int qw0 = q0_wg + wave * 32;
  if (pass) __syncthreads();  // the previous pass staged its dQ tile in the K/V buffers

  // Q, K, V, dO may be strided views with a contiguous head dim (fa_fwd.hip); O and dQ carry their own layouts
  // (contiguous for the reference's launch, packed rows for varlen); LSE / delta rows of one (batch, head) are contiguous
  const int q_rs = p.lq.rs, do_rs = p.ldo.rs, kv_rs = p.lk.rs, o_rs = p.lo.rs, dq_rs = p.ldq.rs;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(
      (const char*)p.q + b_ * p.lq.sb + h_ * p.lq.sh + (long long)si.q0 * q_rs, (unsigned)(Sq - 1) * q_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(
      (const char*)p.dout + b_ * p.ldo.sb + h_ * p.ldo.sh + (long long)si.q0 * do_rs, (unsigned)(Sq - 1) * do_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(
      (const char*)p.o + b_ * p.lo.sb + h_ * p.lo.sh + (long long)si.q0 * o_rs, (unsigned)(Sq - 1) * o_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rdq = make_rsrc(
      (char*)p.dq + b_ * p.ldq.sb + h_ * p.ldq.sh + (long long)si.q0 * dq_rs, (unsigned)(Sq - 1) * dq_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(
      (const char*)p.k + b_ * p.lk.sb + h_ * p.lk.sh + (long long)si.k0 * kv_rs, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(
      (const char*)p.v + b_ * p.lv.sb + h_ * p.lv.sh + (long long)si.k0 * kv_rs, view_bytes(Sk, kv_rs, C::ROWB));
  const long long rowc_off = b_ * p.lse_sb + h_ * p.lse_sh + si.q0;
  const __amdgpu_buffer_rsrc_t rl = make_rsrc(p.lse + rowc_off, (unsigned)Sq * 4);
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.delta + rowc_off, (unsigned)Sq * 4);


  // ---- resident B operands: Q^T and dO^T of this wave's 32 rows; delta ----
  vec8 qf[C::KS], dof[C::KS];
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int col = (2 * ks + h) * 16;
    qf[ks] = as_vec8<T>(buf_load16(rq, (qw0 + r) * q_rs + col));
    dof[ks] = as_vec8<T>(buf_load16(rdo, (qw0 + r) * do_rs + col));
    const vec8 of = as_vec8<T>(buf_load16(ro, (qw0 + r) * o_rs + col));
#pragma unroll
    for (int j = 0; j < 8; ++j) dsum = __builtin_fmaf((float)dof[ks][j], (float)of[j], dsum);
  }
  const float delta = half_sum(dsum);
  const float nl = -buf_load_f32(rl, (qw0 + r) * 4) * kLog2e;
  if (h == 0) buf_store_f32(rd, (qw0 + r) * 4, delta);
  // Both MFMA chains START from a block holding this lane's (= query row's) constant: with Q pre-scaled by
  // softmax_scale*log2(e) the first delivers the exponent argument s*c2 - LSE*log2e, the second dP - delta,
  // and no per-element fma / subtract is left in the hot loop.
  f32x16 ndelta, nlse;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    ndelta[i] = -delta;
    nlse[i] = T::kFoldScale ? nl : 0.f;
  }
  const float c2 = p.scale * kLog2e;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h
  if constexpr (FOLD) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = scale_frag<T>(qf[ks], c2);
  }

  const int kv_end = CAUSAL ? min(Sk, q0_wg + C::BM) : Sk;
  const int ntiles = (kv_end + C::BN - 1) / C::BN;
  const int nfull = CAUSAL ? min(Sk / C::BN, qw0 / C::BN) : Sk / C::BN;

  // LDS-DMA source offsets (see fa_fwd.hip): wave w fills rows [16w, 16w+16) of each tile
  constexpr int RPI = 1024 / C::ROWB;
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + RPI * i + lane / C::CPR;
    dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16;
#ifndef FA_DMA_LEGACY
    dma_src[i] -= 1024 * i;  // dma_pieces: the immediate offset of piece i also moves the global address
#endif
  }
  int row_off[C::KS];  // A-operand row reads (K rows and V rows)
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
  int tr_off[2][C::DB];  // transposed reads of K
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) tr_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);

  f32x16 dqacc[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) dqacc[db][i] = 0.f;

  auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int soff = t * C::BN * kv_rs;
#ifndef FA_DMA_LEGACY
    const int dst0 = buf * C::TILE_BYTES + 16 * wave * C::ROWB;  // this wave's 16 rows = DMA_PER_MAT consecutive KiB
    dma_pieces<C::DMA_PER_MAT>(rk, lds_addr_of(smem + dst0), dma_src, soff);
    dma_pieces<C::DMA_PER_MAT>(rv, lds_addr_of(smem + 2 * C::TILE_BYTES + dst0), dma_src, soff);
    return;
#endif
#pragma unroll
    for (int i = 0; i < C::DMA_PER_MAT; ++i) {
      const int dst = buf * C::TILE_BYTES + (16 * wave + RPI * i) * C::ROWB;
      dma16(rk, lds_addr_of(smem + dst), dma_src[i], soff);
      dma16(rv, lds_addr_of(smem + 2 * C::TILE_BYTES + dst), dma_src[i], soff);
    }
  };
  auto tile_sync = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    // vmcnt(0): the tile fetched during this step has landed.  lgkmcnt(0): every LDS read this wave has ISSUED on the
    // current tile has also RETURNED -- hipcc is free to sink the wait + MFMA of the last fragment below the barrier,
    // and a read still queued in the LDS pipeline then races the other waves' next DMA / epilogue staging into the
    // same buffer (seen as a rare wrong 32x32 block of one wave once three workgroups shared a CU).
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // BUF: the tile's ring buffer as a compile-time constant (0 / 1), or -1 = t & 1 at run time.  With a constant buffer every
  // LDS address of the tile is `per-lane offset + immediate`; with t & 1 hipcc rebuilds them with vector adds per read
  // (D = 128: 3.6 vector instructions per MFMA where the maths needs 1.7, profiles/r04_pmc_summary_d128.txt).
  auto tile = [&](int t, auto masked_tag, auto buf_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int BUF = decltype(buf_tag)::value;
    const int bsel = BUF >= 0 ? BUF : (t & 1);
    const FA_LDS char* kt = smem + bsel * C::TILE_BYTES;
    const FA_LDS char* vt = smem + (2 + bsel) * C::TILE_BYTES;
    const int s0 = t * C::BN;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if constexpr (MASKED) {
        bool use = s0 + 32 * b < Sk;
        if (CAUSAL) use = use && (s0 + 32 * b <= qw0);
        if (!use) continue;
      }
      const FA_LDS char* kbp = kt + b * 32 * C::ROWB;
      const FA_LDS char* vbp = vt + b * 32 * C::ROWB;
      f32x16 sacc = nlse, pacc = ndelta;
      if constexpr (DROP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0.f;
      }
      // DROP: one Philox call per lane and block -- lane j of a quad generates patch g = j for the quad's four rows
      // (fa_common.h quad_bcast); issued here so that its ~100 integer ops run beside the MFMA chains below
      u32x4 mine = {0, 0, 0, 0};
      if constexpr (DROP) {
        const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
        mine = dropout_patch(dr, (qw0 + r) >> 2, ((s0 + 32 * b + 4 * h) >> 2) + 2 * (r & 3), b_ * p.H + h_);
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(kbp + row_off[ks]));
        sacc = T::mfma(a, qf[ks], sacc);
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(vbp + row_off[ks]));
        pacc = T::mfma(a, dof[ks], pacc);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? sacc[i] : __builtin_fmaf(sacc[i], c2, nl);
        if constexpr (MASKED) {
          const int key = s0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool dead = (CAUSAL && key > qw0 + r) || key >= Sk;
          x = dead ? -INFINITY : x;
        }
        if constexpr (!DROP) sacc[i] = __builtin_amdgcn_exp2f(x) * pacc[i];  // dS^T = P^T o (dP^T - delta)
        else sacc[i] = __builtin_amdgcn_exp2f(x);                             // P^T; the mask comes next
      }
      if constexpr (DROP) {
        const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
        const int qrow = qw0 + r;
        auto apply = [&](auto g_tag) __attribute__((always_inline)) {
          constexpr int g = decltype(g_tag)::value;
          const unsigned w = select_word(quad_bcast4<g>(mine), qrow & 3);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 4 * g + j;
            const bool keep = ((w >> (8 * j)) & 255u) >= dr.thresh;
            // dS = P o (dP - delta) with dP = mask / (1 - p) o (dO V^T): one fma, one select, one multiply
            const float t = __builtin_fmaf(pacc[i], dr.rp, -delta);
            sacc[i] = sacc[i] * (keep ? t : -delta);
          }
        };
        apply(std::integral_constant<int, 0>{});
        apply(std::integral_constant<int, 1>{});
        apply(std::integral_constant<int, 2>{});
        apply(std::integral_constant<int, 3>{});
      }
      const vec8 d0 = pack8<T, 0>(sacc);
      const vec8 d1 = pack8<T, 1>(sacc);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        vec8 a0 = lds_read_tr_frag<T>(kbp + tr_off[0][db], kbp + tr_off[1][db]);
        dqacc[db] = T::mfma(a0, d0, dqacc[db]);
        vec8 a1 = lds_read_tr_frag<T>(kbp + 16 * C::ROWB + tr_off[0][db], kbp + 16 * C::ROWB + tr_off[1][db]);
        dqacc[db] = T::mfma(a1, d1, dqacc[db]);
      }
    }
  };

  if (Sk % C::BN != 0) {  // a ragged last tile must not expose uninitialised LDS
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }
  dma_tile(0, 0);
  tile_sync();
  using BR = std::integral_constant<int, -1>;
  int t = 0;
  if constexpr (D == 128 && !DROP) {   // two tiles per trip, constant buffers (the first tile of a pass is tile 0: buffer 0)
    for (; t + 2 <= nfull; t += 2) {
      dma_tile(t + 1, 1);
      tile(t, std::false_type{}, std::integral_constant<int, 0>{});
      tile_sync();
      if (t + 2 < ntiles) dma_tile(t + 2, 0);
      tile(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
      tile_sync();
    }
  }
  for (; t < nfull; ++t) {
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) & 1);
    tile(t, std::false_type{}, BR{});
    tile_sync();
  }
  for (; t < ntiles; ++t) {
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) & 1);
    tile(t, std::true_type{}, BR{});
    tile_sync();
  }

  if constexpr (FOLD) {
    if (p.qs) {  // workspace for the dK/dV launch: the scaled rows exactly as this kernel (and the forward) multiplied
                 // them; stored here, not in the prologue, where the first tile's vmcnt(0) would wait for them
      const __amdgpu_buffer_rsrc_t rqs = make_rsrc(
          (char*)p.qs + b_ * p.lqs.sb + h_ * p.lqs.sh + (long long)si.q0 * p.lqs.rs, (unsigned)(Sq - 1) * p.lqs.rs + C::ROWB);
      const int ln = lane_id_now();  // re-derived: nothing lane-dependent is kept live across the tile loop for this
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks)
        buf_store16(rqs, (qw0 + (ln & 31)) * p.lqs.rs + (2 * ks + (ln >> 5)) * 16, __builtin_bit_cast(u32x4, qf[ks]));
    }
  }
  store_tile_rows<D, T>(dqacc, p.scale, smem + wave * 32 * C::ROWB, rdq, qw0 * dq_rs, lane, dq_rs);
  }  // pass
}

template <int D, typename T, bool CAUSAL, int OCC = 2, bool DROP = false>
static hipError_t launch(const BwdParams& p, hipStream_t s) {
  using C = DqCfg<D>;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  // Three workgroups per CU pay off for the bf16 kernel (no fma/sub in its hot loop, fa_common.h kFoldScale) once
  // the grid fills them: +3 % causal, +7 % non-causal at B4 H32 N4096.  The tighter register budget spills in
  // the prologue only, which costs small grids more than the occupancy gives (B4 H8 S1024: -16 %); fp16: no gain.
  if constexpr (OCC == 2 && D == 64 && T::kFoldScale && !DROP) {
    if (grid >= 3 * 256) return launch<D, T, CAUSAL, 3>(p, s);
  }
  auto kern = fa_bwd_dq_kernel<D, T, CAUSAL, OCC, DROP>;
  if (C::LDS_BYTES > 48 * 1024) {
    static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
    if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dq_v2(BwdParams p, int dtype, int causal, hipStream_t s);  // fa_bwd_dq_v2.hip
hipError_t launch_bwd_dq_v3(BwdParams p, int dtype, int causal, hipStream_t s);  // fa_bwd_dq_v3.hip
hipError_t launch_bwd_dq_v4(BwdParams p, int dtype, int causal, hipStream_t s);  // fa_bwd_dq_v4.hip

hipError_t launch_bwd_dq(BwdParams p, int D, int dtype, int causal, hipStream_t s) {
  const int impl = p.drop.thresh ? 1 : pick_dq_impl(g_force_dq, D, dtype, p.B, p.H, p.Sq, p.Sk, causal != 0, p.all_contiguous(D), !p.vl.cu_q);
  if (impl == 4) return launch_bwd_dq_v4(p, dtype, causal, s);
  if (impl == 3) return launch_bwd_dq_v3(p, dtype, causal, s);
  if (impl == 2) return launch_bwd_dq_v2(p, dtype, causal, s);
  p.n_tiles = (p.Sq + 127) / 128;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
#define FA_GO(DD, TT)                                                                                 \
  (p.drop.thresh ? (causal ? launch<DD, TT, true, 2, true>(p, s) : launch<DD, TT, false, 2, true>(p, s)) \
                 : (causal ? launch<DD, TT, true>(p, s) : launch<DD, TT, false>(p, s)))
  if (D == 64) return dtype == 1 ? FA_GO(64, BF16) : FA_GO(64, FP16);
  if (D == 128) return dtype == 1 ? FA_GO(128, BF16) : FA_GO(128, FP16);
#undef FA_GO
  return hipErrorInvalidValue;
}

}  // namespace fa
