// FlashAttention backward dK / dV, third schedule family for gfx950 (head dim 64): ONE wave per SIMD, 64 keys per wave.
//
// Same maths and rounding points as fa_bwd_dkv.hip / fa_bwd_dkv_v2.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:292-386; runs after the dQ kernel and reads its delta).
//
// Why a third family: family 2 (two waves per SIMD, 32 keys per wave) reads one 1-KiB LDS fragment per MFMA
// (SQ_INSTS_LDS / SQ_INSTS_MFMA = 2.04).  Here a workgroup is 256 keys, a wave owns TWO 32-key groups and every Q / dO
// fragment (row fragment for S and dP, transposed fragment for dV^T and dK^T) is read from LDS once and feeds both
// groups: half the LDS bytes per MFMA.  The price is registers: dK^T and dV^T of 64 keys (128), resident K^T / V^T
// fragments (64), the fragments held for the second group (64) and two blocks of S / dP in flight (64+) are more than
// 256, so a SIMD holds one wave and nothing but the wave's own instruction order hides latencies -- the whole unmasked
// loop is therefore one continuous, hand-ordered software pipeline (no fill / drain per tile):
//
//   block iteration i = (query block qb = i / 2, key group kg = i % 2) of a 128-row Q/dO tile, 16 MFMA slots:
//     slots 0-3   S  = Q K^T          of block i     (chain starts from the row constants -LSE*log2e: C operand != D)
//     slots 4-7   dP = dO V^T         of block i     (chain starts from -delta)
//     slots 8-11  dV^T += dO^T P      of block i-1
//     slots 12-15 dK^T += Q^T dS      of block i-1
//   and beside the MFMAs, per slot: two exp + one pack of block i-1 (slots 0-7), three dS multiplies + packs (slots
//   8-13), the LDS reads of fragments and row constants whose registers have just become free (each is read >= 4 slots before
//   its first use and kept for its second use 16 slots later), one LDS-DMA piece of the next tile in some dK slots,
//   closed by sched_barrier(0) so that hipcc keeps exactly this order.
//   The per-tile commit (vmcnt(0), row constants -> LDS, s_barrier) sits INSIDE iteration 7, after the last read of the
//   current tile's buffer and before the first read of the next one: two LDS buffers, no pipeline bubble.
//
// Causal: a workgroup takes the key-tile pair (i, nk-1-i).  A wave owns key groups {w, 7-w} of the 256 keys, so that the
// diagonal region (the 256 query rows level with the key tile) costs every wave the same 9 of 16 block visits; those two
// tiles run block by block on a simple compiler-scheduled path, everything below the diagonal in the pipeline.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dkv3Cfg {
  static constexpr int D = 64;
  static constexpr int BK = 256, BQ = 128, NT = 256, NW = 4;
  static constexpr int QB = BQ / 32;                       // 32-row query blocks per tile
  static constexpr int NI = 2 * QB;                        // block iterations per tile (query block x key group)
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BQ * ROWB;             // 16 KiB
  static constexpr int NBUF = 2;
  static constexpr int DO_BASE = NBUF * TILE_BYTES;        // Q[NBUF], then dO[NBUF]
  static constexpr int ROWC_OFF = 2 * NBUF * TILE_BYTES;   // then row constants: nl[BQ], nd[BQ] per buffer
  static constexpr int ROWC_BYTES = 2 * BQ * 4;
  static constexpr int LDS_BYTES = 2 * NBUF * TILE_BYTES + NBUF * ROWC_BYTES;  // 66 KiB
  static constexpr int DMA_PER_MAT = TILE_BYTES / (NW * 1024);                 // 4 pieces of Q and 4 of dO per wave
  static constexpr int RPI = 1024 / ROWB;                  // tile rows per 1-KiB DMA piece
};

// where the 8 LDS-DMA pieces of the NEXT tile (0-3 Q, 4-7 dO rows of this wave's 32-row share) are issued: block
// iteration and slot.  dK slots carry the least VALU work.  A/B hooks: -DFA_DKV3_DMA_ITERS / _SLOTS.
#ifndef FA_DKV3_DMA_ITERS
#define FA_DKV3_DMA_ITERS {0, 0, 1, 1, 2, 2, 3, 3}
#endif
#ifndef FA_DKV3_DMA_SLOTS
#define FA_DKV3_DMA_SLOTS {13, 15, 13, 15, 13, 15, 13, 15}
#endif
constexpr int kDkv3DmaIter[8] = FA_DKV3_DMA_ITERS;
constexpr int kDkv3DmaSlot[8] = FA_DKV3_DMA_SLOTS;

#ifdef FA_STAMPS
#define FA3_STAMP(slot)                                                           \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long now_;                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += now_ - last_;                                                    \
    last_ = now_;                                                                 \
  } while (0)
#else
#define FA3_STAMP(slot) do {} while (0)
#endif

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void fa_bwd_dkv3_kernel(BwdParams p) {
  using C = Dkv3Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
#ifdef FA_STAMPS
  unsigned long long clk0_, rt0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_)::"memory");
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0, nblk_ = 0;
#endif
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
  const BatchHead ix = batch_head(bh, p.B, p.H, p.vl.cu_q != nullptr);
  const int b_ = ix.b, h_ = ix.h;
  const SeqInfo si = seq_info(p.vl, b_, p.Sq, p.Sk);
  const int Sq = si.Sq, Sk = si.Sk;
  const int nk = (Sk + C::BK - 1) / C::BK;
  if (idx >= (paired ? (nk + 1) / 2 : nk)) return;
  const int npass = (paired && idx != nk - 1 - idx) ? 2 : 1;

  const int q_rs = p.lq.rs, do_rs = p.ldo.rs, kv_rs = p.lk.rs, dk_rs = p.ldk.rs, dv_rs = p.ldv.rs;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(
      (const char*)p.q + b_ * p.lq.sb + h_ * p.lq.sh + (long long)si.q0 * q_rs, view_bytes(Sq, q_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(
      (const char*)p.dout + b_ * p.ldo.sb + h_ * p.ldo.sh + (long long)si.q0 * do_rs, view_bytes(Sq, do_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(
      (const char*)p.k + b_ * p.lk.sb + h_ * p.lk.sh + (long long)si.k0 * kv_rs, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(
      (const char*)p.v + b_ * p.lv.sb + h_ * p.lv.sh + (long long)si.k0 * kv_rs, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rdk = make_rsrc(
      (char*)p.dk + b_ * p.ldk.sb + h_ * p.ldk.sh + (long long)si.k0 * dk_rs, view_bytes(Sk, dk_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rdv = make_rsrc(
      (char*)p.dv + b_ * p.ldv.sb + h_ * p.ldv.sh + (long long)si.k0 * dv_rs, view_bytes(Sk, dv_rs, C::ROWB));
  const long long rowc_off = b_ * p.lse_sb + h_ * p.lse_sh + si.q0;
  // row constants of a query tile: waves 0-1 load its LSE rows, waves 2-3 its delta rows, through ONE wave-uniform
  // descriptor and an unconditional load (fa_bwd_dkv_v2.hip: a divergent `if` around it costs a hidden vmcnt(0))
  const bool rc_lse = wave < C::BQ / 64;
  const __amdgpu_buffer_rsrc_t rrc = make_rsrc((rc_lse ? p.lse : p.delta) + rowc_off, (unsigned)Sq * 4);
  auto rc_row_now = [&]() __attribute__((always_inline)) -> int {
    int x;
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "n"(C::BQ - 1), "v"(tid));
    return x;
  };

  // ---- loop-invariant per-lane addresses ----
  int dma_q[C::DMA_PER_MAT], dma_do[C::DMA_PER_MAT];   // per-lane global source offsets of this wave's DMA pieces
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = (C::BQ / C::NW) * wave + C::RPI * i + lane / C::CPR;
    const int chunk = swz_chunk<D>(row, lane % C::CPR) * 16;
    dma_q[i] = row * q_rs + chunk;
    dma_do[i] = row * do_rs + chunk;
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
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h: the score chain starts from -LSE*log2e and K (or Q') carries c2
  const int ntiles = (Sq + C::BQ - 1) / C::BQ;

  if (Sq % C::BQ != 0) {  // a ragged last query tile leaves its tail rows to an out-of-range DMA: keep LDS finite
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }

  for (int pass = 0; pass < npass; ++pass) {
    const int kt_idx = paired ? (pass == 0 ? idx : nk - 1 - idx) : idx;  // low key tiles are the heavy ones
    const int k0_wg = kt_idx * C::BK;
    // this wave's two 32-key groups: {w, 7-w} of the workgroup's eight (equal causal work per wave, see the header)
    const int kw[2] = {k0_wg + 32 * wave, k0_wg + 32 * (7 - wave)};
    if (pass) __syncthreads();  // the previous pass staged dK / dV in the tile buffers

    const int t_start = CAUSAL ? k0_wg / C::BQ : 0;
    const int t_diag_end = CAUSAL ? min(ntiles, t_start + C::BK / C::BQ) : 0;  // tiles level with the key tile

    // ---- DMA of one Q/dO tile + the row-constant load (one float per thread) ----
    float rc = 0.f;
    auto dma_piece = [&](int t, int buf, int j) __attribute__((always_inline)) {  // j: 0-3 Q, 4-7 dO
      const int i = j & 3;
      const int dst = buf * C::TILE_BYTES + ((C::BQ / C::NW) * wave + C::RPI * i) * C::ROWB;
      if (j < 4) dma_pieces<1>(rq, lds_addr_of(smem + dst), dma_q + i, t * C::BQ * q_rs);
      else dma_pieces<1>(rdo, lds_addr_of(smem + C::DO_BASE + dst), dma_do + i, t * C::BQ * do_rs);
    };
    auto fetch_rc = [&](int t) __attribute__((always_inline)) { rc = buf_load_f32(rrc, (t * C::BQ + rc_row_now()) * 4); };
    auto fetch_tile = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(t, buf, j);
      fetch_rc(t);
    };
    // everything of the fetched tile has landed (vmcnt(0)): publish the scaled row constants, then meet
    auto commit_tile = [&](int t, int buf, bool fetched) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      if (fetched) {
        FA_LDS float* rcp = (FA_LDS float*)(smem + C::ROWC_OFF + buf * C::ROWC_BYTES);
        // rows past S_q must give P = 0 (K:355-356): exp2(-inf) = 0
        const float lse_c = (t * C::BQ + rc_row_now() < Sq) ? -rc * kLog2e : -INFINITY;
        rcp[tid] = rc_lse ? lse_c : -rc;  // rcp[row] = -LSE*log2e, rcp[BQ + row] = -delta
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the ds_write above and every LDS read issued so far
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };

    // ---- resident B operands: K^T and V^T of this wave's two key groups ----
    if (t_start < ntiles) fetch_tile(t_start, t_start & 1);
    // (kept in accumulator registers: B operands of the VGPR-accumulator MFMA forms, fa_common.h mfma_v_*)
    agpr4_t kf[2][C::KS], vf[2][C::KS];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (kw[g] + r) * kv_rs + (2 * ks + h) * 16;
        vec8 kk = as_vec8<T>(buf_load16(rk, off));
        if (FOLD && !p.q_prescaled) kk = scale_frag<T>(kk, c2);  // K * softmax_scale * log2(e)
        kf[g][ks] = to_agpr(__builtin_bit_cast(u32x4, kk));
        vf[g][ks] = to_agpr(buf_load16(rv, off));
      }
    asm volatile("s_nop 4");  // v_accvgpr_write -> MFMA operand wait states (hipcc pads nothing around asm)
    f32x16 dkacc[2][C::DB], dvacc[2][C::DB];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dkacc[g][db][i] = 0.f;
          dvacc[g][db][i] = 0.f;
        }

    // ---- simple path (diagonal region): one 32-row query block x one 32-key group, compiler-scheduled ----
    auto q_block = [&](int buf, int b, int g, int qb0, auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      const FA_LDS char* qbp = smem + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* dbp = smem + C::DO_BASE + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
      f32x16 nl, pacc, sacc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // per-register row constants: reg i <-> row (i&3) + 8(i>>2) + 4h
        const f32x4 a = *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * q + 4 * h) * 4);
        const f32x4 d = *(const FA_LDS f32x4*)(rcp + (C::BQ + 32 * b + 8 * q + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nl[4 * q + j] = a[j];
          sacc[4 * q + j] = FOLD ? a[j] : 0.f;
          pacc[4 * q + j] = d[j];
        }
      }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) T::mfma_v_acc(sacc, lds_read16(qbp + row_off[ks]), kf[g][ks]);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) T::mfma_v_acc(pacc, lds_read16(dbp + row_off[ks]), vf[g][ks]);
      // asm MFMA results -> VALU readers: 12 wait states that hipcc does not insert for an asm statement
      asm volatile("s_nop 15" : "+v"(sacc), "+v"(pacc));
      vec8 dof[C::DB][2], qtf[C::DB][2];
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          dof[db][e] = lds_read_tr_frag<T>(dbp + 16 * e * C::ROWB + tr_off[0][db], dbp + 16 * e * C::ROWB + tr_off[1][db]);
          qtf[db][e] = lds_read_tr_frag<T>(qbp + 16 * e * C::ROWB + tr_off[0][db], qbp + 16 * e * C::ROWB + tr_off[1][db]);
        }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? sacc[i] : __builtin_fmaf(sacc[i], c2, nl[i]);
        if constexpr (MASKED) {
          const int qrow = qb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          x = (kw[g] + r > qrow) ? -INFINITY : x;
        }
        const float pe = __builtin_amdgcn_exp2f(x);
        sacc[i] = pe;            // P
        pacc[i] = pe * pacc[i];  // dS = P o (dP - delta)
      }
      const vec8 p0 = pack8<T, 0>(sacc), p1 = pack8<T, 1>(sacc);
      const vec8 s0 = pack8<T, 0>(pacc), s1 = pack8<T, 1>(pacc);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        dvacc[g][db] = T::mfma(dof[db][0], p0, dvacc[g][db]);
        dvacc[g][db] = T::mfma(dof[db][1], p1, dvacc[g][db]);
        dkacc[g][db] = T::mfma(qtf[db][0], s0, dkacc[g][db]);
        dkacc[g][db] = T::mfma(qtf[db][1], s1, dkacc[g][db]);
      }
    };
    auto step_simple = [&](int t) __attribute__((always_inline)) {
      const int buf = t & 1;
      const bool more = t + 1 < ntiles;
      if (more) fetch_tile(t + 1, buf ^ 1);
      for (int b = 0; b < C::QB; ++b) {
        const int qb0 = t * C::BQ + 32 * b;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (qb0 < kw[g]) continue;  // every row of the block is above the diagonal of this key group
          if (qb0 == kw[g]) q_block(buf, b, g, qb0, std::true_type{});
          else q_block(buf, b, g, qb0, std::false_type{});
        }
      }
      commit_tile(t + 1, buf ^ 1, more);
    };

    // ---- the pipeline (unmasked tiles): state carried from block to block, tile to tile ----
    f32x16 S_[2], P_[2];   // [key group]: score / dP accumulators of the block in flight; the OTHER set holds the
                           // previous block's exponent arguments -> P and dP - delta -> dS
    f32x16 NL[FOLD ? 1 : 2], ND;      // row constants of a query block ([query block parity] for NL: the exact-fma path (fp16)
                           // needs the previous block's -LSE*log2e while the next block's is being read)
    u32x4 RF[8];           // row fragments of the current query block: Q k-steps 0-3, dO k-steps 0-3
    vec8 TF[8];            // transposed fragments of the previous block's query block: dO^T (db, e) 0-3, Q^T 4-7
    u32x4 pk[2], sk[2];    // packed P and dS of the previous block, k-steps 0 / 1
    auto rowc_read = [&](const FA_LDS char* rcp, int b, int q, f32x16& nl, f32x16& nd) __attribute__((always_inline)) {
      const f32x4 a = *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * q + 4 * h) * 4);
      const f32x4 d = *(const FA_LDS f32x4*)(rcp + (C::BQ + 32 * b + 8 * q + 4 * h) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        nl[4 * q + j] = a[j];
        nd[4 * q + j] = d[j];
      }
    };
    // fill: the first iteration's "previous block" is neutral (P = exp2(-inf) = 0, dS = 0 * 0, fragments 0)
    auto pipe_fill = [&](int buf) __attribute__((always_inline)) {
      const FA_LDS char* qt = smem + buf * C::TILE_BYTES;
      const FA_LDS char* dt = smem + C::DO_BASE + buf * C::TILE_BYTES;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        S_[1][i] = -INFINITY;
        P_[1][i] = 0.f;
        if (!FOLD) NL[FOLD ? 0 : 1][i] = 0.f;
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) TF[n] = as_vec8<T>(u32x4{0u, 0u, 0u, 0u});
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        RF[s] = lds_read16(qt + row_off[s]);
        RF[4 + s] = lds_read16(dt + row_off[s]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) rowc_read(rcp, 0, q, NL[0], ND);
      __builtin_amdgcn_sched_barrier(0);
    };
    // VALU work of the previous block under slot s (X = its exponent arguments -> P, Y = dP - delta -> dS)
    auto prev_valu = [&](int s, f32x16& X, f32x16& Y, const f32x16& nlp) __attribute__((always_inline)) {
      if (s < 8) {
#pragma unroll
        for (int e = 2 * s; e < 2 * s + 2; ++e)
          X[e] = __builtin_amdgcn_exp2f(FOLD ? X[e] : __builtin_fmaf(X[e], c2, nlp[e]));
      }
      if (s >= 1 && s <= 8) {
        const int j = s - 1;
        pk[j >> 2][j & 3] = pack2<T>(X[2 * j], X[2 * j + 1]);
      }
      // dS multiplies: 3 per slot under the dV slots, the last 4 under the first dK slot; each pair is packed as soon as
      // both its products exist (sk[0] = pairs 0-3 is read by slot 12, sk[1] = pairs 4-7 by slot 14)
      constexpr int m0[6] = {0, 3, 6, 9, 12, 16}, m1[6] = {3, 6, 9, 12, 16, 16};
      constexpr int c0[6] = {0, 1, 3, 4, 6, 7}, c1[6] = {1, 3, 4, 6, 7, 8};
      if (s >= 8 && s <= 13) {
#pragma unroll
        for (int e = m0[s - 8]; e < m1[s - 8]; ++e) Y[e] = X[e] * Y[e];
#pragma unroll
        for (int j = c0[s - 8]; j < c1[s - 8]; ++j) sk[j >> 2][j & 3] = pack2<T>(Y[2 * j], Y[2 * j + 1]);
      }
    };
    // one block iteration; I = 2 * query block + key group.  `qt` / `dt` / `rct`: current tile; `qn` / `dn` / `rcn`: next
    // tile (read from iteration 7, slot 4 on -- after the commit).  `hook(I, s, 0)` runs before the slot's MFMA (the
    // commit), `hook(I, s, 1)` right after it (LDS-DMA pieces: their issue then overlaps the MFMA just started).
    auto block_iter = [&](auto i_tag, const FA_LDS char* qt, const FA_LDS char* dt, const FA_LDS char* rct,
                          const FA_LDS char* qn, const FA_LDS char* dn, const FA_LDS char* rcn,
                          auto&& hook) __attribute__((always_inline)) {
      constexpr int I = decltype(i_tag)::value;
      constexpr int qb = I >> 1, g = I & 1, pg = g ^ 1;     // pg: key group of the previous block
      constexpr bool last_qb = qb + 1 == C::QB;
      constexpr int nqb = last_qb ? 0 : qb + 1;              // next query block (block 0 of the next tile after the last)
      const FA_LDS char* qnext = (last_qb ? qn : qt) + nqb * 32 * C::ROWB;
      const FA_LDS char* dnext = (last_qb ? dn : dt) + nqb * 32 * C::ROWB;
      const FA_LDS char* rcnext = last_qb ? rcn : rct;
      const FA_LDS char* qcur = qt + qb * 32 * C::ROWB;
      const FA_LDS char* dcur = dt + qb * 32 * C::ROWB;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        hook(I, s, 0);
        // ---- the MFMA of this slot ----
#ifdef FA_DKV3_BUILTIN_MFMA   // diagnostic: compiler-visible MFMAs for the S / dP chains (results land in AGPRs: slow)
        if (s < 4) {
          const f32x16 zero = {};
          S_[g] = T::mfma(as_vec8<T>(RF[s]), as_vec8<T>(kf[g][s]), s == 0 ? (FOLD ? NL[0] : zero) : S_[g]);
        } else if (s < 8) {
          P_[g] = T::mfma(as_vec8<T>(RF[s]), as_vec8<T>(vf[g][s - 4]), s == 4 ? ND : P_[g]);
        } else if (s < 12) {
#else
        if (s == 0) {
          if constexpr (FOLD) T::mfma_v_first(S_[g], RF[0], kf[g][0], NL[FOLD ? 0 : (qb & 1)]);
          else T::mfma_v_first0(S_[g], RF[0], kf[g][0]);
        } else if (s < 4) {
          T::mfma_v_acc(S_[g], RF[s], kf[g][s]);
        } else if (s == 4) {
          T::mfma_v_first(P_[g], RF[4], vf[g][0], ND);
        } else if (s < 8) {
          T::mfma_v_acc(P_[g], RF[s], vf[g][s - 4]);
        } else if (s < 12) {
#endif   // (k-step e, d block db) = (n >> 1, n & 1): pk[0] is complete first
          const int n = s - 8, e = n >> 1, db = n & 1;
          dvacc[pg][db] = T::mfma(TF[2 * db + e], as_vec8<T>(pk[e]), dvacc[pg][db]);
        } else {
          const int n = s - 12, e = n >> 1, db = n & 1;
          dkacc[pg][db] = T::mfma(TF[4 + 2 * db + e], as_vec8<T>(sk[e]), dkacc[pg][db]);
        }
        hook(I, s, 1);
        // ---- LDS reads into registers that have just become free ----
        if (g == 0) {
          if (s >= 12) {   // dO^T fragments of THIS query block (first used at slot 8 of the next iteration)
            const int n = s - 12, db = n >> 1, e = n & 1;
            TF[n] = lds_read_tr_frag<T>(dcur + 16 * e * C::ROWB + tr_off[0][db], dcur + 16 * e * C::ROWB + tr_off[1][db]);
          }
        } else {
          if (s < 4) {     // Q^T fragments of this query block (first used at slot 12)
            const int n = s, db = n >> 1, e = n & 1;
            TF[4 + n] = lds_read_tr_frag<T>(qcur + 16 * e * C::ROWB + tr_off[0][db], qcur + 16 * e * C::ROWB + tr_off[1][db]);
          } else if (s < 12) {   // row fragments of the NEXT query block (both uses of the old content are over) and
                                 // its row constants (-LSE*log2e last used as C at slot 0, -delta at slot 4): nine and
                                 // more slots ahead of the chains that start from them
            const int f = s - 4;
            RF[f] = lds_read16((f < 4 ? qnext : dnext) + row_off[f & 3]);
            const FA_LDS char* rp = rcnext + ((s < 8 ? 0 : C::BQ) + 32 * nqb + 8 * (f & 3) + 4 * h) * 4;
            const f32x4 v = *(const FA_LDS f32x4*)rp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (s < 8) NL[FOLD ? 0 : ((qb + 1) & 1)][4 * (f & 3) + j] = v[j];
              else ND[4 * (f & 3) + j] = v[j];
            }
          }
        }
        // ---- VALU of the previous block ----
        prev_valu(s, S_[pg], P_[pg], NL[FOLD ? 0 : ((g == 0 ? qb + 1 : qb) & 1)]);
        // An MFMA reads its C operand over its passes: a VALU write to those registers within ~13 wait states corrupts
        // it (hipcc pads this WAR hazard for its own MFMAs, not for an asm statement).  In the second key group's
        // iteration the row constants are DEAD after the chain start that reads them (slot 0 / slot 4) until their
        // reload, so hipcc reuses their registers for exp results at once -- seen as run-to-run differences in dK of
        // the second key group only.  Keep them live for one more slot.
        if (g == 1 && s == 1 && FOLD) keep_live(NL[0]);
        if (g == 1 && s == 5) keep_live(ND);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef FA_STAMPS
      FA3_STAMP(1);
      ++nblk_;
#endif
    };
    // the last block's exp / dS / dV^T / dK^T once nothing follows it
    auto pipe_drain = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s >= 8 && s < 12) {
          const int n = s - 8, e = n >> 1, db = n & 1;
          dvacc[1][db] = T::mfma(TF[2 * db + e], as_vec8<T>(pk[e]), dvacc[1][db]);
        } else if (s >= 12) {
          const int n = s - 12, e = n >> 1, db = n & 1;
          dkacc[1][db] = T::mfma(TF[4 + 2 * db + e], as_vec8<T>(sk[e]), dkacc[1][db]);
        }
        prev_valu(s, S_[1], P_[1], NL[FOLD ? 0 : ((C::QB - 1) & 1)]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto step_pipe = [&](int t, int buf) __attribute__((always_inline)) {
      const bool more = t + 1 < ntiles;
      const FA_LDS char* qt = smem + buf * C::TILE_BYTES;
      const FA_LDS char* dt = smem + C::DO_BASE + buf * C::TILE_BYTES;
      const FA_LDS char* rct = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
      const FA_LDS char* qn = smem + (buf ^ 1) * C::TILE_BYTES;
      const FA_LDS char* dn = smem + C::DO_BASE + (buf ^ 1) * C::TILE_BYTES;
      const FA_LDS char* rcn = smem + C::ROWC_OFF + (buf ^ 1) * C::ROWC_BYTES;
      // No branch anywhere in a tile step: hipcc sinks instructions across a conditional branch into the block that uses
      // their results (the exps of the slots before the branch then run in one burst behind it), which sched_barrier
      // cannot prevent.  Past the last tile the fetches are simply out of range: the buffer descriptors return / write
      // nothing, and the buffer they would have filled is not read again.
      (void)more;
      fetch_rc(t + 1);
      auto hook = [&](int I, int s, int phase) __attribute__((always_inline)) {
        if (phase == 0) {
          if (I == C::NI - 1 && s == 4) {   // every read of this tile's buffers is issued: hand the other buffer over
            FA3_STAMP(2);
            commit_tile(t + 1, buf ^ 1, true);
            FA3_STAMP(3);
          }
          return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (kDkv3DmaIter[j] == I && kDkv3DmaSlot[j] == s) {
            dma_piece(t + 1, buf ^ 1, j);
          }
      };
      block_iter(std::integral_constant<int, 0>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 1>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 2>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 3>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 4>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 5>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 6>{}, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 7>{}, qt, dt, rct, qn, dn, rcn, hook);
    };

#ifdef FA_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    commit_tile(t_start, t_start & 1, t_start < ntiles);  // first tile landed (and the K/V fragments)
    int t = t_start;
    for (; t < t_diag_end; ++t) step_simple(t);
    FA3_STAMP(0);
    if (t < ntiles) {
      pipe_fill(t & 1);
      for (; t < ntiles; ++t) step_pipe(t, t & 1);
      pipe_drain();
    }
    FA3_STAMP(4);

    __syncthreads();  // every wave is done with the tile buffers: they become the staging area
    FA_LDS char* stage = smem + wave * 32 * C::ROWB;
    // dK = dS^T Q * scale; with the pre-scaled Q (= Q * scale * log2e) in LDS that is dS^T Q' * ln 2
    const float dk_mul = (FOLD && p.q_prescaled) ? kLn2 : p.scale;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      store_tile_rows<D, T>(dkacc[g], dk_mul, stage, rdk, kw[g] * dk_rs, lane, dk_rs);
      store_tile_rows<D, T>(dvacc[g], 1.0f, stage, rdv, kw[g] * dv_rs, lane, dv_rs);
    }
    FA3_STAMP(5);
  }  // pass
#ifdef FA_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
    for (int i = 0; i < 8; ++i) d[i] = seg[i];
    d[8] = nblk_;
    unsigned long long clk1_, rt1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_), "=s"(rt1_)::"memory");
    d[9] = clk1_ - clk0_;
    d[10] = rt1_ - rt0_;
  }
#endif
}

template <typename T, bool CAUSAL>
static hipError_t launch3(const BwdParams& p, hipStream_t s) {
  using C = Dkv3Cfg;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  auto kern = fa_bwd_dkv3_kernel<T, CAUSAL>;
  static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
  if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dkv_v3(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sk + Dkv3Cfg::BK - 1) / Dkv3Cfg::BK;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
  if (dtype == 1) return causal ? launch3<BF16, true>(p, s) : launch3<BF16, false>(p, s);
  return causal ? launch3<FP16, true>(p, s) : launch3<FP16, false>(p, s);
}

}  // namespace fa
