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
//   and beside the MFMAs, per slot: ONE exp, one dS multiply and one pack (block i from its slot 5 on, block i-1 until
//   its work is done at slot 10 -- block_valu() below), the LDS reads of fragments and row constants whose registers have just become free (each is read >= 4 slots before
//   its first use and kept for its second use 16 slots later), one LDS-DMA piece of the next tile in some dK slots,
//   closed by sched_barrier(0) so that hipcc keeps exactly this order.
//   The per-tile commit (vmcnt(0), row constants -> LDS, s_barrier) sits INSIDE iteration 7, after the last read of the
//   current tile's buffer and before the first read of the next one: two LDS buffers, no pipeline bubble.
//
// Causal: a workgroup takes the key-tile pair (i, nk-1-i).  A wave owns key groups {w, 7-w} of the 256 keys, so that the
// diagonal region (the 256 query rows level with the key tile) holds the same 9 visible of 16 block visits for every wave;
// those two tiles (and one more) run through the SAME pipeline with a one-compare mask per element -- all 16 visits, the
// invisible ones masked away (family 4, fa_bwd_dkv_v4.hip, visits only the 9).
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

// query rows per Q / dO tile (A/B hook: 128 or 256).  One commit (barrier, row constants, buffer hand-over) per tile; 256
// halves their number, doubles the LDS to 132 KiB -- and measured +0.7 % non-causal / -0.7 % causal (bit-identical): the
// cycles the stamps show next to a tile boundary do not go away with the boundary.  128 stays.
#ifndef FA_DKV3_BQ
#define FA_DKV3_BQ 128
#endif
struct Dkv3Cfg {
  static constexpr int D = 64;
  static constexpr int BK = 256, BQ = FA_DKV3_BQ, NT = 256, NW = 4;
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
  static constexpr int NP = 2 * DMA_PER_MAT;               // pieces per wave and tile: [0, DMA_PER_MAT) Q, then dO
  static constexpr int RCN = 2 * BQ / NT;                  // row constants (LSE and delta rows of a tile) loaded per thread
  static_assert(BQ == 128 || BQ == 256, "one or two row constants per thread");
};

// where the LDS-DMA pieces of a coming tile (first the Q rows, then the dO rows of this wave's share) are issued: block
// iteration and slot.  The commit inside the last iteration of tile t frees tile t's own buffer, so a piece placed in that
// iteration (slot >= 5) already belongs to tile t + 2; pieces in the earlier iterations of tile t + 1 complete that tile.
// Everything is waited for by the next commit (vmcnt(0)), most of a tile time after the first piece.
// (group k = pieces [k G, k G + G): groups 0-1 in the last iteration, slots 9 and 13, then two per iteration from 0 on)
// pieces issued together (one M0 write, consecutive immediates): group k = pieces [k G, k G + G) takes piece k G's place
// FA_DKV3_SPLIT_COMMIT (A/B hook, OFF): publish the row constants four slots before the barrier and wait with lgkmcnt(8)
// instead of lgkmcnt(0).  +0.5-0.8 % -- but the eight reads it leaves in flight across the barrier are the transposed Q
// fragments of THIS tile's buffer, which the other waves' DMA pieces start to overwrite five slots later: safe only by
// timing (the hazard fa_fwd.hip's tile_sync documents), so the product drains them.
#ifndef FA_DKV3_SPLIT_COMMIT
#define FA_DKV3_SPLIT_COMMIT 0
#endif
#ifndef FA_DKV3_DMA_GROUP
#define FA_DKV3_DMA_GROUP 2   // A/B at the headline: 2 per group +0.7 % (non-causal) / +1.2 % (causal) over single pieces, 4 the same
#endif
constexpr int kDkv3DmaGroup = FA_DKV3_DMA_GROUP;
constexpr int dkv3_dma_iter(int j) { return j / kDkv3DmaGroup < 2 ? Dkv3Cfg::NI - 1 : (j / kDkv3DmaGroup - 2) / 2; }
constexpr int dkv3_dma_slot(int j) { return 9 + 4 * ((j / kDkv3DmaGroup) & 1); }

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
  unsigned long long seg[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0, nblk_ = 0;
#ifdef FA_STAMPS_SLOTS
  unsigned long long slot_seg[3][16] = {}, slot_last_ = 0;
#endif
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
  // (BQ = 256: every thread loads one LSE row and one delta row -- two descriptors, still no divergence)
  const bool rc_lse = wave < C::BQ / 64;
  const __amdgpu_buffer_rsrc_t rrc = make_rsrc((rc_lse ? p.lse : p.delta) + rowc_off, (unsigned)Sq * 4);
  const __amdgpu_buffer_rsrc_t rrc_d = make_rsrc(p.delta + rowc_off, (unsigned)Sq * 4);
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
    // (piece i of a group carries the immediate offset 1024 * (i % G), which also moves the global address: taken out here)
    dma_q[i] = row * q_rs + chunk - 1024 * (i % kDkv3DmaGroup);
    dma_do[i] = row * do_rs + chunk - 1024 * (i % kDkv3DmaGroup);
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
    const int t_diag_end = CAUSAL ? min(ntiles, t_start + C::BK / C::BQ + 1) : 0;  // tiles run with the causal mask

    // ---- DMA of one Q/dO tile + the row-constant load (one float per thread) ----
    float rc = 0.f, rc_d = 0.f;
    auto dma_piece = [&](int t, int buf, int j) __attribute__((always_inline)) {  // j: 0-3 Q, 4-7 dO; the group led by j
      if (j % kDkv3DmaGroup != 0) return;
      const int i = j % C::DMA_PER_MAT;
      const int dst = buf * C::TILE_BYTES + ((C::BQ / C::NW) * wave + C::RPI * i) * C::ROWB;
      if (j < C::DMA_PER_MAT) dma_pieces<kDkv3DmaGroup>(rq, lds_addr_of(smem + dst), dma_q + i, t * C::BQ * q_rs);
      else dma_pieces<kDkv3DmaGroup>(rdo, lds_addr_of(smem + C::DO_BASE + dst), dma_do + i, t * C::BQ * do_rs);
    };
    auto fetch_rc = [&](int t) __attribute__((always_inline)) {
      rc = buf_load_f32(rrc, (t * C::BQ + rc_row_now()) * 4);
      if constexpr (C::RCN == 2) rc_d = buf_load_f32(rrc_d, (t * C::BQ + rc_row_now()) * 4);
    };
    auto fetch_tile = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < C::NP; ++j) dma_piece(t, buf, j);
      fetch_rc(t);
    };
    // everything of the fetched tile has landed (vmcnt(0)): publish the scaled row constants, then meet.
    // In the steady state the two halves sit four slots apart (FA_DKV3_SPLIT_COMMIT): LDS operations complete in order, so
    // by the time at most the 8 transposed-fragment reads issued in between are still outstanding the ds_write has
    // completed -- lgkmcnt(8) instead of draining every read in flight with lgkmcnt(0) in front of the barrier.
    auto publish_tile = [&](int t, int buf, bool fetched) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
#ifndef FA_DKV3_NO_VMWAIT   // (timing ablations only: results are wrong without the wait / the barrier)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#endif
      if (fetched) {
        FA_LDS float* rcp = (FA_LDS float*)(smem + C::ROWC_OFF + buf * C::ROWC_BYTES);
        // rows past S_q must give P = 0 (K:355-356): exp2(-inf) = 0
        const float lse_c = (t * C::BQ + rc_row_now() < Sq) ? -rc * kLog2e : -INFINITY;
        rcp[tid] = rc_lse ? lse_c : -rc;  // rcp[row] = -LSE*log2e, rcp[BQ + row] = -delta
        if constexpr (C::RCN == 2) rcp[C::BQ + tid] = -rc_d;
      }
      asm volatile("" ::: "memory");
    };
    auto meet = [&](auto later_reads_tag) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      constexpr int K = decltype(later_reads_tag)::value;   // LDS operations issued after the publishing ds_write
      __builtin_amdgcn_s_waitcnt(0xC07F | (K << 8));        // lgkmcnt(K)
#ifndef FA_DKV3_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
    };
    auto commit_tile = [&](int t, int buf, bool fetched) __attribute__((always_inline)) {
      publish_tile(t, buf, fetched);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the ds_write above and every LDS read issued so far
#ifndef FA_DKV3_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
#ifdef FA_DKV3_SKEW   // A/B hook: wave w leaves the barrier w * FA_DKV3_SKEW s_nop-15 groups late (de-phases the waves' DMA issue)
      wave_skew(wave);
#endif
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

    // ---- the pipeline: state carried from block to block, tile to tile ----
    f32x16 S_[2], P_[2];   // [key group]: score / dP accumulators of the block in flight; the OTHER set holds the
                           // previous block's exponent arguments -> P and dP - delta -> dS
    f32x16 NL[FOLD ? 1 : 2], ND;      // row constants of a query block ([query block parity] for NL: the exact-fma path (fp16)
                           // needs the previous block's -LSE*log2e while the next block's is being read)
    u32x4 RF[8];           // row fragments of the current query block: Q k-steps 0-3, dO k-steps 0-3
    vec8 TF[8];            // transposed fragments of the previous block's query block: dO^T (db, e) 0-3, Q^T 4-7
    u32x4 pk[2][2], sk[2][2];   // [key group][k-step]: packed P and dS of a block (written while the other set is read)
    // causal mask of a block, one comparison per element: the score in register i of lane (r, h) belongs to key kw[g] + r
    // and query row qb0 + c_i + 4h with c_i = (i & 3) + 8 (i >> 2); it is dead iff  thr = kw[g] - qb0 + r - 4h  >  c_i.
    // A block entirely below the diagonal has thr < 0 (nothing masked), one entirely above it thr >= 32 (everything): the
    // same formula serves the three tiles level with the key tile, whatever a wave's key groups see of them.
    int thr[2] = {0, 0};
    const int lane_thr = r - 4 * h;
    auto rowc_read = [&](const FA_LDS char* rcp, int b, int q, f32x16& nl, f32x16& nd) __attribute__((always_inline)) {
      const f32x4 a = *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * q + 4 * h) * 4);
      const f32x4 d = *(const FA_LDS f32x4*)(rcp + (C::BQ + 32 * b + 8 * q + 4 * h) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        nl[4 * q + j] = a[j];
        nd[4 * q + j] = d[j];
      }
    };
    // fill: the first iteration's "previous block" is neutral (P = 0, dS = 0 * 0, packed P / dS and fragments 0)
    auto pipe_fill = [&](int t, int buf) __attribute__((always_inline)) {
      // the pieces a previous tile step would have issued in its iteration 7 (nothing of tile t + 1 is in flight yet)
#pragma unroll
      for (int j = 0; j < C::NP; ++j)
        if (dkv3_dma_iter(j) == C::NI - 1) dma_piece(t + 1, buf ^ 1, j);
      const FA_LDS char* qt = smem + buf * C::TILE_BYTES;
      const FA_LDS char* dt = smem + C::DO_BASE + buf * C::TILE_BYTES;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        S_[1][i] = i < 11 ? 0.f : -INFINITY;   // block_valu: exps 0-10 of a block are over when its iteration ends (P = 0),
        P_[1][i] = 0.f;                        // exps 11-15 follow (exp2(-inf) = 0); dS = 0 * 0
        if (!FOLD) NL[FOLD ? 0 : 1][i] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        pk[1][e] = u32x4{0u, 0u, 0u, 0u};
        sk[1][e] = u32x4{0u, 0u, 0u, 0u};
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
    // VALU work of ONE block at pipeline time tau = slots since the start of the block's own iteration.  A block's 16
    // exps, 16 dS multiplies and 16 packs are spread over 22 slots, one of each kind per slot (the guide's rule for one
    // wave per SIMD: at most one transcendental among <= 5 fillers per MFMA gap; two exps per gap, as family 2 places
    // them, do not hide):   exp e at tau = 5 + e      (the S chain ended at slot 3: past the MFMA -> VALU wait states)
    //                       pack P pair j at 7 + 2j   (pk complete at 21 = slot 5 of the next iteration; read from slot 8)
    //                       dS e at tau = 10 + e      (the dP chain ended at slot 7)
    //                       pack dS pair j at 12 + 2j (sk complete at 26 = slot 10 of the next iteration; read from slot 12)
    // so every slot carries tau = s of the block in flight and tau = 16 + s of the previous one.
    auto block_valu = [&](int tau, f32x16& X, f32x16& Y, u32x4 (&pkb)[2], u32x4 (&skb)[2], const f32x16& nl, auto mask_tag,
                          int thr_b) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(mask_tag)::value;
      const int e = tau - 5, m = tau - 10;
      if (e >= 0 && e < 16) {
        // the block's FIRST exp not above this slot's MFMA (fa_common.h here; hipcc pads an s_nop behind every such
        // statement -- one per block is free, one per exp was not)
        float x = e == 0 ? here(X[e]) : X[e];
        if constexpr (!FOLD) x = __builtin_fmaf(x, c2, nl[e]);
        if constexpr (MASK) x = thr_b > (e & 3) + 8 * (e >> 2) ? -INFINITY : x;
        X[e] = __builtin_amdgcn_exp2f(x);
      }
      if (m >= 0 && m < 16) Y[m] = X[m] * Y[m];
      if (tau >= 7 && tau <= 21 && ((tau - 7) & 1) == 0) {
        const int j = (tau - 7) >> 1;
        pkb[j >> 2][j & 3] = pack2<T>(X[2 * j], X[2 * j + 1]);
      }
      if (tau >= 12 && tau <= 26 && ((tau - 12) & 1) == 0) {
        const int j = (tau - 12) >> 1;
        skb[j >> 2][j & 3] = pack2<T>(Y[2 * j], Y[2 * j + 1]);
      }
    };
    // one block iteration; I = 2 * query block + key group.  `qt` / `dt` / `rct`: current tile; `qn` / `dn` / `rcn`: next
    // tile (read from iteration 7, slot 4 on -- after the commit).  `hook(I, s, 0)` runs before the slot's MFMA (the
    // commit), `hook(I, s, 1)` right after it (LDS-DMA pieces: their issue then overlaps the MFMA just started).
    auto block_iter = [&](auto i_tag, auto mask_tag, int row0, const FA_LDS char* qt, const FA_LDS char* dt,
                          const FA_LDS char* rct, const FA_LDS char* qn, const FA_LDS char* dn, const FA_LDS char* rcn,
                          auto&& hook) __attribute__((always_inline)) {
      constexpr int I = decltype(i_tag)::value;
      if constexpr (decltype(mask_tag)::value) thr[I & 1] = lane_thr + (kw[I & 1] - row0 - 32 * (I >> 1));
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
          dvacc[pg][db] = T::mfma(TF[2 * db + e], as_vec8<T>(pk[pg][e]), dvacc[pg][db]);
        } else {
          const int n = s - 12, e = n >> 1, db = n & 1;
          dkacc[pg][db] = T::mfma(TF[4 + 2 * db + e], as_vec8<T>(sk[pg][e]), dkacc[pg][db]);
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
        // ---- VALU: the previous block at tau = 16 + s, this block at tau = s ----
        block_valu(16 + s, S_[pg], P_[pg], pk[pg], sk[pg], NL[FOLD ? 0 : ((g == 0 ? qb + 1 : qb) & 1)], mask_tag, thr[pg]);
        block_valu(s, S_[g], P_[g], pk[g], sk[g], NL[FOLD ? 0 : (qb & 1)], mask_tag, thr[g]);
        // An MFMA reads its C operand over its passes: a VALU write to those registers within ~13 wait states corrupts
        // it (hipcc pads this WAR hazard for its own MFMAs, not for an asm statement).  In the second key group's
        // iteration the row constants are DEAD after the chain start that reads them (slot 0 / slot 4) until their
        // reload, so hipcc reuses their registers for exp results at once -- seen as run-to-run differences in dK of
        // the second key group only.  Keep them live for one more slot.
        if (g == 1 && s == 1 && FOLD) keep_live(NL[0]);
        if (g == 1 && s == 5) keep_live(ND);
        __builtin_amdgcn_sched_barrier(0);
#ifdef FA_STAMPS_SLOTS   // (with -DFA_STAMPS: where inside the iterations next to the tile boundary the cycles go)
        if (I == 0 || I == C::NI - 1 || I == 3) {
          unsigned long long now_;
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
          slot_seg[I == 0 ? 0 : (I == 3 ? 2 : 1)][s] += now_ - slot_last_;
          slot_last_ = now_;
          __builtin_amdgcn_sched_barrier(0);
        } else if (s == 15) {
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(slot_last_)::"memory");
        }
#endif
      }
#ifdef FA_STAMPS
      FA3_STAMP(I * 8 / C::NI);   // seg[0..7]: block iteration I of a tile (pairs of iterations at 16 per tile; the last without its commit)
      ++nblk_;
#endif
    };
    // the last block's exp / dS / dV^T / dK^T once nothing follows it
    auto pipe_drain = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s >= 8 && s < 12) {
          const int n = s - 8, e = n >> 1, db = n & 1;
          dvacc[1][db] = T::mfma(TF[2 * db + e], as_vec8<T>(pk[1][e]), dvacc[1][db]);
        } else if (s >= 12) {
          const int n = s - 12, e = n >> 1, db = n & 1;
          dkacc[1][db] = T::mfma(TF[4 + 2 * db + e], as_vec8<T>(sk[1][e]), dkacc[1][db]);
        }
        block_valu(16 + s, S_[1], P_[1], pk[1], sk[1], NL[FOLD ? 0 : ((C::QB - 1) & 1)], std::integral_constant<bool, CAUSAL>{}, thr[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto step_pipe = [&](int t, int buf, auto mask_tag) __attribute__((always_inline)) {
      const bool more = t + 1 < ntiles;
      const FA_LDS char* qt = smem + buf * C::TILE_BYTES;
      const FA_LDS char* dt = smem + C::DO_BASE + buf * C::TILE_BYTES;
      const FA_LDS char* rct = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
      const int nb = buf ^ 1;
      const FA_LDS char* qn = smem + nb * C::TILE_BYTES;
      const FA_LDS char* dn = smem + C::DO_BASE + nb * C::TILE_BYTES;
      const FA_LDS char* rcn = smem + C::ROWC_OFF + nb * C::ROWC_BYTES;
      // No branch anywhere in a tile step: hipcc sinks instructions across a conditional branch into the block that uses
      // their results (the exps of the slots before the branch then run in one burst behind it), which sched_barrier
      // cannot prevent.  Past the last tile the fetches are simply out of range: the buffer descriptors return / write
      // nothing, and the buffer they would have filled is not read again.
      (void)more;
      fetch_rc(t + 1);
      auto hook = [&](int I, int s, int phase) __attribute__((always_inline)) {
        if (phase == 0) {
#if FA_DKV3_SPLIT_COMMIT
          if (I == C::NI - 1 && s == 0) publish_tile(t + 1, nb, true);
#endif
          if (I == C::NI - 1 && s == 4) {   // every read of this tile's buffers is issued: hand the other buffer over
            FA3_STAMP(7);
#if FA_DKV3_SPLIT_COMMIT
            meet(std::integral_constant<int, 8>{});   // slots 0-3 of this iteration: 4 x 2 ds_read_b64_tr_b16 since the publish
#else
            commit_tile(t + 1, nb, true);
#endif
            FA3_STAMP(8);   // seg[8]: the commit (vmcnt(0), row constants, lgkmcnt(0), barrier)
          }
          return;
        }
#pragma unroll
        for (int j = 0; j < C::NP; ++j)
          if (dkv3_dma_iter(j) == I && dkv3_dma_slot(j) == s) {
            if (I == C::NI - 1) dma_piece(t + 2, buf, j);   // after this tile's commit: its own buffer is free
            else dma_piece(t + 1, nb, j);
          }
      };
      block_iter(std::integral_constant<int, 0>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 1>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 2>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 3>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 4>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 5>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 6>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      block_iter(std::integral_constant<int, 7>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      if constexpr (C::NI == 16) {
        block_iter(std::integral_constant<int, 8>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 9>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 10>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 11>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 12>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 13>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 14>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
        block_iter(std::integral_constant<int, 15>{}, mask_tag, t * C::BQ, qt, dt, rct, qn, dn, rcn, hook);
      }
    };

#ifdef FA_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    commit_tile(t_start, t_start & 1, t_start < ntiles);  // first tile landed (and the K/V fragments)
    int t = t_start;
    if (t < ntiles) {
      pipe_fill(t, t & 1);
      FA3_STAMP(10);  // seg[10]: pipeline fill
      // the tiles level with the key tile (and one more, so that the last masked block has left the pipeline): masked
      for (; t < t_diag_end; ++t) step_pipe(t, t & 1, std::true_type{});
      FA3_STAMP(9);   // seg[9]: (causal) the masked tiles
      for (; t < ntiles; ++t) step_pipe(t, t & 1, std::false_type{});
      pipe_drain();
    }
    FA3_STAMP(11);    // seg[11]: drain

    __syncthreads();  // every wave is done with the tile buffers: they become the staging area
    FA_LDS char* stage = smem + wave * 32 * C::ROWB;
    // dK = dS^T Q * scale; with the pre-scaled Q (= Q * scale * log2e) in LDS that is dS^T Q' * ln 2
    const float dk_mul = (FOLD && p.q_prescaled) ? kLn2 : p.scale;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      store_tile_rows<D, T>(dkacc[g], dk_mul, stage, rdk, kw[g] * dk_rs, lane, dk_rs);
      store_tile_rows<D, T>(dvacc[g], 1.0f, stage, rdv, kw[g] * dv_rs, lane, dv_rs);
    }
    FA3_STAMP(12);    // seg[12]: epilogue
  }  // pass
#ifdef FA_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 4 + wave) * 16;
    for (int i = 0; i < 13; ++i) d[i] = seg[i];
    d[13] = nblk_;
    unsigned long long clk1_, rt1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_), "=s"(rt1_)::"memory");
    d[14] = clk1_ - clk0_;
    d[15] = rt1_ - rt0_;
#ifdef FA_STAMPS_SLOTS   // a second table behind the first (tools/stamps_dkv3.py --slots)
    unsigned long long* e = (unsigned long long*)p.dbg + (size_t)gridDim.x * 4 * 16 + ((size_t)blockIdx.x * 4 + wave) * 48;
    for (int i = 0; i < 48; ++i) e[i] = slot_seg[i / 16][i % 16];
#endif
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
