// FlashAttention forward, fourth schedule family for gfx950: ONE wave per SIMD, 64 query rows per wave, a continuous
// hand-ordered software pipeline (head dim 64 and 128).
//
// Same maths as fa_fwd.hip (reference kernel code/_flash_attention_kernel_optimized.py:35-129: fp32 scores and softmax
// state, P rounded to the input dtype for P @ V (K:115), O = o / l cast on store (K:120-123), LSE = m + ln l (K:126),
// top-left aligned causal mask (K:102), keys >= S_k masked (K:94)); what differs is the schedule and ONE numerical
// choice (below: the running maximum is not tracked, which changes nothing but the common scale of P, o and l).
//
// Why a fourth family.  The forward is bound by vector issue, not by the matrix pipe: per 32 x 64 score tile 16 MFMAs
// (512 cycles) stand against 32 exp + 32 row-sum adds + 16 packs (~470 issue cycles) + the MFMAs' own issue (128).
// Families 1-3 lose a further third to everything around that minimum: a barrier every 16 MFMAs, phases that do not
// overlap inside a wave, a branch per tile for the lazy-maximum test, and LDS fragment reads that serve one 32-row block.
// Here a workgroup is 256 query rows, a wave owns TWO 32-row blocks and
//   * every K fragment (S^T = K Q^T) and every V^T fragment (O^T += V^T P^T) is read from LDS once and feeds both blocks;
//   * the steady state has NO control flow and no cross-lane work at all.  The score chains start from -m (C operand != D),
//     where m is a per-row constant taken ONCE per pass from the row's first 32 keys; exp2(s - m) is then summed and
//     multiplied into V exactly as with a running maximum -- softmax is invariant to the shift, P keeps its relative
//     precision at any magnitude (16-bit floating point), o and l are fp32.  Only overflow could hurt (a later score
//     exceeding m by ~2^100 for bf16, ~2^15 for fp16): the row sums are checked ONCE at the end of the pass, and a pass that
//     fails is redone with the exact row maxima from a max-only sweep (same pipeline, second attempt) -- results are
//     exact either way, the common case pays nothing per tile;
//   * one block iteration = one 32-key block for both row blocks = 2 KS score MFMAs + 4 DB  P V MFMAs of the PREVIOUS
//     key block, with that block's exps / row sums / packs spread evenly under them (D = 64: two exp, two adds and one
//     pack per MFMA; D = 128: one exp and add per MFMA), each closed by sched_barrier(0);
//   * the per-tile commit (counted vmcnt, s_barrier) sits inside the tile's last iteration; K/V tiles arrive by LDS-DMA
//     into a ring of three (D = 64, 128-key tiles) or four (D = 128, 64-key tiles) buffers, one to two tiles ahead.
// Round 4, causal launches: PERSISTENT workgroups (one per CU walks the work list) and a K/V ring that runs on from pass to
// pass -- the LDS-DMA slots of a pass's last NBUF - 1 tile steps fetch the NEXT pass's first tiles (they used to fetch past
// the end), at D = 64 together with its Q rows, staged through LDS behind the ring and read back as ds_read_b128 fragments;
// the work-list decode is a multiply-shift (fa_kernels.h FastDiv).  A pass that follows another starts without a memory
// wait and without a barrier of its own (the previous end-of-pass check held it).
// Causal: workgroups take the query-tile pair (nq-1-i, i).  A wave's two row blocks lie in the two HALVES of the 256-row
// tile (rows 32w.. and 128 + 32w..): of the 256 keys level with the query tile the first 128 are unmasked for every row
// block 1 and the second 128 invisible to every row block 0 -- those tiles run the same pipeline with a one-compare mask per
// element on one row block only, respectively without row block 0 at all (half the MFMAs); the ragged last tile of any
// launch masks both row blocks.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

template <int D_>
struct Fwd4Cfg {
  static constexpr int D = D_;
  static constexpr int BM = 256, NT = 256, NW = 4;
  static constexpr int BN = D == 64 ? 128 : 64;             // keys per LDS tile
  static constexpr int NBUF = D == 64 ? 3 : 4;              // ring depth
  static constexpr int NKB = BN / 32;                       // 32-key block iterations per tile
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BN * ROWB;              // 16 KiB for both head dims
  static constexpr int V_BASE = NBUF * TILE_BYTES;          // K[NBUF], then V[NBUF]
  static constexpr int FLAG_OFF = 2 * NBUF * TILE_BYTES;    // one word: some wave's row sums overflowed
  // behind the flag (D = 64; at D = 128 the ring leaves no room): the NEXT pass's Q rows, 64 per wave (two 32-row blocks
  // of 4 KiB in the K tiles' swizzled row image), staged by LDS-DMA from inside the previous pass (the kernel says when)
  static constexpr int RB_BYTES = 32 * ROWB;
  static constexpr int QS_OFF = FLAG_OFF + 16;
  static constexpr int QS_BYTES = D == 64 ? 8 * RB_BYTES : 0;
  static constexpr int LDS_BYTES = QS_OFF + QS_BYTES;
  static constexpr int PIECES = TILE_BYTES / (NW * 1024);   // 1-KiB LDS-DMA pieces per wave per matrix (4)
  static constexpr int RPI = 1024 / ROWB;                   // tile rows per piece
  // A/B hook, OFF: row sums on the MATRIX pipe (l^T += 1^T P^T: one more MFMA per (k-step, row block) with an all-ones A
  // fragment, 4 per block iteration at D = 64) instead of 32 row-sum adds.  By issue arithmetic a clear win at D = 64 (~140
  // vector-issue cycles saved per iteration for 32 of MFMA issue, and the matrix pipe is only ~60 % busy); by WALL it loses:
  // B4 H32 N4096 bf16 non-causal 0.508 vs 0.487 ms, causal 0.283 vs 0.280 ms (interleaved A/B, round 3) -- 25 % more MFMA
  // work costs more clock (the chip is power-limited) than the adds cost issue slots.
#ifndef FA_FWD4_LSUM_MFMA
#define FA_FWD4_LSUM_MFMA 0
#endif
  static constexpr bool LSUM = FA_FWD4_LSUM_MFMA && D == 64;
  static constexpr int PVG = 2 * DB + (LSUM ? 2 : 0);       // slots per k-step of the P V phase: (d block, row block) pairs [+ row sums]
  static constexpr int NSS = 2 * KS, NPV = 2 * PVG, NSLOT = NSS + NPV;   // MFMA slots of one block iteration
  static constexpr int INFLIGHT = (NBUF - 3) * 2 * PIECES;  // pieces of later tiles a commit leaves in flight (vmcnt)
};

// largest row sum accepted at the end of a pass (fa_fwd_v4.hip header): P must stay finite in 16 bit, o and l in fp32
template <typename T> struct Fwd4Limit;
template <> struct Fwd4Limit<BF16> { static constexpr float value = 1.2676506e30f; };   // 2^100
template <> struct Fwd4Limit<FP16> { static constexpr float value = 32768.0f; };        // 2^15 (fp16 max 65504)

// -DFA_STAMPS (diagnostic build, tools/stamps_fwd4.py): per-phase cycle account of a wave, written to FwdParams::dbg
#ifdef FA_STAMPS
#define FA4_STAMP(slot)                                                           \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long now_;                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += now_ - last_;                                                    \
    last_ = now_;                                                                 \
  } while (0)
#else
#define FA4_STAMP(slot) do {} while (0)
#endif

template <int D, typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void fa_fwd4_kernel(FwdParams p) {
#ifdef FA_STAMPS
  unsigned long long clk0_, rt0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_)::"memory");
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = clk0_, ntile_ = 0, npass_ = 0;
#endif
  using C = Fwd4Cfg<D>;
  using vec8 = typename T::vec8;
  constexpr bool FOLD = T::kFoldScale;
  // the score chains start from -m (a 16-register block per row block) wherever the scale is folded into Q (bf16); at
  // D = 128, where registers are short, one subtraction per element instead measured -5 % (A/B, round 3): kept
  constexpr bool CHAIN_M = FOLD;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // Work list: one item = one 256-row query tile, or (causal) the tile pair (nq-1-i, i) -> equal work per item.  Causal
  // launches are PERSISTENT: a workgroup per CU walks items blockIdx.x, blockIdx.x + gridDim.x, ... (a multiple of 8 keeps a
  // workgroup on one XCD's slice list), because a pass's first K/V tiles and its Q rows are fetched from inside the PREVIOUS
  // pass (CONT / STAGEQ below) and the first pass of an item has no previous pass unless the workgroup stays.
  // CONT: the K/V ring runs on from pass to pass (and from item to item).  Causal: always.  Non-causal: when the key range
  // is whole tiles and at least NBUF - 1 of them (the launcher makes the same test and only then launches persistently) --
  // the last NBUF - 1 plain tile steps then carry the next item's first tiles as the masked tiles do for a causal pass.
  const bool CONT = CAUSAL || (p.Sk % C::BN == 0 && p.Sk / C::BN >= C::NBUF - 1);
  constexpr bool STAGEQ = D == 64;   // the next pass's Q rows come through LDS (Fwd4Cfg::QS_OFF)
  const bool paired = CAUSAL && p.pair;
  const int nq = p.nq_tiles;
  const int per_bh = paired ? (nq + 1) / 2 : nq;
  const int n_items = per_bh * p.B * p.H;
  const int Sq = p.Sq, Sk = p.Sk;
  struct Work {
    int b, h, idx, npass;
  };
  auto decode = [&](int item) __attribute__((always_inline)) -> Work {
    const int w = xcd_remap(item, n_items);
    const int bh = p.div_per_bh.div(w), idx = w - bh * per_bh, b = p.div_h.div(bh);   // (fa_kernels.h FastDiv)
    return Work{b, bh - b * p.H, idx, (paired && idx != nq - 1 - idx) ? 2 : 1};
  };
  auto tile_of = [&](const Work& wk, int pass) __attribute__((always_inline)) -> int {   // heavy tile first
    return paired ? (pass == 0 ? nq - 1 - wk.idx : wk.idx) : (CAUSAL ? nq - 1 - wk.idx : wk.idx);
  };

  // Q, K, V, O may be strided views with a contiguous head dim (fa_fwd.hip); no variable-length launches here
  const int q_rs = p.lq.rs, kv_rs = p.lk.rs, o_rs = p.lo.rs;
  // (a descriptor that is not `valid` -- nothing follows the last pass -- is empty: its pieces fetch nothing)
  auto rsrc_q = [&](const Work& wk, bool valid) __attribute__((always_inline)) {
    return make_rsrc((const char*)p.q + wk.b * p.lq.sb + wk.h * p.lq.sh, valid ? view_bytes(Sq, q_rs, C::ROWB) : 0u);
  };
  auto rsrc_k = [&](const Work& wk, bool valid) __attribute__((always_inline)) {
    return make_rsrc((const char*)p.k + wk.b * p.lk.sb + wk.h * p.lk.sh, valid ? view_bytes(Sk, kv_rs, C::ROWB) : 0u);
  };
  auto rsrc_v = [&](const Work& wk, bool valid) __attribute__((always_inline)) {
    return make_rsrc((const char*)p.v + wk.b * p.lv.sb + wk.h * p.lv.sh, valid ? view_bytes(Sk, kv_rs, C::ROWB) : 0u);
  };

  // ---- loop-invariant per-lane addresses ----
  int dma_src[C::PIECES];   // per-lane global source offset of this wave's pieces (K and V share their row stride)
#pragma unroll
  for (int i = 0; i < C::PIECES; ++i) {
    const int row = (C::BN / C::NW) * wave + C::RPI * i + lane / C::CPR;
    dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16 - 1024 * (i & 1);   // pieces go out in pairs
  }
  int k_off[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) k_off[ks] = lds_off<D>(r, 2 * ks + h);
  int v_off[2][C::DB];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) v_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);
  const float c2 = p.scale * kLog2e;

  // the ring is read before it is written in two places (pipeline fill, ragged last tile): keep it finite
  if (Sk % C::BN != 0) {
    lds_zero_fill(smem, C::FLAG_OFF, C::NT, tid);
    __syncthreads();
  }

  if (tid == 0) *(FA_LDS int*)(smem + C::FLAG_OFF) = 0;   // "some wave's row sums overflowed" (ordered by the first barrier below)
  bool primed = false;   // CONT: this pass's first NBUF - 1 tiles (and, STAGEQ, its Q rows) were requested from inside the previous pass
  int r0 = 0;            // ring slot of the pass's tile 0: tile t sits in slot (r0 + t) % NBUF

  int item = blockIdx.x, pass = 0;
  Work wk = decode(item);
  Work nwk_item = decode(min(item + (int)gridDim.x, n_items - 1));   // the workgroup's next item, decoded once
  for (;;) {
    const int b_ = wk.b, h_ = wk.h;
    const __amdgpu_buffer_rsrc_t rk = rsrc_k(wk, true), rv = rsrc_v(wk, true);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((char*)p.o + b_ * p.lo.sb + h_ * p.lo.sh, view_bytes(Sq, o_rs, C::ROWB));
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(p.lse + b_ * p.lse_sb + h_ * p.lse_sh, (unsigned)Sq * 4);
    // what follows this pass: the pair's second pass, or the first pass of the workgroup's next item
    const bool more_pass = pass + 1 < wk.npass, more_item = item + (int)gridDim.x < n_items;
    const bool has_next = CONT && (more_pass || more_item);
    const Work nwk = more_pass ? wk : nwk_item;
    const int nq0_wg = tile_of(nwk, more_pass ? pass + 1 : 0) * C::BM;

    FA4_STAMP(8);   // seg[8]: loop bookkeeping (the next item decoded, descriptors)
    const int q0_wg = tile_of(wk, pass) * C::BM;
    // this wave's two 32-row blocks: rows qrow(rb) + r with qrow(rb) = q0_wg + 128 rb + 32 wave -- one block in each half
    // of the 256-row tile, so that under the causal mask the SECOND half of the key tiles level with the query tile is
    // invisible to row block 0 of EVERY wave (skipped outright) and the first half is unmasked for every row block 1
    const int qw = q0_wg + 32 * wave;
    auto qrow = [&](int rb) __attribute__((always_inline)) { return qw + 128 * rb; };

    const int kv_end = CAUSAL ? min(Sk, q0_wg + C::BM) : Sk;
    const int ntiles = (kv_end + C::BN - 1) / C::BN;
    // tiles [0, nplain) need no mask for ANY row of the workgroup (the tile schedule must be workgroup-uniform)
    const int nplain = CAUSAL ? min(Sk / C::BN, q0_wg / C::BN) : Sk / C::BN;

    // ---- LDS-DMA of one K/V tile: 2 x PIECES pieces per wave, issued in pairs (one M0 write each) ----
    auto dma_pair_of = [&](const __amdgpu_buffer_rsrc_t& rkx, const __amdgpu_buffer_rsrc_t& rvx, int t, int slot, int j) __attribute__((always_inline)) {
      constexpr int HALF = C::PIECES / 2;   // j: 0 .. PIECES-1; pairs 0 .. PIECES/2-1 K, then V
      const int i = 2 * (j % HALF);
      const int dst = slot * C::TILE_BYTES + ((C::BN / C::NW) * wave + C::RPI * i) * C::ROWB;
      if (j < HALF) dma_pieces<2>(rkx, lds_addr_of(smem + dst), dma_src + i, t * C::BN * kv_rs);
      else dma_pieces<2>(rvx, lds_addr_of(smem + C::V_BASE + dst), dma_src + i, t * C::BN * kv_rs);
    };
    auto dma_pair = [&](int t, int slot, int j) __attribute__((always_inline)) { dma_pair_of(rk, rv, t, slot, j); };
    auto fetch_tile = [&](int t, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j) dma_pair(t, slot, j);
    };
    // Q rows of row block rb of the query tile at q0 -> this wave's part of the staging area, as two pairs of 1-KiB pieces
    // (8 rows each) in the K tiles' swizzled row image: the fragments are then ds_read_b128 row reads like K's.  Fetched as
    // fragments (8 loads per lane, each touching 32 cache lines) the same rows cost the CU's address unit ~63 cycles per
    // load and the wave the whole memory latency at the top of a pass (fa_bwd_dq_v4.hip; profiles/r04_ab_lines.txt).
    auto stage_q_pair = [&](const __amdgpu_buffer_rsrc_t& rqx, int q0, int rb, int hf) __attribute__((always_inline)) {
      if constexpr (STAGEQ) {
        const int prow = lane >> 3;
        int voff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = 8 * (2 * hf + i) + prow;   // row of the 32-row block
          voff[i] = (q0 + 128 * rb + 32 * wave + row) * q_rs + swz_chunk<D>(row, lane & 7) * 16 - 1024 * i;
        }
        dma_pieces<2>(rqx, lds_addr_of(smem + C::QS_OFF + (2 * wave + rb) * C::RB_BYTES + 2048 * hf), voff, 0);
      }
    };
    // The K/V tiles do not depend on the query tile: the ring is primed BEFORE the Q fragments are fetched (one memory
    // latency per pass instead of two).  Only the first pass of a causal workgroup (and every non-causal pass) comes here
    // unprimed: later passes found their first tiles and Q rows requested by the masked tiles of the pass before.
    const bool was_primed = primed;
    if (!primed) {
      r0 = 0;
      if constexpr (STAGEQ) {
        const __amdgpu_buffer_rsrc_t rq = rsrc_q(wk, true);
#pragma unroll
        for (int g = 0; g < 4; ++g) stage_q_pair(rq, q0_wg, g >> 1, g & 1);
      }
#pragma unroll
      for (int t = 0; t < C::NBUF - 1; ++t) fetch_tile(t, t);
    }
    FA4_STAMP(9);   // seg[9]: an unprimed pass's requests
    // ---- resident operands: Q^T fragments of both row blocks (B operand), scaled once (bf16) ----
    u32x4 qf[2][C::KS];
    if constexpr (STAGEQ) {
      // this wave's own pieces, nobody else's: no barrier.  Primed passes waited at the previous end-of-pass check.
      if (!was_primed) __builtin_amdgcn_s_waitcnt(0x0F70 | (((C::NBUF - 1) * 2 * C::PIECES) & 15) | ((((C::NBUF - 1) * 2 * C::PIECES) >> 4) << 14));
      asm volatile("" ::: "memory");
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          vec8 q = as_vec8<T>(lds_read16(smem + C::QS_OFF + (2 * wave + rb) * C::RB_BYTES + k_off[ks]));
          if constexpr (FOLD) q = scale_frag<T>(q, c2);
          qf[rb][ks] = __builtin_bit_cast(u32x4, q);
        }
    } else {
      const __amdgpu_buffer_rsrc_t rq = rsrc_q(wk, true);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          vec8 q = as_vec8<T>(buf_load16(rq, (qrow(rb) + r) * q_rs + (2 * ks + h) * 16));
          if constexpr (FOLD) q = scale_frag<T>(q, c2);
          qf[rb][ks] = __builtin_bit_cast(u32x4, q);
        }
    }
    FA4_STAMP(10);   // seg[10]: Q fragments
    // per-lane mask base: score register i of lane (r, h) in a block starting at key kb0 is key kb0 + c_i + 4h, row
    // qrow(rb) + r; it is dead iff the key exceeds the row (causal) or the last key:  c_i > thr = base[rb] - kb0
    int mask_base[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) mask_base[rb] = (CAUSAL ? min(qrow(rb) + r, Sk - 1) : Sk - 1) - 4 * h;

    // tile t + 1 has landed for every wave, and every wave is past its reads of tile t - 1
    auto commit = [&]() __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      // vmcnt(INFLIGHT) only.  No LDS wait: the slot the barrier hands over to the DMA held tile t - 1, whose last reads
      // (the V fragments of its last key block, first iteration of tile t) fed MFMAs a whole tile ago, and draining the
      // K / V fragment reads in flight (lgkmcnt(0)) would stall the pipeline once per tile for nothing.
#ifdef FA_FWD4_COMMIT_LGKM0   // A/B hook: the conservative form
      __builtin_amdgcn_s_waitcnt(0x0070 | (C::INFLIGHT & 15) | ((C::INFLIGHT >> 4) << 14));
#else
      __builtin_amdgcn_s_waitcnt(0x0F70 | (C::INFLIGHT & 15) | ((C::INFLIGHT >> 4) << 14));
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };

    // ---- pipeline state ----
    f32x16 S_[2][2];        // [set][row block]: score accumulators -> exponent arguments -> P of a key block
    f32x16 negm[CHAIN_M ? 2 : 1];   // [row block]: -m in every register (CHAIN_M: the score chains start from it)
    float mrow[2], nmc[2];  // the row constant m (accumulator units) and -m * c2 (exact-fma path)
    float l[2];             // this lane's partial row sums
    u32x4 pk[2][2][2];      // [set][row block][k-step]: packed P of a key block
#ifdef FA_FWD4_KR_FULL   // A/B hook
    constexpr int KR = C::KS;
#else
    constexpr int KR = C::KS < 4 ? C::KS : 4;   // K fragment ring: fragment ks lives in KF[ks % KR]
#endif
    u32x4 KF[KR];           // K row fragments of the key block being scored
    vec8 VF[2 * C::DB];     // V^T fragments (k-step e, d block db) -> index e * DB + db of the key block being multiplied
    f32x16 oacc[2][C::DB];
    f32x16 lacc[C::LSUM ? 2 : 1];   // LSUM: row sums as an MFMA accumulator (every register of a lane holds its row's sum)
    const unsigned one2 = std::is_same<T, BF16>::value ? 0x3F803F80u : 0x3C003C00u;   // two 16-bit ones
    const u32x4 ones = {one2, one2, one2, one2};
    int thr[2][2];   // [set][row block]: mask threshold of a block; plain blocks leave the neutral 1 << 20 of the fill (masks nothing)

    // VALU work of ONE key block at pipeline time tau (slots since the start of its own iteration): exps q = 0..31 in the
    // order rb0[0..7], rb1[0..7], rb0[8..15], rb1[8..15], spread evenly over the NSLOT slots from tau = NSS on; the row-sum
    // add of a value and the pack of a pair follow ONE SLOT later (the last at tau = NSLOT + NSS, before the MFMA that needs it).
    auto block_valu = [&](int tau, f32x16 (&X)[2], u32x4 (&PK)[2][2], const int (&TH)[2], auto mask_tag, auto has0_tag) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(mask_tag)::value, HAS0 = decltype(has0_tag)::value;   // HAS0: row block 0 takes part
      // exp q issues at slot NSS + q * NSLOT / 32 (the 32 exps spread evenly over one iteration): the q of a slot are
      // [ceil(32 (tau - NSS) / NSLOT), ceil(32 (tau - NSS + 1) / NSLOT)), at most two
      auto first_q = [](int t) { return t <= 0 ? 0 : (32 * t + C::NSLOT - 1) / C::NSLOT; };
      auto work = [&](int t, bool exps) __attribute__((always_inline)) {   // t = tau - NSS of the exps concerned
        if (t < 0 || t >= C::NSLOT) return;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = first_q(t) + u;
          if (q >= first_q(t + 1) || q >= 32) continue;
          const int rb = (q >> 3) & 1, e = (q & 7) + 8 * (q >> 4);
          if (rb == 0 && !HAS0) continue;
          if (exps) {
            float x = CHAIN_M ? X[rb][e] : (FOLD ? X[rb][e] + nmc[rb] : __builtin_fmaf(X[rb][e], c2, nmc[rb]));
            if constexpr (MASK) x = (e & 3) + 8 * (e >> 2) > TH[rb] ? -INFINITY : x;
            X[rb][e] = __builtin_amdgcn_exp2f(x);
          } else {
            if constexpr (!C::LSUM) l[rb] += X[rb][e];
            if (q & 1) {
              const int j = e >> 1;   // pair (e - 1, e)
              PK[rb][j >> 2][j & 3] = pack2<T>(X[rb][e - 1], X[rb][e]);
            }
          }
        }
      };
      work(tau - C::NSS, true);
      // row sums (unless the matrix pipe takes them) and packs ONE SLOT later: a transcendental's result is not there for the next instruction
      work(tau - 1 - C::NSS, false);
    };
    // LDS addresses are `per-lane base register (set once per tile, opaque to hipcc) + immediate`: left alone, hipcc hoists
    // every (lane offset + constant) pair out of the tile loop -- thirty values parked in AGPRs and ~100 vector instructions
    // per tile to rebuild the addresses from them (the ring is larger than the 16-bit immediate of ds_read)
    // MODE of the block in flight: 0 plain, 1 both row blocks masked element-wise, 2 row block 0 masked / row block 1 plain,
    // 3 row block 0 absent (invisible to every wave) / row block 1 masked.  PST = what the PREVIOUS block was: 0 plain, 1 masked
    // (some row block, element-wise; a plain row block of it carries a threshold that masks nothing), 2 masked without row block 0.
    auto block_iter = [&](auto j_tag, auto set_tag, auto mode_tag, auto pst_tag, int kb0, const int (&kc)[C::KS], int kc_imm,
                          const int (&kn)[C::KS], int kn_delta, int kn_imm,
                          const int (&vc)[2][C::DB], int vc_imm, int vp0, int vp1, int vp_imm,
                          auto&& hook) __attribute__((always_inline)) {
      constexpr int J = decltype(j_tag)::value, SET = decltype(set_tag)::value, PSET = SET ^ 1;
      constexpr int MODE = decltype(mode_tag)::value, PST = decltype(pst_tag)::value;
      constexpr bool OWN0 = MODE != 3, PREV0 = PST != 2;
      using OwnMask = std::integral_constant<bool, MODE != 0>;
      using PrevMask = std::integral_constant<bool, PST != 0>;
      if constexpr (MODE != 0) {
        thr[SET][0] = mask_base[0] - kb0;
        thr[SET][1] = MODE == 2 ? (1 << 20) : mask_base[1] - kb0;
      }
#pragma unroll
      for (int s = 0; s < C::NSLOT; ++s) {
        hook(J, s);
        // ---- the MFMA of this slot ----
        if (s < C::NSS) {
          const int ks = s >> 1, rb = s & 1;
          if (rb == 1 || OWN0) {
            if (ks == 0) {
              if constexpr (CHAIN_M) T::mfma_v_first(S_[SET][rb], KF[0], qf[rb][0], negm[CHAIN_M ? rb : 0]);
              else T::mfma_v_first0(S_[SET][rb], KF[0], qf[rb][0]);
            } else {
              T::mfma_v_acc(S_[SET][rb], KF[ks % KR], qf[rb][ks]);
            }
          }
        } else {   // the previous key block, k-step e: P V for (d block, row block) pairs, then (LSUM) the two row sums
          const int n = s - C::NSS, e = n / C::PVG, m = n % C::PVG, rb = m & 1;
          if (rb == 1 || PREV0) {
            if (m < 2 * C::DB) oacc[rb][m >> 1] = T::mfma(VF[e * C::DB + (m >> 1)], as_vec8<T>(pk[PSET][rb][e]), oacc[rb][m >> 1]);
            else lacc[rb] = T::mfma(as_vec8<T>(ones), as_vec8<T>(pk[PSET][rb][e]), lacc[rb]);
          }
        }
        // ---- LDS reads into registers whose last use is just over ----
        if (s >= 2 && s <= C::NSS && (s & 1) == 0) {
          // the ring register of fragment ks = (s - 2) / 2 is free (its MFMAs were slots s - 2, s - 1): it takes fragment
          // ks + KR of THIS key block while one is left, else fragment ks + KR - KS of the NEXT key block
          const int ks = (s - 2) >> 1, nk = ks + KR;
          if (nk < C::KS) KF[ks % KR] = lds_read16(lds_at(kc[nk] + kc_imm));
          else KF[ks % KR] = lds_read16(lds_at(kn[nk - C::KS] + kn_delta + kn_imm));
        }
        // V^T fragment (e, db) of THIS key block (multiplied in the next iteration) goes into its register right after the
        // last MFMA that read the old content (slot NSS + e PVG + 2 db + 1); only without LSUM does the very last one fall
        // off the end of the iteration -- it is then read at slot 0 of the next one, from the PREVIOUS key block's image
#pragma unroll
        for (int pp = 0; pp < 2 * C::DB; ++pp) {
          const int e = pp / C::DB, db = pp % C::DB, at = C::NSS + e * C::PVG + 2 * db + 2;
          if (at < C::NSLOT && s == at)
            VF[pp] = lds_read_tr_frag<T>(lds_at(vc[0][db] + vc_imm + 16 * e * C::ROWB), lds_at(vc[1][db] + vc_imm + 16 * e * C::ROWB));
          if (at >= C::NSLOT && s == 0)
            VF[pp] = lds_read_tr_frag<T>(lds_at(vp0 + vp_imm), lds_at(vp1 + vp_imm));
        }
        // ---- VALU: the previous key block at tau = NSLOT + s, this one at tau = s ----
        block_valu(C::NSLOT + s, S_[PSET], pk[PSET], thr[PSET], PrevMask{}, std::integral_constant<bool, PREV0>{});
        block_valu(s, S_[SET], pk[SET], thr[SET], OwnMask{}, std::integral_constant<bool, OWN0>{});
        // An MFMA reads its C operand over its passes and hipcc pads that hazard for its own MFMAs only: in the LAST block of
        // a pass -m is dead after the chain starts (slots 0 and 1), and hipcc reused its registers for a masked exponent
        // argument six instructions later (tools/mfma_lint.py rule R2; fa_bwd_dkv_v3.hip met the same) -- live one more slot
        if constexpr (CHAIN_M) {
          if (s == 1 && OWN0) keep_live(negm[0]);
          if (s == 2) keep_live(negm[CHAIN_M ? 1 : 0]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the last key block's remaining softmax and its P V once nothing follows it
    auto drain = [&](auto set_tag, auto pst_tag, int rt_last) __attribute__((always_inline)) {   // rt_last: the last tile's ring slot
      constexpr int PSET = decltype(set_tag)::value ^ 1, PST = decltype(pst_tag)::value;
      constexpr bool PREV0 = PST != 2;
      if constexpr (!C::LSUM) {   // the last V^T fragment of the last key block (slot 0 of the following iteration in the steady state)
        const int base = C::V_BASE + rt_last * C::TILE_BYTES + (C::NKB - 1) * 32 * C::ROWB + 16 * C::ROWB;
        VF[2 * C::DB - 1] = lds_read_tr_frag<T>(smem + v_off[0][C::DB - 1] + base, smem + v_off[1][C::DB - 1] + base);
      }
#pragma unroll
      for (int s = 0; s < C::NSLOT; ++s) {
        if (s >= C::NSS) {
          const int n = s - C::NSS, e = n / C::PVG, m = n % C::PVG, rb = m & 1;
          if (rb == 1 || PREV0) {
            if (m < 2 * C::DB) oacc[rb][m >> 1] = T::mfma(VF[e * C::DB + (m >> 1)], as_vec8<T>(pk[PSET][rb][e]), oacc[rb][m >> 1]);
            else lacc[rb] = T::mfma(as_vec8<T>(ones), as_vec8<T>(pk[PSET][rb][e]), lacc[rb]);
          }
        }
        block_valu(C::NSLOT + s, S_[PSET], pk[PSET], thr[PSET], std::integral_constant<bool, PST != 0>{}, std::integral_constant<bool, PREV0>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // one tile: NKB block iterations; the commit + the DMA of tile t + NBUF - 1 ride in the last one.  NX != 0 (CONT: the
    // pass's last NBUF - 1 tiles): that tile lies past the end of the pass -- it is tile t + NBUF - 1 - ntiles of the NEXT
    // pass, which thereby finds its ring primed and simply runs on (r0); NX == 2 also carries the next pass's Q rows
    // (STAGEQ: four pairs of pieces, two in each of the first two block iterations).
    auto tile_step = [&](int t, int rt, auto mode_tag, auto pst_tag, auto nx_tag) __attribute__((always_inline)) {   // rt: the tile's ring slot (kept by the caller)
      constexpr int MODE = decltype(mode_tag)::value, NX = decltype(nx_tag)::value;
      // (the next pass's descriptors are put together where they are used: seven descriptors live across the plain-tile
      //  loop cost scalar registers the loop does not have)
      const __amdgpu_buffer_rsrc_t nrk = rsrc_k(nwk, has_next && NX != 0), nrv = rsrc_v(nwk, has_next && NX != 0);
      const __amdgpu_buffer_rsrc_t nrq = rsrc_q(nwk, has_next && NX == 2);
      using OwnSt = std::integral_constant<int, MODE == 0 ? 0 : (MODE == 3 ? 2 : 1)>;   // what a block of this tile is to its successor
      const int rn = rt + 1 == C::NBUF ? 0 : rt + 1, rp = rt == 0 ? C::NBUF - 1 : rt - 1;
      const int lds0 = (int)lds_addr_of(smem);   // inside the opaque bases: otherwise every address costs a second add
      // per-lane bases of this tile's K and V images, of the next tile's K image and of the previous tile's last V fragment
      // (the very first iteration has no previous key block -- its P is 0 -- but must read a FINITE image: ring slot
      // NBUF - 1 has not been written yet and uninitialised LDS may hold NaN patterns, so it reads this tile's own)
      int kA[C::KS], vA[2][C::DB], vP[2];
      const int kdelta = (rn - rt) * C::TILE_BYTES;   // next tile's K image relative to this one's (one add per read, last iteration only)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        kA[ks] = opaque(lds0 + k_off[ks] + rt * C::TILE_BYTES);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int db = 0; db < C::DB; ++db) vA[e][db] = opaque(lds0 + v_off[e][db] + C::V_BASE + rt * C::TILE_BYTES);
        vP[e] = opaque(lds0 + v_off[e][C::DB - 1] + C::V_BASE + (t == 0 ? rt * C::TILE_BYTES
                                                                  : rp * C::TILE_BYTES + (C::NKB - 1) * 32 * C::ROWB) + 16 * C::ROWB);
      }
      auto hook = [&](int J, int s) __attribute__((always_inline)) {
        if (J == C::NKB - 1) {
          if (s == 1) commit();   // before the first read of tile t + 1 (slot 2), after every read of tile t - 1
#pragma unroll
          for (int j = 0; j < C::PIECES; ++j)
            if (s == C::NSS + 1 + 2 * j) {   // into the buffer tile t - 1 has just left
              if constexpr (NX != 0) dma_pair_of(nrk, nrv, t + C::NBUF - 1 - ntiles, rp, j);
              else dma_pair(t + C::NBUF - 1, rp, j);
            }
        }
        if constexpr (NX == 2 && STAGEQ) {
          if (J < 2 && s == C::NSS + 1) stage_q_pair(nrq, nq0_wg, J, 0);
          if (J < 2 && s == C::NSS + 5) stage_q_pair(nrq, nq0_wg, J, 1);
        }
      };
      auto go = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        constexpr int kPrevImm = (J - 1) * 32 * C::ROWB + 16 * C::ROWB;   // J > 0: the previous key block lies in this tile
        if constexpr (J == 0)
          block_iter(j_tag, std::integral_constant<int, 0>{}, mode_tag, pst_tag, t * C::BN, kA, 0, kA, 0, 32 * C::ROWB, vA, 0, vP[0], vP[1], 0, hook);
        else if constexpr (J + 1 < C::NKB)
          block_iter(j_tag, std::integral_constant<int, J & 1>{}, mode_tag, OwnSt{}, t * C::BN + 32 * J, kA, J * 32 * C::ROWB, kA, 0,
                     (J + 1) * 32 * C::ROWB, vA, J * 32 * C::ROWB, vA[0][C::DB - 1], vA[1][C::DB - 1], kPrevImm, hook);
        else
          block_iter(j_tag, std::integral_constant<int, J & 1>{}, mode_tag, OwnSt{}, t * C::BN + 32 * J, kA, J * 32 * C::ROWB, kA, kdelta, 0,
                     vA, J * 32 * C::ROWB, vA[0][C::DB - 1], vA[1][C::DB - 1], kPrevImm, hook);
      };
      go(std::integral_constant<int, 0>{});
      go(std::integral_constant<int, 1>{});
      if constexpr (C::NKB == 4) {
        go(std::integral_constant<int, 2>{});
        go(std::integral_constant<int, 3>{});
      }
    };

    // ---- the row constant m: first attempt from the row's first 32 keys, second attempt (only after an overflow) exact ----
    // scores of key block `kb0` of the tile at `kt` for both row blocks (masked), folded into a running maximum
    auto scout_block = [&](const FA_LDS char* kblk, int kb0, float (&mx)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        f32x16 s;
        T::mfma_v_first0(s, lds_read16(kblk + k_off[0]), qf[rb][0]);
#pragma unroll
        for (int ks = 1; ks < C::KS; ++ks) T::mfma_v_acc(s, lds_read16(kblk + k_off[ks]), qf[rb][ks]);
        settle_mfma(s);   // asm MFMA result -> VALU reader (hipcc pads nothing around asm)
        // (opaque: the first attempt's thresholds do not change inside the attempt loop, and hipcc hoists the 32 compares
        //  out of it -- into 32 scalar register PAIRS it does not have: ~130 lane spills and refills per pass)
        const int th = opaque(mask_base[rb] - kb0);
        float mm = mx[rb];
#pragma unroll
        for (int i = 0; i < 16; ++i) mm = __builtin_fmaxf(mm, (i & 3) + 8 * (i >> 2) > th ? -INFINITY : s[i]);
        mx[rb] = mm;
      }
    };
    auto set_m = [&](const float (&mx)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        // a row that sees no key at all (rows past S_q) keeps a finite constant: its P is 0 everywhere, O is never stored
        const float m = half_max(mx[rb]);
        mrow[rb] = m > -INFINITY ? m : 0.f;
        nmc[rb] = FOLD ? -mrow[rb] : -mrow[rb] * c2;   // FOLD: scores are in log2 units already
        if constexpr (CHAIN_M) {
#pragma unroll
          for (int i = 0; i < 16; ++i) negm[rb][i] = -mrow[rb];
        }
      }
    };

    for (int attempt = 0; attempt < 2; ++attempt) {
      // ---- prologue: the first NBUF - 1 tiles on their way, tile 0 landed ----
      if (attempt == 1) {   // (the ring restarts at slot 0; the barrier of the redo branch below came first)
        r0 = 0;
#pragma unroll
        for (int t = 0; t < C::NBUF - 1; ++t) fetch_tile(t, t);
      }
      if (!was_primed || attempt == 1) {
        asm volatile("" ::: "memory");
        // tile 0 = the oldest 2 * PIECES pieces: leave the later tiles' pieces in flight
        __builtin_amdgcn_s_waitcnt(0x0070 | (((C::NBUF - 2) * 2 * C::PIECES) & 15) | ((((C::NBUF - 2) * 2 * C::PIECES) >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }   // (a primed pass: the previous end-of-pass check waited for tile 0 and held the barrier)

      FA4_STAMP(0);   // seg[0]: pass prologue -- Q fragments, ring primed, tile 0 landed (first barrier)
      float mx[2] = {-INFINITY, -INFINITY};
      if (attempt == 0) {
        // (bf16 could do without any row constant -- P = exp2(score) has the exponent range of fp32 -- and it was measured:
        //  +0.9 % causal.  But then the dominant term of a peaked row is no longer 1.0 exactly: its rounding to 16 bit no
        //  longer cancels against l, O picks up a second rounding (<= 2^-9), and a row with ONE visible key gets a dQ of
        //  5e-2 instead of exactly 0 (tests/test_gpu_fuzz.py).  The scout keeps that property for every row whose maximum
        //  lies in its first 32 keys, and costs 8 MFMAs per pass.)
        scout_block(smem + r0 * C::TILE_BYTES, 0, mx);   // keys 0..31: key 0 is visible to every row (top-left aligned mask)
        set_m(mx);
      } else {
        // exact row maxima: a max-only sweep over every tile (cold path: reached only after an overflow)
        for (int t = 0; t < ntiles; ++t) {
          if (t > 0) {
            __syncthreads();
            fetch_tile(t, t % C::NBUF);
            __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
            __syncthreads();
          }
          for (int j = 0; j < C::NKB; ++j)
            scout_block(smem + (t % C::NBUF) * C::TILE_BYTES + j * 32 * C::ROWB, t * C::BN + 32 * j, mx);
        }
        set_m(mx);
        // restart the ring
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C::NBUF - 1; ++t) fetch_tile(t, t);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070 | (((C::NBUF - 2) * 2 * C::PIECES) & 15) | ((((C::NBUF - 2) * 2 * C::PIECES) >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }

      FA4_STAMP(1);   // seg[1]: scout block + row constants
      // ---- pipeline fill: a neutral "previous key block" (P = 0), the first K fragments ----
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        l[rb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          // an element whose exp belongs to the block's own iteration (slot < NSLOT) is "done" (P = 0); the others are
          // still exponent arguments when the next iteration takes over: exp2(-inf) = 0
          const int q = (i >> 3) * 16 + rb * 8 + (i & 7);
          S_[1][rb][i] = C::NSS + q * C::NSLOT / 32 < C::NSLOT ? 0.f : -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) pk[1][rb][e] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int db = 0; db < C::DB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) oacc[rb][db][i] = 0.f;
        if constexpr (C::LSUM) {
#pragma unroll
          for (int i = 0; i < 16; ++i) lacc[rb][i] = 0.f;
        }
        thr[0][rb] = thr[1][rb] = 1 << 20;   // neutral: nothing is masked (plain blocks never touch their threshold)
      }
#pragma unroll
      for (int ks = 0; ks < KR; ++ks) KF[ks] = lds_read16(smem + r0 * C::TILE_BYTES + k_off[ks]);
#pragma unroll
      for (int pp = 0; pp < 2 * C::DB; ++pp) VF[pp] = as_vec8<T>(u32x4{0u, 0u, 0u, 0u});
      __builtin_amdgcn_sched_barrier(0);

      FA4_STAMP(2);   // seg[2]: pipeline fill
      int t = 0, rt = r0;
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      // (non-causal CONT: the last NBUF - 1 plain tile steps are written out below, with the next item's requests)
      const int nloop = (!CAUSAL && CONT) ? nplain - (C::NBUF - 1) : nplain;
      for (; t < nloop; ++t, rt = rt + 1 == C::NBUF ? 0 : rt + 1) tile_step(t, rt, I0{}, I0{}, I0{});
#ifdef FA_STAMPS
      FA4_STAMP(3);   // seg[3]: the plain tiles
      ntile_ += nplain;
      ++npass_;
#endif
      // The tiles level with the query tile (causal) and / or the ragged last tile.  A tile wholly inside S_k that lies in
      // the first half of the 256 keys level with the query tile masks row block 0 only (mode 2), one in the second half
      // has no row block 0 at all (mode 3); anything else masks both row blocks element-wise (mode 1).
      // The common causal case -- all 256 keys exist -- is written out straight-line: tile variants selected in a loop
      // merge control flow after every tile, and hipcc reconciles the register assignment of everything live (64-128
      // accumulator registers) with v_accvgpr_mov at each merge.
      constexpr int RT = 256 / C::BN;   // region tiles (2 or 4), half of them per mode
      auto nxt = [](int x) __attribute__((always_inline)) { return x + 1 == C::NBUF ? 0 : x + 1; };
      // (a masked tile treats its predecessor as masked too -- a plain predecessor carries the neutral threshold: fewer
      //  variants, and each variant costs registers in ALL paths, hipcc keeps one assignment of the pipeline state)
      if constexpr (CAUSAL) {
        // the launcher sends a causal shape here only if every query tile has all 256 keys level with it (S_k a multiple
        // of 256 and >= the padded S_q): no ragged tile, no generic path -- and fewer variants to keep registers for
        // (their DMA slots carry the NEXT pass's first NBUF - 1 tiles, the first of them its Q rows too)
        int rt_last;
        if constexpr (RT == 2) {
          tile_step(t, rt, I2{}, I1{}, I2{});
          tile_step(t + 1, rt_last = nxt(rt), I3{}, I1{}, I1{});
        } else {
          tile_step(t, rt, I2{}, I1{}, I0{});
          tile_step(t + 1, nxt(rt), I2{}, I1{}, I1{});
          tile_step(t + 2, nxt(nxt(rt)), I3{}, I1{}, I1{});
          tile_step(t + 3, rt_last = nxt(nxt(nxt(rt))), I3{}, I2{}, I1{});
        }
        static_assert(C::NBUF - 1 == (RT == 2 ? 2 : 3), "the masked tiles carry exactly the next pass's first NBUF - 1 tiles");
        FA4_STAMP(4);   // seg[4]: the masked tiles
        drain(I0{}, I2{}, rt_last);
        rt = rt_last;
      } else {
        // a ragged last tile masks both row blocks element-wise (a tile treats its predecessor as masked too: a plain
        // predecessor carries the neutral threshold)
        if (CONT) {   // whole tiles only: no ragged last tile in this branch
          if constexpr (C::NBUF == 3) {
            tile_step(t, rt, I0{}, I0{}, I2{});
            tile_step(t + 1, nxt(rt), I0{}, I0{}, I1{});
            rt = nxt(nxt(rt));
          } else {
            tile_step(t, rt, I0{}, I0{}, I1{});
            tile_step(t + 1, nxt(rt), I0{}, I0{}, I1{});
            tile_step(t + 2, nxt(nxt(rt)), I0{}, I0{}, I1{});
            rt = nxt(nxt(nxt(rt)));
          }
          t += C::NBUF - 1;
        }
        for (; t < ntiles; ++t, rt = nxt(rt)) tile_step(t, rt, I1{}, I1{}, I0{});
        rt = rt == 0 ? C::NBUF - 1 : rt - 1;   // the last tile's slot
        // the set of the last block: NKB is even, so it is always set 1 -> the drain's "previous" set is 1
        FA4_STAMP(4);
        drain(I0{}, I1{}, rt);
      }
      FA4_STAMP(5);   // seg[5]: drain

      // ---- end-of-pass check: every row sum finite and below the limit, for the whole workgroup ----
      float lt[2];
      bool bad = false;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        lt[rb] = C::LSUM ? lacc[C::LSUM ? rb : 0][0] : half_sum(l[rb]);   // LSUM: the MFMA summed over both lane halves' keys already
        bad = bad || !(lt[rb] <= Fwd4Limit<T>::value);
      }
      FA_LDS int* flag = (FA_LDS int*)(smem + C::FLAG_OFF);
      if (attempt == 0 && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *flag = 1;
      // the flag written, and every DMA retired but (CONT) the next pass's tiles 1 .. NBUF - 2, which target slots other
      // than the one the epilogue stages in (the last tile's, rt): its tile 0 and its Q rows have landed for the next pass
      if (CONT) __builtin_amdgcn_s_waitcnt(0x0070 | (((C::NBUF - 2) * 2 * C::PIECES) & 15) | ((((C::NBUF - 2) * 2 * C::PIECES) >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      const int redo = __builtin_amdgcn_readfirstlane(*flag);
      FA4_STAMP(6);   // seg[6]: end-of-pass check (row sums, flag, barrier)
      if (!redo) {
        // ---- epilogue: O = o / l, staged in the last tile's ring slot (K image for waves 0-1, V image for waves 2-3), which
        // the next pass's DMA reaches only after its first commit -- every wave is past its epilogue by then
        FA_LDS char* stage = smem + (wave >> 1) * C::V_BASE + rt * C::TILE_BYTES + (wave & 1) * 32 * C::ROWB;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const float inv = lt[rb] > 0.f ? 1.0f / lt[rb] : 0.f;
          store_tile_rows<D, T>(oacc[rb], inv, stage, ro, qrow(rb) * o_rs, lane, o_rs);
          if (h == 0)
            buf_store_f32(rl, (qrow(rb) + r) * 4, mrow[rb] * (FOLD ? kLn2 : p.scale) + __builtin_logf(lt[rb]));
        }
        FA4_STAMP(7);   // seg[7]: epilogue
        primed = has_next;
        r0 = rt + 1 == C::NBUF ? 0 : rt + 1;   // where the masked tiles put the next pass's tile 0
        break;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // (the next pass's tiles still in flight: the second attempt restarts the ring)
      __syncthreads();
      if (tid == 0) {
        *flag = 0;   // (ordered before the second attempt's check by the barriers of its sweep)
#ifndef FA_STAMPS   // tests count the passes that come here through the debug buffer (fa_debug_set_buffer; NULL otherwise):
        if (p.dbg) atomicAdd((unsigned*)p.dbg, 1u);   // no state in the library, and this is the cold path
#endif
      }
    }  // attempt
    // ---- on to the pair's second pass, or to the workgroup's next item ----
    if (more_pass) {
      ++pass;
    } else {
      item += gridDim.x;
      if (item >= n_items) break;
      wk = nwk_item;
      nwk_item = decode(min(item + (int)gridDim.x, n_items - 1));
      pass = 0;
    }
  }  // passes and items
#ifdef FA_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 4 + wave) * 16;
    for (int i = 0; i < 12; ++i) d[i] = seg[i];
    d[12] = npass_;
    d[13] = ntile_;
    unsigned long long clk1_, rt1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_), "=s"(rt1_)::"memory");
    d[14] = clk1_ - clk0_;
    d[15] = rt1_ - rt0_;
  }
#endif
}

template <int D, typename T, bool CAUSAL>
static hipError_t launch4(const FwdParams& p, hipStream_t s) {
  using C = Fwd4Cfg<D>;
  int grid = (CAUSAL && p.pair ? (p.nq_tiles + 1) / 2 : p.nq_tiles) * p.B * p.H;
  if (CAUSAL || (p.Sk % C::BN == 0 && p.Sk / C::BN >= C::NBUF - 1)) {   // persistent: one workgroup per CU, a multiple of 8 (fa_fwd4_kernel: the work list, CONT)
    static std::atomic<int> cus{0};
    int n = cus.load(std::memory_order_relaxed);
    if (n == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
      n -= n % 8;
      cus.store(n, std::memory_order_relaxed);
    }
    if (grid > n) grid = n;
  }
  auto kern = fa_fwd4_kernel<D, T, CAUSAL>;
  static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
  if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

// (fa_kernels.h pick_fwd_impl sends only shapes this family takes: fixed-length launches; causal ones only when every
// 256-row query tile has all the 256 keys level with it)
hipError_t launch_fwd_v4(FwdParams p, int D, int dtype, int causal, hipStream_t s) {
  p.nq_tiles = (p.Sq + 255) / 256;
  p.pair = causal != 0;
  p.div_per_bh = make_fastdiv(p.pair ? (p.nq_tiles + 1) / 2 : p.nq_tiles);
  p.div_h = make_fastdiv(p.H);
#define FA_GO(DD, TT) (causal ? launch4<DD, TT, true>(p, s) : launch4<DD, TT, false>(p, s))
  if (D == 64) return dtype == 1 ? FA_GO(64, BF16) : FA_GO(64, FP16);
  if (D == 128) return dtype == 1 ? FA_GO(128, BF16) : FA_GO(128, FP16);
#undef FA_GO
  return hipErrorInvalidValue;
}

}  // namespace fa
