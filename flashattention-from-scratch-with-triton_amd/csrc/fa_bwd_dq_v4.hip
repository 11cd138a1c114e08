// FlashAttention backward dQ (+ delta), fourth schedule family for gfx950 (head dim 64): ONE wave per SIMD, 64 query rows
// per wave, one continuous hand-ordered software pipeline.
//
// Same maths, rounding points and accumulation order as fa_bwd_dq.hip / fa_bwd_dq_v3.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:165-258; also writes delta, K:210-211): bit-identical results.
//
// Why a fourth family: family 3 (two waves per SIMD, 32 rows per wave) reads one LDS fragment per 0.74 MFMAs
// (SQ_INSTS_LDS / SQ_INSTS_MFMA = 1.35) and its two waves contend for one vector issue port.  Here a workgroup is 256
// query rows, a wave owns TWO 32-row blocks, and every K row fragment (S^T = K Q^T), V row fragment (dP^T = V dO^T) and
// K^T fragment (dQ^T += K^T dS^T) is read from LDS once and feeds both row blocks: 16 LDS instructions per 24 MFMAs.
// The resident operands of two row blocks (Q^T, dO^T fragments 64 registers, -LSE*log2e / -delta blocks 64, dQ^T
// accumulators 64, two score / dP sets 64, the fragment rings 48) need the whole 512-entry register file, so a SIMD
// holds one wave and the whole key loop is ONE software pipeline without fill / drain per tile (the method of
// fa_bwd_dkv_v3.hip):
//
//   block iteration (key block c, row block rb), 12 MFMA slots:
//     slots 0-3   S^T  = K Q^T      of this block     (VGPR-form asm MFMAs; chain starts from -LSE*log2e: C operand != D)
//     slots 4-7   dP^T = V dO^T     of this block     (chain starts from -delta)
//     slots 8-11  dQ^T += K^T dS^T  of the PREVIOUS block (builtin MFMAs accumulating in AGPRs)
//   and beside the MFMAs the exp2 / multiply / pack of the two blocks in flight by a fixed timetable (dq4_*_tau below:
//   at most two exps per slot, <= 24 issue cycles of vector work per 32-cycle MFMA), the LDS reads of fragments into
//   registers whose last use is just over (>= 8 slots ahead of their next use), each slot closed by sched_barrier(0).
//   K / V tiles of 128 keys arrive by LDS-DMA in a ring of three, two tiles ahead; the per-tile commit (counted vmcnt,
//   s_barrier) sits inside the tile's last iteration, before the first read of the next tile.
//
// Causal: workgroups take the query-tile pair (nq-1-i, i).  A wave owns row blocks {w, 7-w} of the 256 rows, so the 8 x 8
// blocks level with the query tile cost every wave the same 9 visits.  Those 256 keys (two tiles) are resident before
// that phase starts, which then needs no barrier and runs per wave: key blocks 0..w for both row blocks, key blocks
// w+1..7-w for row block 1 alone ("solo" iterations); blocks above the diagonal are never visited, and the mask of the two
// diagonal blocks is FREE: their score chains start from (dead ? -inf : -LSE*log2e) instead of the row constant.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dq4Cfg {
  static constexpr int D = 64;
  static constexpr int BM = 256, BN = 128, NT = 256, NW = 4;
  static constexpr int NKB = BN / 32;                       // 32-key blocks per tile
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int KBLK = 32 * ROWB;                    // bytes of one 32-key block in a tile image
  static constexpr int TILE_BYTES = BN * ROWB;              // 16 KiB
  static constexpr int NBUF = 3;
  static constexpr int V_BASE = NBUF * TILE_BYTES;          // K[NBUF], then V[NBUF]
  // behind the ring: the NEXT pass's Q and dO rows, 64 per wave (row blocks w and 7 - w, 32 rows x 128 B each), staged by
  // LDS-DMA; its O rows are staged in the ring slot that is free when they arrive (the kernel says where)
  static constexpr int QS_OFF = 2 * NBUF * TILE_BYTES, DOS_OFF = QS_OFF + BM * ROWB;
  static constexpr int RB_BYTES = 32 * ROWB;                // one row block's rows: 4 KiB = 4 pieces
  static constexpr int LDS_BYTES = DOS_OFF + BM * ROWB;     // 160 KiB: the whole LDS of a CU
  static constexpr int PIECES = TILE_BYTES / (NW * 1024);   // 1-KiB LDS-DMA pieces per wave per matrix (4)
  static constexpr int RPI = 1024 / ROWB;                   // tile rows per piece
  static constexpr int NS = 2 * KS + 2 * DB;                // MFMA slots per block iteration (12)
  static constexpr int kOOB = 0x7FFFFFFF;                   // scalar offset of a fetch past the last tile: out of range, no traffic
};

// Timetable of one block's vector work, in slots since the start of its own iteration (tau).  The S chain ends at slot 3,
// the dP chain at slot 7; its dQ MFMAs are slots 20-23 (k-step 0 first).  Issue cost per slot (exp 8, mul / pack 4
// cycles) of the two blocks in flight together: 24 20 24 24 16 20 16 16 16 16 16 16 of the 24 that hide behind an MFMA.
constexpr int dq4_exp_tau(int e) { return e < 14 ? 5 + e / 2 : 12; }
constexpr int dq4_mul_tau(int m) { return m < 2 ? 12 : 13 + (m - 2) / 4; }
constexpr int dq4_cvt_tau(int j) { return j == 0 ? 13 : 14 + (j - 1) / 2; }

// -DFA_STAMPS (diagnostic build, tools/stamps_dq4.py): per-phase cycle account of a wave, written to BwdParams::dbg.
// seg[0] pass prologue (ring primed, resident operands, first barrier, fill)   seg[1] unmasked tiles   seg[2] diagonal phase
// seg[3] drain   seg[4] epilogue   seg[5..7] parts of the prologue (seg[0] is then the fill alone)   -DFA_STAMPS_ITER adds seg[8 + I] block iteration I of an unmasked tile, seg[16] the commit
#ifdef FA_STAMPS
#define FA4Q_STAMP(slot)                                                          \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long now_;                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += now_ - last_;                                                    \
    last_ = now_;                                                                 \
  } while (0)
#else
#define FA4Q_STAMP(slot) do {} while (0)
#endif
#ifdef FA_STAMPS_ITER
#define FA4Q_ISTAMP(slot) FA4Q_STAMP(slot)
#else
#define FA4Q_ISTAMP(slot) do {} while (0)
#endif

// Where a tile's eight LDS-DMA pieces per wave are issued.  Grouped (the dK/dV family-3 placement): two pairs in the last
// iteration of a tile step, two in the first of the next.  Spread (A/B hook, OFF): one piece per block iteration, on the theory
// that four waves x four pieces at once queue up in the CU's one address unit (the two iterations that carry them stamp
// 250-300 cycles above the others) -- measured 0.4135 vs 0.3463 ms at the headline, 0.7138 vs 0.6533 non-causal: every
// separate issue point costs more than the queue does (profiles/r04_ab_lines.txt).
#ifndef FA_DQ4_DMA_SPREAD
#define FA_DQ4_DMA_SPREAD 0
#endif
#ifndef FA_DQ4_SPREAD_SLOT
#define FA_DQ4_SPREAD_SLOT 2
#endif
// A/B hook (OFF): a barrier after every block visit of the diagonal phase (every wave makes nine) -- does lockstep matter?
#ifdef FA_DQ4_DIAG_BARRIER
#define FA_DQ4_DIAG_SYNC() __builtin_amdgcn_s_barrier()
#else
#define FA_DQ4_DIAG_SYNC() do {} while (0)
#endif
constexpr bool kDq4Spread = FA_DQ4_DMA_SPREAD != 0;   // (2: the four pairs in the MIDDLE of a tile step, iterations 2 and 4)
constexpr bool kDq4Mid = FA_DQ4_DMA_SPREAD == 2;
constexpr int kDq4SpreadSlot = FA_DQ4_SPREAD_SLOT;

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void fa_bwd_dq4_kernel(BwdParams p) {
#ifdef FA_STAMPS
  unsigned long long clk0_, rt0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_)::"memory");
  unsigned long long seg[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = clk0_, ntile_ = 0, npass_ = 0;
#endif
  using C = Dq4Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // Work list: one item = one 256-row query tile, or (causal, paired) the tile pair (nq-1-i, i) -> equal work per item (see
  // fa_fwd.hip); the XCD-aware order gives every XCD a contiguous run of (batch, head) slices.  Causal launches are
  // PERSISTENT: one workgroup per CU walks items blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8 keeps all
  // of them on one XCD's slice list), because a pass's resident operands are fetched during the PREVIOUS pass (below) and the
  // first pass of an item has no previous pass unless the workgroup stays.
  const bool paired = CAUSAL && p.pair;
  const int nq = p.n_tiles;
  const int per_bh = paired ? (nq + 1) / 2 : nq;
  const int n_items = per_bh * p.B * p.H;
  const int Sq = p.Sq, Sk = p.Sk;
  // Q, K, V, dO, O, dQ may be strided views with a contiguous head dim (fa_fwd.hip); no variable-length launches here
  const int q_rs = p.lq.rs, do_rs = p.ldo.rs, kv_rs = p.lk.rs, o_rs = p.lo.rs, dq_rs = p.ldq.rs;
  const float c2 = p.scale * kLog2e;
  const int lds0 = (int)lds_addr_of(smem);
  pin_reserve();

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
  // The resident operands of a pass -- Q, dO (MFMA B operands) and O (delta) of this wave's 64 rows, and their LSE -- are
  // STAGED through LDS a whole phase before the pass needs them: the rows are cold (nobody else reads them), every workgroup
  // of a launch starts a pass at about the same moment, and fetched as fragments -- 26 loads per lane, each touching 32
  // cache lines -- they cost a wave 8k cycles of a 75k-cycle pass wherever they are issued (the CU's address unit takes ~63
  // cycles per such load: stamps, tools/stamps_dq4.py; without the traffic the kernel ran 11 % faster).  As LDS-DMA pieces
  // (8 whole rows each, the K / V tile machinery) they cost ~27 cycles of the address unit apiece, no register, and ride in
  // slots of the previous pass's last block iterations; the fragments are then ds_read_b128 row reads like K's.
  // Six groups of four pieces per wave: g = 0, 1 Q rows of row block 0, 1 -> QS; 2, 3 dO -> DOS; 4 O of row block 0 -> this
  // wave's rows of the K half of ring slot `oslot`, 5 O of row block 1 -> its rows of the V half: exactly the 2 x 4 KiB this
  // wave fills itself with the pass's third tile, once it has consumed them -- no other wave ever touches them.
  struct Stage {
    __amdgpu_buffer_rsrc_t rq, rdo, ro, rl;
    int row0[2];
  };
  auto stage_of = [&](const Work& wk, int pass, bool valid) __attribute__((always_inline)) -> Stage {
    const int q0 = tile_of(wk, pass) * C::BM;
    // an invalid stage (nothing follows this pass) has empty descriptors: its pieces fetch nothing
    const unsigned nb = valid ? view_bytes(Sq, q_rs, C::ROWB) : 0u, nbd = valid ? view_bytes(Sq, do_rs, C::ROWB) : 0u;
    const unsigned nbo = valid ? view_bytes(Sq, o_rs, C::ROWB) : 0u, nbl = valid ? (unsigned)Sq * 4 : 0u;
    return Stage{make_rsrc((const char*)p.q + wk.b * p.lq.sb + wk.h * p.lq.sh, nb),
                 make_rsrc((const char*)p.dout + wk.b * p.ldo.sb + wk.h * p.ldo.sh, nbd),
                 make_rsrc((const char*)p.o + wk.b * p.lo.sb + wk.h * p.lo.sh, nbo),
                 make_rsrc(p.lse + wk.b * p.lse_sb + wk.h * p.lse_sh, nbl),
                 {q0 + 32 * wave, q0 + 32 * (7 - wave)}};
  };
  auto stage_group = [&](const Stage& st, int oslot, int g) __attribute__((always_inline)) {
    const int ln = lane_id_now(), prow = ln >> 3;                     // piece i holds rows 8 i + prow of the row block
    const int rb = g & 1, rs = g < 2 ? q_rs : (g < 4 ? do_rs : o_rs);
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)   // (dma_pieces: the immediate 1024 i moves LDS and global address together: taken out here)
      voff[i] = (st.row0[rb] + 8 * i + prow) * rs + swz_chunk<D>(8 * i + prow, ln & 7) * 16 - 1024 * i;
    const int dst = g < 4 ? (g < 2 ? C::QS_OFF : C::DOS_OFF) + (2 * wave + rb) * C::RB_BYTES
                          : (rb ? C::V_BASE : 0) + oslot * C::TILE_BYTES + wave * C::RB_BYTES;
    dma_pieces<4>(g < 2 ? st.rq : (g < 4 ? st.rdo : st.ro), (unsigned)(lds0 + dst), voff, 0);
  };
  auto stage_lse = [&](const Stage& st) __attribute__((always_inline)) {   // the two LSE rows of this lane: a[128], a[129]
    const int r = lane_id_now() & 31;
    pf_load4<kPfLse>(st.rl, (st.row0[0] + r) * 4);
    pf_load4<kPfLse + 1>(st.rl, (st.row0[1] + r) * 4);
  };

  int item = blockIdx.x;
  Work wk = decode(item);
  Work nwk_item = decode(min(item + (int)gridDim.x, n_items - 1));   // the workgroup's next item, decoded once per item
  int b0 = 0, b1 = 1, b2 = 2;   // ring slots of tiles t, t + 1, t + 2; they keep rotating from pass to pass
  bool staged = false;          // this pass's rows are on their way (issued from inside the previous pass)
  for (; item < n_items; item += gridDim.x, wk = nwk_item, nwk_item = decode(min(item + (int)gridDim.x, n_items - 1))) {
  const int b_ = wk.b, h_ = wk.h, npass = wk.npass;
  const __amdgpu_buffer_rsrc_t rdq = make_rsrc((char*)p.dq + b_ * p.ldq.sb + h_ * p.ldq.sh, view_bytes(Sq, dq_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rk = make_rsrc((const char*)p.k + b_ * p.lk.sb + h_ * p.lk.sh, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc((const char*)p.v + b_ * p.lv.sb + h_ * p.lv.sh, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.delta + b_ * p.lse_sb + h_ * p.lse_sh, (unsigned)Sq * 4);

  for (int pass = 0; pass < npass; ++pass) {
#ifndef FA_STAMPS_ITER
    FA4Q_STAMP(9);    // seg[9]: loop bookkeeping (item decode, descriptors) since the end of the previous epilogue
#endif
    // lane coordinates re-derived per pass (fa_common.h lane_id_now): nothing lane-dependent stays live across passes
    const int lane = lane_id_now(), r = lane & 31, h = lane >> 5;
    const int q0_wg = tile_of(wk, pass) * C::BM;
    // this wave's two 32-row blocks: {w, 7-w} of the workgroup's eight (equal causal work per wave, see the header)
    const int qrow[2] = {q0_wg + 32 * wave, q0_wg + 32 * (7 - wave)};

    // tiles [0, nfull) are unmasked for every row of the workgroup; causal launches add the two tiles level with the
    // query tile (the launcher guarantees that they exist: S_k is a multiple of 256 and covers every query tile)
    const int nfull = CAUSAL ? q0_wg / C::BN : Sk / C::BN;
    const int ntot = nfull + (CAUSAL ? 2 : 0);

    // ---- LDS-DMA: a wave fills rows [32w, 32w+32) of each K and V tile, 2 x 4 pieces, issued in pairs ----
    int dma_src[C::PIECES];
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
      const int row = (C::BN / C::NW) * wave + C::RPI * i + lane / C::CPR;
      dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16 - 1024 * (i & 1);  // dma_pieces: immediate taken out
    }
    // group g4: 0, 1 = the K pairs, 2, 3 = the V pairs of this wave's share of tile t (ring slot `buf`); a tile past the
    // last one is out of range for the descriptor (no branch: hipcc sinks code across branches, fa_bwd_dkv_v3.hip)
    auto dma_group = [&](int t, int buf, int g4) __attribute__((always_inline)) {
      const int i = 2 * (g4 & 1);
      const int dst = buf * C::TILE_BYTES + ((C::BN / C::NW) * wave + C::RPI * i) * C::ROWB;
      const int soff = t < ntot ? t * C::BN * kv_rs : C::kOOB;
      if (g4 < 2) dma_pieces<2>(rk, (unsigned)(lds0 + dst), dma_src + i, soff);
      else dma_pieces<2>(rv, (unsigned)(lds0 + C::V_BASE + dst), dma_src + i, soff);
    };
    // ---- resident B operands: Q^T and dO^T fragments of both row blocks; delta (K:210-211, from the rounded O) ----
    // They live in the pinned accumulator registers of fa_common.h (pin_write / MfmaPin): fragment F = 8 rb + ks holds
    // Q^T k-step ks of row block rb, F = 8 rb + 4 + ks its dO^T -- everything else a VGPR-form asm MFMA touches must sit in the
    // 256 ARCHITECTURAL registers (C and D share one AGPR bit), which these 64 registers would overflow.
    float nl[2];       // -LSE * log2e of this lane's row
    f32x16 NL[FOLD ? 2 : 1], ND[2];   // the same / -delta in every register: C operands of the chain starts
    if (!staged) {   // the first pass of a workgroup (and every pass of a non-causal launch): nothing to hide behind
      const Stage st = stage_of(wk, pass, true);
#pragma unroll
      for (int g = 0; g < 6; ++g) stage_group(st, b2, g);
      stage_lse(st);
    }
    // this wave's staged rows have landed: everything but the previous pass's epilogue stores, which are younger -- 8 dQ
    // stores (+ 8 of the scaled-Q workspace) -- and whose write acknowledgements are not worth 1-2k cycles of waiting
    if (!staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (FOLD && p.qs) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifndef FA_STAMPS_ITER
    FA4Q_STAMP(10);   // seg[10]: lane addresses, the wait for the staged rows
#endif
    // the ring: tiles 0 and 1 now (their address-unit time runs under the arithmetic below); tile 2's first half follows once
    // this wave has consumed the O rows that sit in its part of slot b2, the second half rides in the first tile step
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) dma_group(0, b0, g4);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) dma_group(1, b1, g4);
    FA4Q_STAMP(5);   // seg[5]: staged rows landed, tiles 0 and 1 requested
    __builtin_amdgcn_sched_barrier(0);
    int row_off[C::KS];   // A-operand row reads (K rows and V rows; here: the staged Q / dO / O rows)
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
    auto load_rows = [&](auto rb_tag) __attribute__((always_inline)) {
      constexpr int rb = decltype(rb_tag)::value;
      float dsum = 0.f;
      vec8 qv[C::KS], dv[C::KS];
      const FA_LDS char* qs = smem + C::QS_OFF + (2 * wave + rb) * C::RB_BYTES;
      const FA_LDS char* ds = smem + C::DOS_OFF + (2 * wave + rb) * C::RB_BYTES;
      const FA_LDS char* os = smem + (rb ? C::V_BASE : 0) + b2 * C::TILE_BYTES + wave * C::RB_BYTES;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        qv[ks] = as_vec8<T>(lds_read16(qs + row_off[ks]));
        dv[ks] = as_vec8<T>(lds_read16(ds + row_off[ks]));
        const vec8 ov = as_vec8<T>(lds_read16(os + row_off[ks]));
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum = __builtin_fmaf((float)dv[ks][j], (float)ov[j], dsum);
      }
      const float delta = half_sum(dsum);
      nl[rb] = -acc_read1<kPfLse + rb>() * kLog2e;
      if (h == 0) buf_store_f32(rd, (qrow[rb] + r) * 4, delta);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        ND[rb][i] = -delta;
        if constexpr (FOLD) NL[rb][i] = nl[rb];
      }
      if constexpr (FOLD) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qv[ks] = scale_frag<T>(qv[ks], c2);
      }
      pin_write<8 * rb + 0>(__builtin_bit_cast(u32x4, qv[0]));
      pin_write<8 * rb + 1>(__builtin_bit_cast(u32x4, qv[1]));
      pin_write<8 * rb + 2>(__builtin_bit_cast(u32x4, qv[2]));
      pin_write<8 * rb + 3>(__builtin_bit_cast(u32x4, qv[3]));
      pin_write<8 * rb + 4>(__builtin_bit_cast(u32x4, dv[0]));
      pin_write<8 * rb + 5>(__builtin_bit_cast(u32x4, dv[1]));
      pin_write<8 * rb + 6>(__builtin_bit_cast(u32x4, dv[2]));
      pin_write<8 * rb + 7>(__builtin_bit_cast(u32x4, dv[3]));
    };
    load_rows(std::integral_constant<int, 0>{});
    load_rows(std::integral_constant<int, 1>{});
    // the O rows are consumed (every LDS read above has fed arithmetic): this wave's part of slot b2 takes tile 2
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!kDq4Spread) {
      dma_group(2, b2, 0);
      dma_group(2, b2, 1);
    }
    FA4Q_STAMP(6);   // seg[6]: delta, scaled, pinned; tile 2's first half requested
    asm volatile("s_nop 4");  // v_accvgpr_write -> MFMA operand wait states (hipcc pads nothing around asm)

    // ---- loop-invariant per-lane LDS offsets ----
    int tr_off[2][C::DB];  // transposed reads of K
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int db = 0; db < C::DB; ++db) tr_off[x][db] = tr_lane_off<D>(lane, 8 * x, db);

    // dQ^T accumulators: block (row block rb, d block db) = a[16 (2 rb + db) ..] of the pinned file
    static_for<2 * C::DB>([](auto i_) __attribute__((always_inline)) { acc_zero16<kAccBase + 16 * decltype(i_)::value>(); });

    // ---- pipeline state ----
    f32x16 S_[2], P_[2];   // [set]: score / dP accumulators of the block in flight; the other set holds the previous
                           // block's exponent arguments -> P and dP - delta -> dS
    u32x4 sk[2][2];        // [set][k-step]: packed dS of a block
    u32x4 KR[C::KS], VR[C::KS];   // K / V row fragments of the key block being scored (shared by both row blocks)
    vec8 KT[2 * C::DB];    // K^T fragments (k-step e, d block db) -> index 2 e + db, of the previous block's key block
    f32x16 CD;             // chain start of a diagonal block: dead ? -inf : (-LSE*log2e | 0)

    // VALU work of ONE block at pipeline time tau (the timetable above)
    auto block_valu = [&](int tau, f32x16& X, f32x16& Y, u32x4 (&skb)[2], float nlr) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (dq4_exp_tau(e) == tau) X[e] = __builtin_amdgcn_exp2f(FOLD ? X[e] : __builtin_fmaf(X[e], c2, nlr));
#pragma unroll
      for (int m = 0; m < 16; ++m)
        if (dq4_mul_tau(m) == tau) Y[m] = X[m] * Y[m];   // dS^T = P^T o (dP^T - delta)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (dq4_cvt_tau(j) == tau) skb[j >> 2][j & 3] = pack2<T>(Y[2 * j], Y[2 * j + 1]);
    };

    // dQ^T MFMA n = 2 e + db of a block of row block RB: (k-step e, d block db); sk[.][0] is complete first
    auto dq_mfma = [&](auto rb_tag, int n, const u32x4 (&skb)[2]) __attribute__((always_inline)) {
      constexpr int RB = decltype(rb_tag)::value;
      const u32x4 kt = __builtin_bit_cast(u32x4, KT[n]);
      if (n == 0) MfmaPin::template into<kAccBase + 32 * RB>((T*)nullptr, kt, skb[0]);
      else if (n == 1) MfmaPin::template into<kAccBase + 32 * RB + 16>((T*)nullptr, kt, skb[0]);
      else if (n == 2) MfmaPin::template into<kAccBase + 32 * RB>((T*)nullptr, kt, skb[1]);
      else MfmaPin::template into<kAccBase + 32 * RB + 16>((T*)nullptr, kt, skb[1]);
    };
    // One block iteration.  G: accumulator set of this block (the previous block's is G ^ 1); RB: its row block; PRB: the
    // previous block's row block; KTR: reload the K^T fragments (each right after the dQ MFMA that read the old one) from
    // `tb[x][db] + t_imm` = THIS key block; KRV: reload the K / V row fragments from `kb[ks] + k_imm` = the NEXT key
    // block (K rows under slots 4-7, V rows under 8-11; a solo iteration does both reloads); DIAG: the score chain
    // starts from CD.  hook(s, 0) runs before the slot's MFMA (the commit), hook(s, 1) right after it (LDS-DMA).
    auto block_iter = [&](auto g_tag, auto rb_tag, auto prb_tag, auto ktr_tag, auto krv_tag, auto diag_tag,
                          const int (&tb)[2][C::DB], int t_imm, const int (&kb)[C::KS], int k_imm,
                          auto&& hook) __attribute__((always_inline)) {
      constexpr int G = decltype(g_tag)::value, PG = G ^ 1, RB = decltype(rb_tag)::value, PRB = decltype(prb_tag)::value;
      constexpr bool KTR = decltype(ktr_tag)::value, KRV = decltype(krv_tag)::value, DIAG = decltype(diag_tag)::value;
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        hook(s, 0);
        // ---- the MFMA of this slot ----
        if (s == 0) {
          if constexpr (DIAG) MfmaPin::template first<8 * RB>((T*)nullptr, S_[G], KR[0], CD);
          else if constexpr (FOLD) MfmaPin::template first<8 * RB>((T*)nullptr, S_[G], KR[0], NL[FOLD ? RB : 0]);
          else MfmaPin::template first0<8 * RB>((T*)nullptr, S_[G], KR[0]);
        } else if (s == 1) { MfmaPin::template acc<8 * RB + 1>((T*)nullptr, S_[G], KR[1]);
        } else if (s == 2) { MfmaPin::template acc<8 * RB + 2>((T*)nullptr, S_[G], KR[2]);
        } else if (s == 3) { MfmaPin::template acc<8 * RB + 3>((T*)nullptr, S_[G], KR[3]);
        } else if (s == 4) { MfmaPin::template first<8 * RB + 4>((T*)nullptr, P_[G], VR[0], ND[RB]);
        } else if (s == 5) { MfmaPin::template acc<8 * RB + 5>((T*)nullptr, P_[G], VR[1]);
        } else if (s == 6) { MfmaPin::template acc<8 * RB + 6>((T*)nullptr, P_[G], VR[2]);
        } else if (s == 7) { MfmaPin::template acc<8 * RB + 7>((T*)nullptr, P_[G], VR[3]);
        } else {   // (k-step e, d block db) = (n >> 1, n & 1): sk[.][0] is complete first
          dq_mfma(std::integral_constant<int, PRB>{}, s - 2 * C::KS, sk[PG]);
        }
        hook(s, 1);
        // ---- LDS reads into registers whose last use is just over ----
        if constexpr (KRV) {
          if (s >= C::KS && s < 2 * C::KS) KR[s - C::KS] = lds_read16(lds_at(kb[s - C::KS] + k_imm));
          if (s >= 2 * C::KS) VR[s - 2 * C::KS] = lds_read16(lds_at(kb[s - 2 * C::KS] + k_imm + C::V_BASE));
        }
        if constexpr (KTR) {
          if (s >= 2 * C::KS) {
            const int n = s - 2 * C::KS, e = n >> 1, db = n & 1;
            KT[n] = lds_read_tr_frag<T>(lds_at(tb[0][db] + t_imm + 16 * e * C::ROWB), lds_at(tb[1][db] + t_imm + 16 * e * C::ROWB));
          }
        }
        // ---- VALU: the previous block at tau = 12 + s, this block at tau = s ----
        block_valu(C::NS + s, S_[PG], P_[PG], sk[PG], nl[PRB]);
        block_valu(s, S_[G], P_[G], sk[G], nl[RB]);
        // An MFMA reads its C operand over its passes and hipcc pads that hazard for its own MFMAs only: a chain-start block
        // that is DEAD after this use (CD always; the row constants in the last iteration that uses them) would have its
        // registers reused for exp / multiply results at once (fa_bwd_dkv_v3.hip; seen here as a wrong row block 1 in the
        // fp16 causal kernel, -delta overwritten under the last dP chain start) -- keep them live for one more slot
        if (s == 1) {
          if constexpr (DIAG) keep_live(CD);
          else if constexpr (FOLD) keep_live(NL[FOLD ? RB : 0]);
        }
        if (s == C::KS + 1) keep_live(ND[RB]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the last block's remaining exp / dS and its dQ MFMAs once nothing follows it
    auto pipe_drain = [&](auto g_tag, auto rb_tag) __attribute__((always_inline)) {
      constexpr int G = decltype(g_tag)::value, RB = decltype(rb_tag)::value;
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        if (s >= 2 * C::KS) dq_mfma(std::integral_constant<int, RB>{}, s - 2 * C::KS, sk[G]);
        block_valu(C::NS + s, S_[G], P_[G], sk[G], nl[RB]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto no_hook = [](int, int) __attribute__((always_inline)) {};
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Yes = std::true_type;
    using No = std::false_type;

    // per-lane read bases of the key block at LDS byte offset `off` (opaque: otherwise hipcc hoists every (lane offset +
    // constant) pair out of the tile loop and parks them in AGPRs, fa_fwd_v4.hip)
    auto row_bases = [&](int (&kb)[C::KS], int off) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) kb[ks] = opaque(lds0 + row_off[ks] + off);
    };
    auto tr_bases = [&](int (&tb)[2][C::DB], int off) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int db = 0; db < C::DB; ++db) tb[x][db] = opaque(lds0 + tr_off[x][db] + off);
    };

    // ---- first tiles landed (hipcc waited vmcnt(0) for the fragment loads above, which are younger) ----
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0), lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    FA4Q_STAMP(7);   // seg[7]: first barrier
    // ---- fill: the first iteration's "previous block" is neutral (P = 0, dS = 0, zero fragments) ----
    {
      // A zero hipcc cannot see through (the lane id is below 64).  With constants it folds the neutral block's whole
      // arithmetic wherever the first tile step is a copy of its own (few tiles: the written-out steps below), keeps the
      // resulting zero fragments in a scalar register and splats them into place right in front of the asm MFMA that reads
      // them -- no wait state between (tools/mfma_lint.py rule R1).
      const unsigned lz = (unsigned)lane_id_now() >> 6;
      const float lzf = __builtin_bit_cast(float, lz);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        S_[1][i] = i < 14 ? lzf : -INFINITY + lzf;   // exps 0-13 of a block are over when its iteration ends; 14, 15 follow
        P_[1][i] = lzf;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) sk[1][e] = u32x4{lz, lz, lz, lz};
#pragma unroll
      for (int n = 0; n < 2 * C::DB; ++n) KT[n] = as_vec8<T>(u32x4{lz, lz, lz, lz});
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        KR[ks] = lds_read16(smem + b0 * C::TILE_BYTES + row_off[ks]);
        VR[ks] = lds_read16(smem + C::V_BASE + b0 * C::TILE_BYTES + row_off[ks]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    FA4Q_STAMP(0);
    // ---- the unmasked tiles: eight block iterations per tile, ring slots rotate ----
    // Non-causal launches are persistent too, and a pass has no diagonal phase to stage the next item's rows from: they ride
    // in the LDS-DMA slots of its last three tile steps, which have no tile left to fetch -- SEL 1 (tile nfull - 3): the O rows
    // into the ring slot of the tile that does not exist (free since this step's commit; three rotations later it is the next
    // pass's b2), SEL 2: Q and dO into their staging areas, SEL 3: the LSE.  The commits of those steps count the staging
    // pieces in flight on top of the tile pieces they have always left in flight.
    const bool nc_stage = !CAUSAL && item + (int)gridDim.x < n_items && nfull >= 3;
    auto tile_step = [&](int t, auto sel_tag) __attribute__((always_inline)) {
      constexpr int SEL = decltype(sel_tag)::value;
      // (defined HERE: a descriptor captured through two levels of closures goes through memory and comes back as a vector)
      const Stage est = stage_of(nwk_item, 0, SEL != 0);
      int kA[C::KS], kN[C::KS], tA[2][C::DB];
      row_bases(kA, b0 * C::TILE_BYTES);
      row_bases(kN, b1 * C::TILE_BYTES);
      tr_bases(tA, b0 * C::TILE_BYTES);
      // tile t + 2's V pairs ride in the first iteration (its K pairs went out in the previous tile's last one)
      auto hook_first = [&](int s, int phase) __attribute__((always_inline)) {
        if constexpr (SEL == 0 || SEL == 1) {
          if (phase == 1 && s == 1) dma_group(t + 2, b2, 2);
          if (phase == 1 && s == 5) dma_group(t + 2, b2, 3);
        } else if constexpr (SEL == 2) {
          if (phase == 1 && s == 1) stage_group(est, b2, 0);
          if (phase == 1 && s == 5) stage_group(est, b2, 1);
        } else {
          if (phase == 1 && s == 1) stage_lse(est);
        }
      };
      // the commit: tile t + 1 has landed for every wave (vmcnt(8): the eight pieces of tile t + 2 may still fly) and every
      // wave's reads of tile t are complete (the youngest, K^T of its last key block, are four slots old: lgkmcnt(0) is
      // free here) -- the ring slot of tile t then takes tile t + 3
      auto hook_last = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 0 && s == C::KS) {
          FA4Q_ISTAMP(15);
          asm volatile("" ::: "memory");
          if constexpr (SEL == 2) __builtin_amdgcn_s_waitcnt(0x4070);        // vmcnt(16): 8 + 8 staging pieces younger than tile t + 1
          else if constexpr (SEL == 3) __builtin_amdgcn_s_waitcnt(0x4072);   // vmcnt(18): nothing follows; staging stays in flight
          else __builtin_amdgcn_s_waitcnt(0x0078);                           // vmcnt(8), lgkmcnt(0)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          FA4Q_ISTAMP(16);
        }
        if constexpr (SEL == 0) {
          if (phase == 1 && s == 5) dma_group(t + 3, b0, 0);
          if (phase == 1 && s == 9) dma_group(t + 3, b0, 1);
        } else if constexpr (SEL == 1) {   // slot b0 (tile t) is free from the commit above on: the O rows of the next item
          if (phase == 1 && s == 5) stage_group(est, b0, 4);
          if (phase == 1 && s == 9) stage_group(est, b0, 5);
        } else if constexpr (SEL == 2) {
          if (phase == 1 && s == 5) stage_group(est, b0, 2);
          if (phase == 1 && s == 9) stage_group(est, b0, 3);
        }
      };
      block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, No{}, tA, 0 * C::KBLK, kA, 0, hook_first);
      FA4Q_ISTAMP(8);
      block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tA, 0, kA, 1 * C::KBLK, no_hook);
      FA4Q_ISTAMP(9);
      block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, No{}, tA, 1 * C::KBLK, kA, 0, no_hook);
      FA4Q_ISTAMP(10);
      block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tA, 0, kA, 2 * C::KBLK, no_hook);
      FA4Q_ISTAMP(11);
      block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, No{}, tA, 2 * C::KBLK, kA, 0, no_hook);
      FA4Q_ISTAMP(12);
      block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tA, 0, kA, 3 * C::KBLK, no_hook);
      FA4Q_ISTAMP(13);
      block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, No{}, tA, 3 * C::KBLK, kA, 0, no_hook);
      FA4Q_ISTAMP(14);
      block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tA, 0, kN, 0, hook_last);
      FA4Q_ISTAMP(15);
#ifdef FA_STAMPS
      ++ntile_;
#endif
      const int bt = b0;
      b0 = b1;
      b1 = b2;
      b2 = bt;
    };
    {
      int t = 0;
      const int nmain = nc_stage ? nfull - 3 : nfull;
      for (; t < nmain; ++t) tile_step(t, I0{});
      if constexpr (!CAUSAL) {
        if (nc_stage) {
          tile_step(t, I1{});
          tile_step(t + 1, std::integral_constant<int, 2>{});
          tile_step(t + 2, std::integral_constant<int, 3>{});
        }
      }
    }

    FA4Q_STAMP(1);
    if constexpr (!CAUSAL) {
      pipe_drain(I1{}, I1{});
      if (nc_stage) {
        // the O rows went into SEL 1's b0, which three rotations have made b0 again: one more and it is b2, where the next
        // pass looks for them, and b0 -- where the epilogue below stages dQ -- is the slot of the last tile, which no DMA targets
        const int bt = b0;
        b0 = b1;
        b1 = b2;
        b2 = bt;
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the (out-of-range) fetches past the last tile are over before the rings are reused
      }
      staged = nc_stage;
    } else {
      // ---- the 256 keys level with the query tile: tiles nfull (ring slot b0, landed) and nfull + 1 (b1, in flight) ----
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0), lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // No vmcnt wait from here to the end of the pass and ring slot b2 is free: the place to stage the NEXT pass's rows (the
      // pair's second pass, or the first pass of the workgroup's next item).  The six piece groups ride in the three block
      // iterations below that every wave executes exactly once; with nothing to follow the descriptors are empty.
      const bool more_pass = pass + 1 < npass, more_item = item + (int)gridDim.x < n_items;
      const Stage nst = stage_of(more_pass ? wk : nwk_item, more_pass ? pass + 1 : 0,
                                 more_pass || more_item);
      auto hook_a = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 2) stage_group(nst, b2, 0);
        if (phase == 1 && s == 8) stage_group(nst, b2, 1);
      };
      auto hook_b = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 2) stage_group(nst, b2, 2);
        if (phase == 1 && s == 8) stage_group(nst, b2, 3);
      };
      auto hook_c = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 2) stage_group(nst, b2, 4);
        if (phase == 1 && s == 6) stage_group(nst, b2, 5);
        if (phase == 1 && s == 10) stage_lse(nst);
      };
      staged = more_pass || more_item;
      // LDS byte offset of key block c = 0..7 of that region
      auto cbase = [&](int c) __attribute__((always_inline)) {
        return (c < C::NKB ? b0 * C::TILE_BYTES : b1 * C::TILE_BYTES - C::NKB * C::KBLK) + c * C::KBLK;
      };
      // chain start of a diagonal block: score register i of lane (r, h) is key c_i + 4h against query r of the block
      auto diag_start = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) CD[i] = ((i & 3) + 8 * (i >> 2) + 4 * h > r) ? -INFINITY : (FOLD ? nl[rb] : 0.f);
        settle_mfma(CD);   // VALU write -> asm MFMA operand: hipcc pads nothing in front of an asm statement
      };
      int kb[C::KS], tb[2][C::DB];
      int c = 0;
#ifndef FA_STAMPS_ITER
      FA4Q_STAMP(11);   // seg[11]: diagonal phase -- landing wait, barrier, next-stage descriptors
#endif
      for (; c < wave; ++c) {   // key blocks below both diagonals
        tr_bases(tb, cbase(c));
        row_bases(kb, cbase(c + 1));
        block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, No{}, tb, 0, kb, 0, no_hook); FA_DQ4_DIAG_SYNC();
        block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tb, 0, kb, 0, no_hook); FA_DQ4_DIAG_SYNC();
      }
#ifndef FA_STAMPS_ITER
      FA4Q_STAMP(12);   // seg[12]: 2 w visits below both diagonals
#endif
      // c = wave: row block 0's diagonal block, below row block 1's
      tr_bases(tb, cbase(c));
      row_bases(kb, cbase(c + 1));
      diag_start(0);
      block_iter(I0{}, I0{}, I1{}, Yes{}, No{}, Yes{}, tb, 0, kb, 0, hook_a); FA_DQ4_DIAG_SYNC();
      block_iter(I1{}, I1{}, I0{}, No{}, Yes{}, No{}, tb, 0, kb, 0, hook_b); FA_DQ4_DIAG_SYNC();
#ifndef FA_STAMPS_ITER
      FA4Q_STAMP(13);   // seg[13]: the two visits of key block w (row block 0's diagonal)
#endif
      // row block 1 alone: key blocks wave + 1 .. 6 - wave (an even number), then its diagonal block 7 - wave
      for (c = wave + 1; c < 7 - wave; c += 2) {
        tr_bases(tb, cbase(c));
        row_bases(kb, cbase(c + 1));
        block_iter(I0{}, I1{}, I1{}, Yes{}, Yes{}, No{}, tb, 0, kb, 0, no_hook); FA_DQ4_DIAG_SYNC();
        tr_bases(tb, cbase(c + 1));
        row_bases(kb, cbase(c + 2));
        block_iter(I1{}, I1{}, I1{}, Yes{}, Yes{}, No{}, tb, 0, kb, 0, no_hook); FA_DQ4_DIAG_SYNC();
      }
#ifndef FA_STAMPS_ITER
      FA4Q_STAMP(14);   // seg[14]: 6 - 2 w solo visits of row block 1
#endif
      tr_bases(tb, cbase(7 - wave));
      diag_start(1);
      block_iter(I0{}, I1{}, I1{}, Yes{}, No{}, Yes{}, tb, 0, kb, 0, hook_c); FA_DQ4_DIAG_SYNC();
      FA4Q_STAMP(2);
      pipe_drain(I0{}, I1{});
    }
    FA4Q_STAMP(3);

    __syncthreads();  // every wave is out of the rings: they become the staging area
    if constexpr (FOLD) {
      if (p.qs) {  // workspace for the dK/dV launch: the scaled rows exactly as this kernel (and the forward) multiplied them
        const __amdgpu_buffer_rsrc_t rqs =
            make_rsrc((char*)p.qs + b_ * p.lqs.sb + h_ * p.lqs.sh, view_bytes(Sq, p.lqs.rs, C::ROWB));
        const int ln = lane_id_now();
        auto put = [&](int rb, int ks, u32x4 v) __attribute__((always_inline)) {
          buf_store16(rqs, (qrow[rb] + (ln & 31)) * p.lqs.rs + (2 * ks + (ln >> 5)) * 16, v);
        };
        put(0, 0, pin_read<0>());
        put(0, 1, pin_read<1>());
        put(0, 2, pin_read<2>());
        put(0, 3, pin_read<3>());
        put(1, 0, pin_read<8>());
        put(1, 1, pin_read<9>());
        put(1, 2, pin_read<10>());
        put(1, 3, pin_read<11>());
      }
    }
    // staged in this wave's rows of the K half of slot b0 (dead since the barrier above; the wave's own next fetch goes there)
    FA_LDS char* stage = smem + b0 * C::TILE_BYTES + wave * C::RB_BYTES;
    {
      const f32x16 acc0[C::DB] = {acc_read16<kAccBase>(), acc_read16<kAccBase + 16>()};
      store_tile_rows<D, T>(acc0, p.scale, stage, rdq, qrow[0] * dq_rs, lane, dq_rs);
      const f32x16 acc1[C::DB] = {acc_read16<kAccBase + 32>(), acc_read16<kAccBase + 48>()};
      store_tile_rows<D, T>(acc1, p.scale, stage, rdq, qrow[1] * dq_rs, lane, dq_rs);
    }
    FA4Q_STAMP(4);
#ifdef FA_STAMPS
    ++npass_;
#endif
  }  // pass
  }  // item
#ifdef FA_STAMPS
  if (p.dbg && (threadIdx.x & 63) == 0) {
    unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 4 + wave) * 32;
    for (int i = 0; i < 17; ++i) d[i] = seg[i];
    d[17] = ntile_;
    unsigned long long clk1_, rt1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1_), "=s"(rt1_)::"memory");
    d[18] = clk1_ - clk0_;
    d[19] = rt1_ - rt0_;
    d[20] = npass_;
  }
#endif
}

template <typename T, bool CAUSAL>
static hipError_t launch4(const BwdParams& p, hipStream_t s) {
  using C = Dq4Cfg;
  int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  {   // persistent: one workgroup per CU walks the work list (a multiple of 8 keeps a workgroup on one XCD's items)
    static std::atomic<int> cus{0};   // CU count of the device first launched on (devices of one node are alike)
    int n = cus.load(std::memory_order_relaxed);
    if (n == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
      n -= n % 8;
      cus.store(n, std::memory_order_relaxed);
    }
    if (grid > n) grid = n;
  }
  auto kern = fa_bwd_dq4_kernel<T, CAUSAL>;
  static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
  if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dq_v4(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sq + Dq4Cfg::BM - 1) / Dq4Cfg::BM;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
  p.div_per_bh = make_fastdiv(p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles);
  p.div_h = make_fastdiv(p.H);
  if (dtype == 1) return causal ? launch4<BF16, true>(p, s) : launch4<BF16, false>(p, s);
  return causal ? launch4<FP16, true>(p, s) : launch4<FP16, false>(p, s);
}

}  // namespace fa
