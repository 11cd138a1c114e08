// FlashAttention backward dQ (+ delta), head dim 64: hand-ordered three-stage software pipeline for gfx950.
//
// Same maths, rounding points and accumulation order as fa_bwd_dq.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:165-258): the results are bit-identical to that kernel's.
// Same decomposition too (workgroup = 4 waves = 128 query rows, wave = 32 rows, 64-key K/V tiles by LDS-DMA).
// What changes is the ORDER of the work inside a wave.  fa_bwd_dq.hip runs, per 32-key block, the score MFMAs,
// then the exp / multiply / pack VALU work, then the dQ MFMAs: the matrix pipe idles during the VALU phase and the
// vector issue port idles during the MFMA phases, and co-resident waves only partly fill the holes (measured: 2 and 3
// workgroups per CU run at the same speed).  Here every wave overlaps the three phases of three consecutive blocks
// itself:
//
//     block iteration b :   MFMA   dQ^T += K^T dS^T      of block b-2      (4 slots)
//                           MFMA   S^T = K Q^T, dP^T = V dO^T  of block b  (8 slots)
//                           VALU   dS = exp2(S') * dP', pack               of block b-1  (2 exp + 2 mul + 1 cvt_pk per slot)
//                           LDS    the operand of the slot FOUR slots ahead (a 4-deep register ring)
//
// Every slot is closed by __builtin_amdgcn_sched_barrier(0): hipcc keeps exactly this order and only allocates
// registers, counts waits and pads hazards (the method of fa_bwd_dkv_v2.hip).  Two accumulator sets and two packed-dS
// sets alternate between the two key blocks of a tile, so the tile loop carries no register copies.
//
// LDS: K ring of 3 tiles (tile t-1 feeds the dQ MFMAs while tile t feeds the scores and tile t+1 lands), V ring of 3
// (2 would do; 3 keeps the slot a compile-time function of t % 3); the loop is unrolled three tiles deep so that every
// LDS address is a base register plus an immediate.  One raw s_barrier per tile; the operand reads that cross it are
// K^T fragments of the CURRENT tile, which no DMA overwrites before the next barrier.
// Masked tiles (causal diagonal, ragged tail) come last in the key loop and run on the plain per-block code path.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dq3Cfg {
  static constexpr int D = 64;
  static constexpr int BM = 128, BN = 64, NT = 256;
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BN * ROWB;
  static constexpr int RING = 3;
  static constexpr int V_BASE = RING * TILE_BYTES;
  static constexpr int LDS_BYTES = 2 * RING * TILE_BYTES;       // 48 KiB
  static constexpr int DMA_PER_MAT = TILE_BYTES / (4 * 1024);   // 1-KiB LDS-DMA pieces per wave per matrix
  static constexpr int NS = 2 * DB + 2 * KS;                    // MFMA slots per block iteration: dQ, S, dP
};

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void fa_bwd_dq3_kernel(BwdParams p) {
  using C = Dq3Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h
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


  // ---- resident B operands: Q^T and dO^T of this wave's 32 rows; delta (K:210-211, from the rounded O) ----
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
  // both score chains START from this lane's (= query row's) constant (see fa_bwd_dq.hip)
  f32x16 ndelta, nlse;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    ndelta[i] = -delta;
    nlse[i] = FOLD ? nl : 0.f;
  }
  const float c2 = p.scale * kLog2e;
  if constexpr (FOLD) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = scale_frag<T>(qf[ks], c2);
  }

  const int kv_end = CAUSAL ? min(Sk, q0_wg + C::BM) : Sk;
  const int ntiles = (kv_end + C::BN - 1) / C::BN;
  // tiles [0, npipe) need no mask for ANY wave of the workgroup: they run through the pipelined loop (the trip count
  // must be workgroup-uniform: one barrier per tile); the rest run on the masked path below
  const int npipe = CAUSAL ? min(Sk / C::BN, q0_wg / C::BN) : Sk / C::BN;

  // LDS-DMA source offsets (see fa_fwd.hip): wave w fills rows [16w, 16w+16) of each tile
  constexpr int RPI = 1024 / C::ROWB;
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + RPI * i + lane / C::CPR;
    dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16 - 1024 * i;  // dma_pieces: immediate offset taken out
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

  auto dma_tile = [&](int t, int slot) __attribute__((always_inline)) {
    const int soff = t * C::BN * kv_rs;
    const int dst0 = slot * C::TILE_BYTES + 16 * wave * C::ROWB;  // this wave's 16 rows = DMA_PER_MAT consecutive KiB
    dma_pieces<C::DMA_PER_MAT>(rk, lds_addr_of(smem + dst0), dma_src, soff);
    dma_pieces<C::DMA_PER_MAT>(rv, lds_addr_of(smem + C::V_BASE + dst0), dma_src, soff);
  };
  // the tile fetched during this step has landed (vmcnt(0)); every wave is done with the tiles the next DMA overwrites
  auto tile_sync = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) and lgkmcnt(0), see fa_bwd_dq.hip
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // pipelined loop: vmcnt(0) only -- the LDS reads in flight across this barrier are the next iteration's first dQ
  // operands, K^T fragments of the tile just scored, whose ring slot is not rewritten before the NEXT barrier
  auto pipe_sync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");       // the DMA rewrites LDS behind hipcc's back: no LDS load may move across
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- dS = exp2(S') * dP' of one block, in place in `xs` (FOLD: S' already is the exponent argument) ----
  auto ds_elem_exp = [&](f32x16& xs, int e) __attribute__((always_inline)) {
    xs[e] = __builtin_amdgcn_exp2f(FOLD ? xs[e] : __builtin_fmaf(xs[e], c2, nl));
  };

  // One 32-key block on the plain path (masked tiles, and the pipeline's drain).  `slot` = ring slot of its tile.
  auto block_plain = [&](int t, int slot, int b, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int s0 = t * C::BN;
    if constexpr (MASKED) {
      bool use = s0 + 32 * b < Sk;
      if (CAUSAL) use = use && (s0 + 32 * b <= qw0);
      if (!use) return;
    }
    const FA_LDS char* kbp = smem + slot * C::TILE_BYTES + b * 32 * C::ROWB;
    const FA_LDS char* vbp = smem + C::V_BASE + slot * C::TILE_BYTES + b * 32 * C::ROWB;
    f32x16 sacc = nlse, pacc = ndelta;
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
      sacc[i] = __builtin_amdgcn_exp2f(x) * pacc[i];  // dS^T = P^T o (dP^T - delta)
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
  };

  // ---- pipeline state (see the header): two accumulator sets, two packed-dS sets, the operand ring ----
  f32x16 sA, pA, sB, pB;   // set A: key block 0 of a tile, set B: key block 1
  u32x4 dkA[2], dkB[2];    // packed dS (k-steps 0, 1) of the block whose dQ MFMAs come next / after next
  vec8 fr[4];

  // K^T fragment n (d block n>>1, k-step n&1) of key block `b` of the tile in ring slot SLOT
  auto ktr_frag = [&](int slot, int b, int n) __attribute__((always_inline)) -> vec8 {
    const FA_LDS char* base = smem + slot * C::TILE_BYTES + b * 32 * C::ROWB + (n & 1) * 16 * C::ROWB;
    return lds_read_tr_frag<T>(base + tr_off[0][n >> 1], base + tr_off[1][n >> 1]);
  };

  // One block iteration.  PH = ring slot of the current tile t (= t % 3), KB = key block of t whose scores are
  // computed; HAS_X: the previous block exists (its VALU work runs here); HAS_Q: the block before that exists (its dQ
  // MFMAs run here, operands already in `fr`); NEXT_Q: the next iteration has dQ MFMAs (prefetch their operands).
  // sW / pW: accumulators written (block (t, KB)); sX / pX: the previous block's, turned into dS in place;
  // dk_in: packed dS consumed by the dQ MFMAs; dk_out: packed dS produced from sX / pX.
  auto blk = [&](auto ph_tag, auto kb_tag, auto x_tag, auto q_tag, auto nq_tag, f32x16& sW, f32x16& pW, f32x16& sX,
                 f32x16& pX, u32x4 (&dk_in)[2], u32x4 (&dk_out)[2]) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value, KB = decltype(kb_tag)::value;
    constexpr bool HAS_X = decltype(x_tag)::value, HAS_Q = decltype(q_tag)::value, NEXT_Q = decltype(nq_tag)::value;
    constexpr int S0 = 2 * C::DB, P0 = S0 + C::KS, NS = C::NS;
    constexpr int PREV = (PH + 2) % 3;  // ring slot of tile t-1
    const FA_LDS char* kb_rows = smem + PH * C::TILE_BYTES + KB * 32 * C::ROWB;
    const FA_LDS char* vb_rows = smem + C::V_BASE + PH * C::TILE_BYTES + KB * 32 * C::ROWB;
    // operand of slot s of THIS iteration (s >= S0) or of slot s - NS of the NEXT one (its dQ slots)
    auto frag = [&](int s) __attribute__((always_inline)) -> vec8 {
      if (s < P0) return as_vec8<T>(lds_read16(kb_rows + row_off[s - S0]));
      if (s < NS) return as_vec8<T>(lds_read16(vb_rows + row_off[s - P0]));
      // next iteration: KB = 0 -> (t, 1), dQ of block (t-1, 1);  KB = 1 -> (t+1, 0), dQ of block (t, 0)
      return KB == 0 ? ktr_frag(PREV, 1, s - NS) : ktr_frag(PH, 0, s - NS);
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // ---- the MFMA of this slot ----
      const vec8 a = fr[s & 3];
      if (s < S0) {
        if (HAS_Q) dqacc[s >> 1] = T::mfma(a, as_vec8<T>(dk_in[s & 1]), dqacc[s >> 1]);
      } else if (s < P0) {
        sW = T::mfma(a, qf[s - S0], s == S0 ? nlse : sW);
      } else {
        pW = T::mfma(a, dof[s - P0], s == P0 ? ndelta : pW);
      }
      // ---- LDS: operand four slots ahead ----
      if (s + 4 < NS || NEXT_Q) fr[s & 3] = frag(s + 4);
      // ---- VALU of the previous block: exp under slots 0..7, multiply one slot later, pack the slot after ----
      if (HAS_X) {
        if (s < 8) {
          ds_elem_exp(sX, 2 * s);
          ds_elem_exp(sX, 2 * s + 1);
        }
        if (s >= 1 && s < 9) {
          sX[2 * s - 2] *= pX[2 * s - 2];
          sX[2 * s - 1] *= pX[2 * s - 1];
        }
        if (s >= 2 && s < 10) {
          const int j = s - 2;  // pair j = registers 2j, 2j+1
          dk_out[j >> 2][j & 3] = pack2<T>(sX[2 * j], sX[2 * j + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using Yes = std::true_type;
  using No = std::false_type;

  // one pipelined tile: DMA of tile t+1, the two block iterations of tile t, then the barrier
  auto tile_pipe = [&](auto ph_tag, auto first_tag, int t) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    if (t + 1 < ntiles) dma_tile(t + 1, (PH + 1) % 3);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST) {  // tile 0: nothing behind it yet; the (empty) dQ slots of the first block iteration just
                            // prefetch the S operands, the second one starts the dS pipeline and prefetches for tile 1
      blk(ph_tag, I0{}, No{}, No{}, No{}, sA, pA, sB, pB, dkA, dkB);
      blk(ph_tag, I1{}, Yes{}, No{}, Yes{}, sB, pB, sA, pA, dkB, dkA);
    } else {
      blk(ph_tag, I0{}, Yes{}, Yes{}, Yes{}, sA, pA, sB, pB, dkA, dkB);
      blk(ph_tag, I1{}, Yes{}, Yes{}, Yes{}, sB, pB, sA, pA, dkB, dkA);
    }
    pipe_sync();
  };

  if (Sk % C::BN != 0) {  // a ragged last tile must not expose uninitialised LDS
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }
  dma_tile(0, 0);
  tile_sync();

  int t = 0;
  if (npipe > 0) {
    tile_pipe(I0{}, Yes{}, 0);
    t = 1;
    // Canonical single-exit loop, three tiles per trip (ring slots 1, 2, 0).  With early exits after each tile hipcc
    // reconciles the register assignment of every exit with the drain code below by copying accumulators around
    // inside the loop and spills ~150 registers; up to two left-over full tiles run on the plain path instead.
    while (t + 3 <= npipe) {
      tile_pipe(I1{}, No{}, t);
      tile_pipe(I2{}, No{}, t + 1);
      tile_pipe(I0{}, No{}, t + 2);
      t += 3;
    }
    // ---- drain: block (t-1, 0) has its packed dS in dkA and its K^T fragments in `fr`; block (t-1, 1) is raw in sB / pB
    // (t - 1 is a multiple of 3 here: the last pipelined tile sits in ring slot 0)
    constexpr int last_slot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ds_elem_exp(sB, i);
      sB[i] *= pB[i];
    }
#pragma unroll
    for (int n = 0; n < 2 * C::DB; ++n) dqacc[n >> 1] = T::mfma(fr[n & 3], as_vec8<T>(dkA[n & 1]), dqacc[n >> 1]);
    const vec8 d0 = pack8<T, 0>(sB), d1 = pack8<T, 1>(sB);
#pragma unroll
    for (int n = 0; n < 2 * C::DB; ++n) dqacc[n >> 1] = T::mfma(ktr_frag(last_slot, 1, n), (n & 1) ? d1 : d0, dqacc[n >> 1]);
    tile_sync();  // every wave is out of the rings before the masked path or the epilogue reuses them
  }
  // ---- left-over full tiles, then the masked ones (causal diagonal, ragged tail): plain path, ring slot t % 3 ----
  for (; t < npipe; ++t) {
    const int slot = t % 3;
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) % 3);
    block_plain(t, slot, 0, std::false_type{});
    block_plain(t, slot, 1, std::false_type{});
    tile_sync();
  }
  for (; t < ntiles; ++t) {
    const int slot = t % 3;
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) % 3);
    block_plain(t, slot, 0, std::true_type{});
    block_plain(t, slot, 1, std::true_type{});
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

template <typename T, bool CAUSAL>
static hipError_t launch3(const BwdParams& p, hipStream_t s) {
  using C = Dq3Cfg;
  const int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  hipLaunchKernelGGL((fa_bwd_dq3_kernel<T, CAUSAL>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dq_v3(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sq + Dq3Cfg::BM - 1) / Dq3Cfg::BM;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
  if (dtype == 1) return causal ? launch3<BF16, true>(p, s) : launch3<BF16, false>(p, s);
  return causal ? launch3<FP16, true>(p, s) : launch3<FP16, false>(p, s);
}

}  // namespace fa
