// FlashAttention forward, head dim 64: hand-ordered three-stage software pipeline for gfx950 (schedule family 3).
//
// Same maths and rounding points as fa_fwd.hip (reference kernel code/_flash_attention_kernel_optimized.py:35-129):
// fp32 scores and softmax state, l sums the un-rounded p (K:111), P rounded to the input dtype for P @ V (K:115),
// O = o / l cast on store (K:120-123), LSE = m + ln(l) (K:126), top-left causal mask (K:102), keys >= S_k masked (K:94).
// Same decomposition (workgroup = 4 waves = 128 query rows, wave = 32 rows, 64-key K/V tiles by LDS-DMA, scores
// transposed so the query is the MFMA lane) and the same lazy running max (fa_fwd.hip: exponentiate against the stale
// row max, prove with the row sums that nothing overflowed, otherwise redo exactly before anything was committed).
//
// What changes is the ORDER of the work inside a wave (the method of fa_bwd_dq_v3.hip).  fa_fwd.hip runs QK^T, then the
// softmax VALU work, then PV: the matrix pipe idles during the softmax and the vector port during the MFMA chains.
// Here every wave overlaps the three phases of three consecutive 32-key blocks itself:
//
//     block iteration b :   MFMA   O^T += V^T P^T       of block b-2      (4 slots)
//                           MFMA   S^T = K Q^T (- m)    of block b        (4 slots)
//                           VALU   p = exp2(S'), row sum, pack            of block b-1  (2 exp + 2 add + 1 cvt_pk per slot)
//                           LDS    the operand of the slot FOUR slots ahead (a 4-deep register ring)
//
// every slot closed by __builtin_amdgcn_sched_barrier(0).  Two score sets and two packed-P sets alternate between the two
// key blocks of a tile (no register copies in the loop); K / V rings of 3 tiles, the loop unrolled three tiles deep so
// that ring slots are immediates; one raw s_barrier per tile.
//
// Lazy-max bookkeeping in a pipeline: block b-1 is exponentiated while block b is already being scored against the same
// stale max, and block b-1 is committed (PV) one iteration later.  The overflow test of block b-1 is known at the end
// of iteration b, i.e. BEFORE its PV.  On a failure the wave zeroes the packed P of the failed block (its PV then adds
// nothing), and at the end of the tile iteration -- before the barrier, so every tile it needs is still in the rings and
// no other wave can move on -- redoes the uncommitted blocks exactly (block_plain), then neutralises the pipeline
// registers (scores = -inf, packed P = 0) so that the next iterations have nothing left to do for those blocks.  No
// control flow leaves the loop, nothing is shared between waves.  The first tile (m = -inf) fails by construction and
// takes exactly this path.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Fwd3Cfg {
  static constexpr int D = 64;
  static constexpr int BM = 128, BN = 64, NT = 256;
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BN * ROWB;
  static constexpr int RING = 3;
  static constexpr int V_BASE = RING * TILE_BYTES;
  static constexpr int LDS_BYTES = 2 * RING * TILE_BYTES;       // 48 KiB
  static constexpr int DMA_PER_MAT = TILE_BYTES / (4 * 1024);   // 1-KiB LDS-DMA pieces per wave per matrix
  static constexpr int NS = 2 * DB + KS;                        // MFMA slots per block iteration: PV, S
};

constexpr float kLazySumMax3 = 8192.0f;  // as fa_fwd.hip kLazySumMax, per 32-key block

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void fa_fwd3_kernel(FwdParams p) {
  using C = Fwd3Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // work list as fa_fwd.hip: causal workgroups take the query-tile pair (nq-1-i, i)
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.nq_tiles + 1) / 2 : p.nq_tiles;
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const BatchHead ix = batch_head(bh, p.B, p.H, p.vl.cu_q != nullptr);
  const int b_ = ix.b, h_ = ix.h;
  const SeqInfo si = seq_info(p.vl, b_, p.Sq, p.Sk);
  const int Sq = si.Sq, Sk = si.Sk;
  const int nq = (Sq + C::BM - 1) / C::BM;
  if (idx >= (paired ? (nq + 1) / 2 : nq)) return;
  const int npass = (paired && idx != nq - 1 - idx) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int lane = lane_id_now(), tid = wave * 64 + lane, r = lane & 31, h = lane >> 5;
  const int qt = paired ? (pass == 0 ? nq - 1 - idx : idx) : (CAUSAL ? nq - 1 - idx : idx);  // heavy first
  const int q0_wg = qt * C::BM;
  const int qw0 = q0_wg + wave * 32;
  if (pass) __syncthreads();  // the previous pass staged its O tile in the rings

  const int q_rs = p.lq.rs, kv_rs = p.lk.rs, o_rs = p.lo.rs;
  const char* qb = (const char*)p.q + b_ * p.lq.sb + h_ * p.lq.sh + (long long)si.q0 * q_rs;
  const char* kb = (const char*)p.k + b_ * p.lk.sb + h_ * p.lk.sh + (long long)si.k0 * kv_rs;
  const char* vb = (const char*)p.v + b_ * p.lv.sb + h_ * p.lv.sh + (long long)si.k0 * kv_rs;
  char* ob = (char*)p.o + b_ * p.lo.sb + h_ * p.lo.sh + (long long)si.q0 * o_rs;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(qb, (unsigned)(Sq - 1) * q_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(kb, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(vb, view_bytes(Sk, kv_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(ob, (unsigned)(Sq - 1) * o_rs + C::ROWB);
  const __amdgpu_buffer_rsrc_t rl = make_rsrc(p.lse + b_ * p.lse_sb + h_ * p.lse_sh + si.q0, (unsigned)Sq * 4);


  // ---- Q^T fragments (B operand), resident ----
  vec8 qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) qf[ks] = as_vec8<T>(buf_load16(rq, (qw0 + r) * q_rs + (2 * ks + h) * 16));

  const int kv_end = CAUSAL ? min(Sk, q0_wg + C::BM) : Sk;
  const int ntiles = (kv_end + C::BN - 1) / C::BN;
  // tiles [0, npipe) need no mask for ANY wave of the workgroup: pipelined (the trip count must be workgroup-uniform)
  const int npipe = CAUSAL ? min(Sk / C::BN, q0_wg / C::BN) : Sk / C::BN;

  constexpr int RPI = 1024 / C::ROWB;
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + RPI * i + lane / C::CPR;
    dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16 - 1024 * i;  // dma_pieces: immediate offset taken out
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
  const float cs = FOLD ? 1.0f : c2;  // accumulator units -> log2 units
  if constexpr (FOLD) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = scale_frag<T>(qf[ks], c2);
  }
  float m = -INFINITY;  // running row max in accumulator units (raw scores, or log2 units if FOLD)
  f32x16 negm;          // FOLD: -m in every register: the lazy score chain starts from it
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = INFINITY;
  float l = 0.f;        // this lane's partial row sum
  f32x16 oacc[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;

  auto dma_tile = [&](int t, int slot) __attribute__((always_inline)) {
    const int soff = t * C::BN * kv_rs;
    const int dst0 = slot * C::TILE_BYTES + 16 * wave * C::ROWB;
    dma_pieces<C::DMA_PER_MAT>(rk, lds_addr_of(smem + dst0), dma_src, soff);
    dma_pieces<C::DMA_PER_MAT>(rv, lds_addr_of(smem + C::V_BASE + dst0), dma_src, soff);
  };
  auto tile_sync = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) and lgkmcnt(0), see fa_fwd.hip
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto pipe_sync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");       // the DMA rewrites LDS behind hipcc's back: no LDS load may move across
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); the LDS reads in flight belong to a tile no DMA rewrites yet
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // V^T fragment n (d block n >> 1, k-step n & 1) of key block `b` of the tile in ring slot `slot`
  auto vtr_frag = [&](int slot, int b, int n) __attribute__((always_inline)) -> vec8 {
    const FA_LDS char* base = smem + C::V_BASE + slot * C::TILE_BYTES + b * 32 * C::ROWB + (n & 1) * 16 * C::ROWB;
    return lds_read_tr_frag<T>(base + v_off[0][n >> 1], base + v_off[1][n >> 1]);
  };

  // One 32-key block on the exact path: true row max, rescale, exp, P @ V.  Used for the masked tiles, the left-over
  // tiles, the pipeline's drain and its recovery.  `slot` = ring slot of tile t (run-time).
  auto block_plain = [&](int t, int slot, int b, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int s0 = t * C::BN + 32 * b;
    if constexpr (MASKED) {
      bool use = s0 < Sk;
      if (CAUSAL) use = use && (s0 <= qw0);
      if (!use) return;
    }
    const FA_LDS char* kbp = smem + slot * C::TILE_BYTES + b * 32 * C::ROWB;
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      vec8 a = as_vec8<T>(lds_read16(kbp + k_off[ks]));
      sacc = T::mfma(a, qf[ks], sacc);
    }
    if constexpr (MASKED) {
      const int qrow = qw0 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = s0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool dead = (CAUSAL && key > qrow) || key >= Sk;
        sacc[i] = dead ? -INFINITY : sacc[i];
      }
    }
    float tm = sacc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tm = __builtin_fmaxf(tm, sacc[i]);
    tm = half_max(tm);
    const float mn = __builtin_fmaxf(m, tm);
    // a row that sees no key of this block yet and has none before keeps mn = -inf: exp2(nan) must not enter l / O
    const float corr = (mn == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m - mn) * cs);  // m = -inf -> 0
    l *= corr;
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[db][i] *= corr;
    m = mn;
    if constexpr (FOLD) {
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = (mn == -INFINITY) ? INFINITY : -mn;
    }
    const float mc = m * cs;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float x = FOLD ? sacc[i] - mc : __builtin_fmaf(sacc[i], c2, -mc);
      const float pe = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(x);
      sacc[i] = pe;
      ls[i & 3] += pe;
    }
    l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
    const vec8 pf0 = pack8<T, 0>(sacc), pf1 = pack8<T, 1>(sacc);
#pragma unroll
    for (int n = 0; n < 2 * C::DB; ++n) oacc[n >> 1] = T::mfma(vtr_frag(slot, b, n), (n & 1) ? pf1 : pf0, oacc[n >> 1]);
  };

  // ---- pipeline state: two score sets, two packed-P sets, the operand ring ----
  f32x16 sA, sB;         // set A: key block 0 of a tile, set B: key block 1
  u32x4 pkA[2], pkB[2];  // packed P (k-steps 0, 1) of the block whose PV MFMAs come next / after next
  vec8 fr[4];

  // One block iteration.  PH = ring slot of the current tile t, KB = key block of t whose scores are computed.
  // HAS_X: the previous block exists (its softmax runs here, in place in sX, packed into pk_out, row sum in `lsum`);
  // HAS_PV: the block before that exists (its PV MFMAs run here: operands in `fr`, P in pk_in);
  // NEXT_PV: the next iteration has PV MFMAs (prefetch their V^T operands).
  auto blk = [&](auto ph_tag, auto kb_tag, auto x_tag, auto pv_tag, auto npv_tag, f32x16& sW, f32x16& sX, u32x4 (&pk_in)[2],
                 u32x4 (&pk_out)[2], float& lsum) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value, KB = decltype(kb_tag)::value;
    constexpr bool HAS_X = decltype(x_tag)::value, HAS_PV = decltype(pv_tag)::value, NEXT_PV = decltype(npv_tag)::value;
    constexpr int S0 = 2 * C::DB, NS = C::NS;
    constexpr int PREV = (PH + 2) % 3;  // ring slot of tile t-1
    const FA_LDS char* kb_rows = smem + PH * C::TILE_BYTES + KB * 32 * C::ROWB;
    auto frag = [&](int s) __attribute__((always_inline)) -> vec8 {
      if (s < NS) return as_vec8<T>(lds_read16(kb_rows + k_off[s - S0]));
      // next iteration: KB = 0 -> (t, 1), PV of block (t-1, 1);  KB = 1 -> (t+1, 0), PV of block (t, 0)
      return KB == 0 ? vtr_frag(PREV, 1, s - NS) : vtr_frag(PH, 0, s - NS);
    };
    const float mc = m * c2;  // !FOLD: exponent argument = s * c2 - m * c2
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const vec8 a = fr[s & 3];
      if (s < S0) {
        if (HAS_PV) oacc[s >> 1] = T::mfma(a, as_vec8<T>(pk_in[s & 1]), oacc[s >> 1]);
      } else if (s == S0) {
        if constexpr (FOLD) {
          sW = T::mfma(a, qf[0], negm);
        } else {
          f32x16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          sW = T::mfma(a, qf[0], z);
        }
      } else {
        sW = T::mfma(a, qf[s - S0], sW);
      }
      if (s + 4 < NS || NEXT_PV) fr[s & 3] = frag(s + 4);
      if (HAS_X) {
        // exp of elements 2s, 2s+1; row-sum adds and the pack of the pair exponentiated one slot earlier
#pragma unroll
        for (int e = 2 * s; e < 2 * s + 2; ++e)
          sX[e] = __builtin_amdgcn_exp2f(FOLD ? sX[e] : __builtin_fmaf(sX[e], c2, -mc));
        if (s >= 1) {
          ls[(2 * s - 2) & 3] += sX[2 * s - 2];
          ls[(2 * s - 1) & 3] += sX[2 * s - 1];
          const int j = s - 1;
          pk_out[j >> 2][j & 3] = pack2<T>(sX[2 * j], sX[2 * j + 1]);
        }
        if (s == NS - 1) {  // the last pair
          ls[(2 * s) & 3] += sX[2 * s];
          ls[(2 * s + 1) & 3] += sX[2 * s + 1];
          pk_out[s >> 2][s & 3] = pack2<T>(sX[2 * s], sX[2 * s + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using Yes = std::true_type;
  using No = std::false_type;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  bool pending = false;  // (wave-uniform) the last pipelined tile's second block is raw in sB

  // one pipelined tile: DMA of tile t+1, the two block iterations of tile t, the lazy-max checks, recovery, barrier
  auto tile_pipe = [&](auto ph_tag, auto first_tag, int t) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int PREV = (PH + 2) % 3;
    if (t + 1 < ntiles) dma_tile(t + 1, (PH + 1) % 3);
    __builtin_amdgcn_sched_barrier(0);
    float lsum0 = 0.f, lsum1 = 0.f;
    bool ok0 = true;
    if constexpr (FIRST) {
      blk(ph_tag, I0{}, No{}, No{}, No{}, sA, sB, pkA, pkB, lsum0);
    } else {
      // block (t-1, 1) is exponentiated (sB -> pkB), block (t-1, 0) committed (pkA).  Branch-free commit: on an overflow
      // the packed P is zeroed (its PV in the next block iteration then adds nothing) and the row sum is not taken.
      blk(ph_tag, I0{}, Yes{}, Yes{}, Yes{}, sA, sB, pkA, pkB, lsum0);
      ok0 = __builtin_amdgcn_ballot_w64(!(lsum0 <= kLazySumMax3)) == 0;
      const unsigned keep0 = ok0 ? 0xFFFFFFFFu : 0u;
      l += ok0 ? lsum0 : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pkB[i][j] &= keep0;
    }
    // block (t, 0) is exponentiated (sA -> pkA), block (t-1, 1) committed (pkB)
    if constexpr (FIRST) blk(ph_tag, I1{}, Yes{}, No{}, Yes{}, sB, sA, pkB, pkA, lsum1);
    else blk(ph_tag, I1{}, Yes{}, Yes{}, Yes{}, sB, sA, pkB, pkA, lsum1);
    const bool ok1 = ok0 && __builtin_amdgcn_ballot_w64(!(lsum1 <= kLazySumMax3)) == 0;
    l += ok1 ? lsum1 : 0.f;
    if (!ok1) {
      // Recovery (rare; always on the first tile).  Everything up to block (t-1, 0) -- and (t-1, 1) if ok0 -- is
      // committed.  Redo the rest exactly from the rings (all still resident: the barrier has not been passed),
      // then leave nothing for the pipeline to do on these blocks.
#pragma nounroll
      for (int i = ok0 ? 1 : 0; i < 3; ++i)   // (t-1, 1) if it failed, then (t, 0), (t, 1): one copy of the block code
        block_plain(i == 0 ? t - 1 : t, i == 0 ? PREV : PH, i == 0 ? 1 : i - 1, No{});
      pkA[0] = zero4;
      pkA[1] = zero4;
#pragma unroll
      for (int i = 0; i < 16; ++i) sB[i] = -INFINITY;  // block (t, 1): exp2(-inf) = 0 in the next iteration
    }
    pending = ok1;  // block (t, 1) still waits in sB unless the recovery has just processed it
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
    // canonical single-exit loop, three tiles per trip (ring slots 1, 2, 0); see fa_bwd_dq_v3.hip
    while (t + 3 <= npipe) {
      tile_pipe(I1{}, No{}, t);
      tile_pipe(I2{}, No{}, t + 1);
      tile_pipe(I0{}, No{}, t + 2);
      t += 3;
    }
    // ---- drain: block (t-1, 0) has its packed P in pkA and its V^T fragments in `fr` (ring slot 0: t - 1 is a multiple
    // of 3); block (t-1, 1) is raw in sB -- it is simply redone on the exact path (four extra MFMAs per pass)
#pragma unroll
    for (int n = 0; n < 2 * C::DB; ++n) oacc[n >> 1] = T::mfma(fr[n & 3], as_vec8<T>(pkA[n & 1]), oacc[n >> 1]);
    if (pending) block_plain(t - 1, 0, 1, No{});
    tile_sync();  // every wave is out of the rings before the plain path or the epilogue reuses them
  }
  // ---- left-over full tiles, then the masked ones (causal diagonal, ragged tail): exact path, ring slot t % 3 ----
  for (; t < npipe; ++t) {
    const int slot = t % 3;
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) % 3);
#pragma nounroll
    for (int b = 0; b < 2; ++b) block_plain(t, slot, b, No{});
    tile_sync();
  }
  for (; t < ntiles; ++t) {
    const int slot = t % 3;
    if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) % 3);
#pragma nounroll
    for (int b = 0; b < 2; ++b) block_plain(t, slot, b, Yes{});
    tile_sync();
  }

  // ---- epilogue ----
  const float lt = half_sum(l);
  // lt = 0 only for a variable-length sequence with queries but no keys (S_k = 0: no tile was visited): O = 0, LSE = -inf
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  store_tile_rows<D, T>(oacc, inv, smem + wave * 32 * C::ROWB, ro, qw0 * o_rs, lane, o_rs);
  if (h == 0) buf_store_f32(rl, (qw0 + r) * 4, m * (FOLD ? kLn2 : p.scale) + __builtin_logf(lt));
  }  // pass
}

template <typename T, bool CAUSAL>
static hipError_t launch3(const FwdParams& p, hipStream_t s) {
  using C = Fwd3Cfg;
  const int grid = (CAUSAL && p.pair ? (p.nq_tiles + 1) / 2 : p.nq_tiles) * p.B * p.H;
  hipLaunchKernelGGL((fa_fwd3_kernel<T, CAUSAL>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_fwd_v3(FwdParams p, int dtype, int causal, hipStream_t s) {
  p.nq_tiles = (p.Sq + Fwd3Cfg::BM - 1) / Fwd3Cfg::BM;
  p.pair = want_pairs(causal != 0, p.nq_tiles, (long)p.B * p.H);
  if (dtype == 1) return causal ? launch3<BF16, true>(p, s) : launch3<BF16, false>(p, s);
  return causal ? launch3<FP16, true>(p, s) : launch3<FP16, false>(p, s);
}

}  // namespace fa
