// FlashAttention backward dK / dV, fourth schedule family for gfx950 (head dim 64): family 3's one-wave-per-SIMD pipeline on
// the pinned accumulator file, with a barrier-free per-wave diagonal phase.
//
// Same maths, rounding points and accumulation order as fa_bwd_dkv_v3.hip / _v2.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:292-386; runs after the dQ kernel and reads its delta): bit-identical results.
// Same decomposition (workgroup = 256 keys, a wave owns key groups {w, 7 - w} of 32 keys, 128-row Q / dO tiles by LDS-DMA,
// every Q / dO fragment read feeding both key groups) and the same block iteration:
//     slots 0-3   S  = Q K^T        of block i     (VGPR-form asm MFMAs, chain starts from -LSE*log2e: C operand != D)
//     slots 4-7   dP = dO V^T       of block i     (chain starts from -delta)
//     slots 8-11  dV^T += dO^T P    of block i-1   (asm MFMAs into pinned accumulator registers)
//     slots 12-15 dK^T += Q^T dS    of block i-1
// with the exp2 / multiply / pack work of both blocks under them, one of each kind per slot (family 3's timetable).
// What family 4 changes:
//   * Everything that need not be architectural lives in LITERALLY named accumulator registers (fa_common.h): dK^T / dV^T
//     (a[0:127]) and the resident K^T / V^T fragments (a[128:191]).  hipcc allocates no accumulator register, copies none,
//     and the row-constant loads land in a[192] without touching the architectural file.
//   * A ring of THREE tile slots, tiles two ahead, a counted vmcnt(8) at the per-tile commit: a wave never waits for a fresh
//     LDS-DMA (family 3: two slots, vmcnt(0)).
//   * Causal: the two query tiles level with the key tile come LAST in a pass (the order of the query tiles is free: dK / dV
//     are sums over them), fetched by the ring like any other tile, and are both resident when their turn comes.  That phase
//     then runs per wave without barriers: both key groups for query blocks 7 .. 7 - w, key group w alone ("solo"
//     iterations) for 6 - w .. w -- the 9 visible block visits of 16; family 3 computes all 16 plus a whole extra tile on its
//     masked path, with a compare + select per element.  Here the mask of the two diagonal blocks is free: their score chains
//     start from (dead ? -inf : -LSE*log2e).  (Non-causal results are bit-identical to family 3's; causal ones differ by
//     the fp32 summation order only.)
//   * Causal launches are PERSISTENT and a pass stages the NEXT pass's resident K / V rows through LDS from inside its
//     diagonal phase, its first two tiles and first row constant from in front of its epilogue (fa_bwd_dq_v4.hip says why).
// ORDER OF REQUESTS (the invariant every counted wait below relies on; round 4 broke it once, tools/race_stress.py found it):
// a commit requests the NEXT row constant first, then the Q pairs of the tile two ahead, then (next iteration) its dO pairs --
// so that the following commit's vmcnt(8) leaves exactly those eight pieces in flight and the row constant, older, has
// landed.  The pass prologue issues its requests in the same order.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct Dkv4Cfg {
  static constexpr int D = 64;
  static constexpr int BK = 256, BQ = 128, NT = 256, NW = 4;
  static constexpr int QB = BQ / 32;                        // 32-row query blocks per tile
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int QBLK = 32 * ROWB;                    // bytes of one query block in a tile image
  static constexpr int TILE_BYTES = BQ * ROWB;              // 16 KiB
  static constexpr int NBUF = 3;
  static constexpr int DO_BASE = NBUF * TILE_BYTES;         // Q[NBUF], then dO[NBUF]
  static constexpr int ROWC_OFF = 2 * NBUF * TILE_BYTES;    // then row constants: nl[BQ], nd[BQ] per slot
  static constexpr int ROWC_BYTES = 2 * BQ * 4;
  // behind them: the NEXT pass's K rows, 64 per wave (key groups w and 7 - w, 32 rows x 128 B each), staged by LDS-DMA; its
  // V rows are staged in the ring slot that is free when they arrive (the kernel says where)
  static constexpr int KS_OFF = ROWC_OFF + NBUF * ROWC_BYTES;
  static constexpr int KG_BYTES = 32 * ROWB;                // one key group's rows: 4 KiB = 4 pieces
  // and the epilogue's own staging area (4 KiB per wave): the ring then belongs to the next pass while dK / dV are written out
  static constexpr int EPI_OFF = KS_OFF + BK * ROWB;
  static constexpr int LDS_BYTES = EPI_OFF + NW * KG_BYTES;   // 147 KiB
  static constexpr int PIECES = TILE_BYTES / (NW * 1024);   // 1-KiB LDS-DMA pieces per wave per matrix (4)
  static constexpr int RPI = 1024 / ROWB;                   // tile rows per piece
  static constexpr int NS = 16;                             // MFMA slots per block iteration
  static constexpr int kOOB = 0x7FFFFFFF;                   // scalar offset of a fetch past the last tile: out of range, no traffic
  // the pinned accumulator file: dV^T block (key group kg, d block db) = a[16 (2 kg + db)], dK^T = a[64 + 16 (2 kg + db)];
  // resident fragment F = 16 + 4 kg + ks: K^T k-step ks of key group kg (a[128 + ..]), F = 24 + 4 kg + ks: V^T; a[192]: row constant
  static constexpr int ACC_DV = 0, ACC_DK = 64, F_K = 16, F_V = 24, A_RC = 192;
};

// -DFA_STAMPS (diagnostic build, tools/stamps_dq4.py --dkv): per-phase cycle account of a wave, written to BwdParams::dbg.
// seg[0] pass prologue (ring primed, K / V fragments, first barrier, fill)   seg[1] unmasked tiles   seg[2] diagonal phase
// seg[3] drain   seg[4] epilogue
#ifdef FA_STAMPS
#define FA4K_STAMP(slot)                                                          \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long now_;                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += now_ - last_;                                                    \
    last_ = now_;                                                                 \
  } while (0)
#else
#define FA4K_STAMP(slot) do {} while (0)
#endif

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void fa_bwd_dkv4_kernel(BwdParams p) {
#ifdef FA_STAMPS
  unsigned long long clk0_, rt0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_)::"memory");
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = clk0_, ntile_ = 0, npass_ = 0;
#endif
  using C = Dkv4Cfg;
  using vec8 = typename T::vec8;
  constexpr int D = C::D;
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h: the score chain starts from -LSE*log2e and K (or Q') carries c2
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Work list: one item = one 256-key tile, or (causal, paired) the tile pair (i, nk-1-i) -> equal work per item; the
  // XCD-aware order gives every XCD a contiguous run of (batch, head) slices.  Causal launches are PERSISTENT (one workgroup
  // per CU walks items blockIdx.x, blockIdx.x + gridDim.x, ...): a pass's resident K / V rows are staged through LDS from
  // inside the PREVIOUS pass (below), and an item's first pass has no previous pass unless the workgroup stays.
  const bool paired = CAUSAL && p.pair;
  const int nk = p.n_tiles;
  const int per_bh = paired ? (nk + 1) / 2 : nk;
  const int n_items = per_bh * p.B * p.H;
  const int Sq = p.Sq, Sk = p.Sk;
  const int q_rs = p.lq.rs, do_rs = p.ldo.rs, kv_rs = p.lk.rs;
  const float c2 = p.scale * kLog2e;
  const int lds0 = (int)lds_addr_of(smem);
  const int ntiles = Sq / C::BQ;   // the launcher guarantees whole tiles
  // row constants of a query tile: threads 0-127 load its LSE rows, 128-255 its delta rows, through ONE wave-uniform
  // descriptor (waves 0-1 / 2-3)
  const bool rc_lse = wave < 2;
  pin_reserve();

  struct Work {
    int b, h, idx, npass;
  };
  auto decode = [&](int item) __attribute__((always_inline)) -> Work {
    const int w = xcd_remap(item, n_items);
    const int bh = p.div_per_bh.div(w), idx = w - bh * per_bh, b = p.div_h.div(bh);   // (fa_kernels.h FastDiv)
    return Work{b, bh - b * p.H, idx, (paired && idx != nk - 1 - idx) ? 2 : 1};
  };
  auto ktile_of = [&](const Work& wk, int pass) __attribute__((always_inline)) -> int {   // low key tiles are the heavy ones
    return paired ? (pass == 0 ? wk.idx : nk - 1 - wk.idx) : wk.idx;
  };
  // The resident operands of a pass -- the K and V rows of this wave's two key groups -- are STAGED through LDS a whole phase
  // before the pass needs them (fa_bwd_dq_v4.hip says why: fetched as per-lane fragments they cost a wave thousands of cycles
  // of address-unit time at the top of a pass, with nothing else on the CU to cover them).  Four groups of four LDS-DMA pieces
  // per wave: g = 0, 1 K rows of key group 0, 1 -> KS; 2 V rows of key group 0 -> this wave's rows of the Q half of ring slot
  // `vslot`, 3 V rows of key group 1 -> its rows of the dO half: exactly the 2 x 4 KiB this wave fills itself with the pass's
  // third tile, once it has consumed them -- no other wave ever touches them.
  struct Stage {
    __amdgpu_buffer_rsrc_t rk, rv;
    int key0[2];
  };
  auto stage_of = [&](const Work& wk, int pass, bool valid) __attribute__((always_inline)) -> Stage {
    const int k0 = ktile_of(wk, pass) * C::BK;
    const unsigned nb = valid ? view_bytes(Sk, kv_rs, C::ROWB) : 0u;   // nothing follows: empty descriptors fetch nothing
    return Stage{make_rsrc((const char*)p.k + wk.b * p.lk.sb + wk.h * p.lk.sh, nb),
                 make_rsrc((const char*)p.v + wk.b * p.lv.sb + wk.h * p.lv.sh, nb), {k0 + 32 * wave, k0 + 32 * (7 - wave)}};
  };
  auto stage_group = [&](const Stage& st, int vslot, int g) __attribute__((always_inline)) {
    const int ln = lane_id_now(), prow = ln >> 3;   // piece i holds rows 8 i + prow of the key group
    const int kg = g & 1;
    int voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)   // (dma_pieces: the immediate 1024 i moves LDS and global address together: taken out here)
      voff[i] = (st.key0[kg] + 8 * i + prow) * kv_rs + swz_chunk<D>(8 * i + prow, ln & 7) * 16 - 1024 * i;
    const int dst = g < 2 ? C::KS_OFF + (2 * wave + kg) * C::KG_BYTES : (kg ? C::DO_BASE : 0) + vslot * C::TILE_BYTES + wave * C::KG_BYTES;
    dma_pieces<4>(g < 2 ? st.rk : st.rv, (unsigned)(lds0 + dst), voff, 0);
  };

  int item = blockIdx.x;
  Work wk = decode(item);
  Work nwk_item = decode(min(item + (int)gridDim.x, n_items - 1));   // the workgroup's next item, decoded once per item
  int b0 = 0, b1 = 1, b2 = 2;   // ring slots of stream positions i, i + 1, i + 2; they keep rotating from pass to pass
  bool staged = false;          // this pass's rows are on their way (issued from inside the previous pass)
  for (; item < n_items; item += gridDim.x, wk = nwk_item, nwk_item = decode(min(item + (int)gridDim.x, n_items - 1))) {
  const int b_ = wk.b, h_ = wk.h, npass = wk.npass;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc((const char*)p.q + b_ * p.lq.sb + h_ * p.lq.sh, view_bytes(Sq, q_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc((const char*)p.dout + b_ * p.ldo.sb + h_ * p.ldo.sh, view_bytes(Sq, do_rs, C::ROWB));
  const __amdgpu_buffer_rsrc_t rrc = make_rsrc((rc_lse ? p.lse : p.delta) + b_ * p.lse_sb + h_ * p.lse_sh, (unsigned)Sq * 4);

  for (int pass = 0; pass < npass; ++pass) {
    const int lane = lane_id_now(), r = lane & 31, h = lane >> 5;
    const int k0_wg = ktile_of(wk, pass) * C::BK;
    // this wave's two 32-key groups: {w, 7-w} of the workgroup's eight (equal causal work per wave, see the header)
    const int kw[2] = {k0_wg + 32 * wave, k0_wg + 32 * (7 - wave)};
    // The pass's tile STREAM: position i = query tile t0 + 2 + i for the n_main unmasked tiles, then (causal) the two tiles
    // level with the key tile, t0 and t0 + 1 (the launcher guarantees that both exist: S_q >= S_k, whole tiles); past the end: -1
    const int t0 = CAUSAL ? k0_wg / C::BQ : 0;
    const int n_main = CAUSAL ? ntiles - t0 - 2 : ntiles, n_pos = n_main + (CAUSAL ? 2 : 0);
    auto tile_at = [&](int i) __attribute__((always_inline)) -> int {
      if constexpr (CAUSAL) return i < n_main ? t0 + 2 + i : (i < n_pos ? t0 + i - n_main : -1);
      return i < n_main ? i : -1;
    };

    // ---- LDS-DMA: a wave fills rows [32w, 32w+32) of each Q and dO tile, 2 x 4 pieces, issued in pairs ----
    int dma_q[C::PIECES], dma_do[C::PIECES];
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
      const int row = (C::BQ / C::NW) * wave + C::RPI * i + lane / C::CPR;
      const int chunk = swz_chunk<D>(row, lane % C::CPR) * 16;
      dma_q[i] = row * q_rs + chunk - 1024 * (i & 1);     // dma_pieces: immediate taken out
      dma_do[i] = row * do_rs + chunk - 1024 * (i & 1);
    }
    // group g4: 0, 1 = the Q pairs, 2, 3 = the dO pairs of this wave's share of tile t (ring slot `buf`); a tile past the
    // last one is out of range for the descriptor (no branch: hipcc sinks code across branches, fa_bwd_dkv_v3.hip)
    auto dma_tile = [&](__amdgpu_buffer_rsrc_t rq_, __amdgpu_buffer_rsrc_t rdo_, int t, int buf, int g4) __attribute__((always_inline)) {
      const int i = 2 * (g4 & 1);
      const int dst = buf * C::TILE_BYTES + ((C::BQ / C::NW) * wave + C::RPI * i) * C::ROWB;
      const int sq = t >= 0 ? t * C::BQ * q_rs : C::kOOB, sd = t >= 0 ? t * C::BQ * do_rs : C::kOOB;
      if (g4 < 2) dma_pieces<2>(rq_, (unsigned)(lds0 + dst), dma_q + i, sq);
      else dma_pieces<2>(rdo_, (unsigned)(lds0 + C::DO_BASE + dst), dma_do + i, sd);
    };
    auto dma_group = [&](int pos, int buf, int g4) __attribute__((always_inline)) { dma_tile(rq, rdo, tile_at(pos), buf, g4); };
    // the row constant of this thread for tile t: requested into a[192] (nothing architectural waits for it), read back and
    // published -- scaled, negated; rows past S_q cannot occur (whole tiles) -- by the commit that makes tile t current
    auto rc_tile = [&](__amdgpu_buffer_rsrc_t rrc_, auto a_tag, int t) __attribute__((always_inline)) {
      int x;
      asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "n"(C::BQ - 1), "v"(tid));   // row of the tile, re-derived (nothing kept live)
      pf_load4<decltype(a_tag)::value>(rrc_, t >= 0 ? (t * C::BQ + x) * 4 : C::kOOB);
    };
    auto rc_request = [&](auto a_tag, int pos) __attribute__((always_inline)) { rc_tile(rrc, a_tag, tile_at(pos)); };
    auto rc_publish = [&](auto a_tag, int buf) __attribute__((always_inline)) {
      const float v = acc_read1<decltype(a_tag)::value>();
      FA_LDS float* rcp = (FA_LDS float*)(smem + C::ROWC_OFF + buf * C::ROWC_BYTES);
      rcp[tid] = rc_lse ? -v * kLog2e : -v;   // rcp[row] = -LSE*log2e, rcp[BQ + row] = -delta
    };
    using RC0 = std::integral_constant<int, C::A_RC>;
    using RC1 = std::integral_constant<int, C::A_RC + 1>;

    // ---- the staged K / V rows; the ring: stream positions 0 and 1 whole, the first half of 2 once V is consumed ----
    if (!staged) {   // the first pass of a workgroup (and every pass of a non-causal launch): nothing to hide behind
      const Stage st = stage_of(wk, pass, true);
#pragma unroll
      for (int g = 0; g < 4; ++g) stage_group(st, b2, g);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) dma_group(0, b0, g4);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) dma_group(1, b1, g4);
      rc_request(RC0{}, 0);
    } else {
      // staged from inside the previous pass, and behind them -- requested in front of its epilogue -- this pass's first two
      // tiles and first row constant (17 requests), then the epilogue's 16 stores: the staged rows have landed
      asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
    }
    int row_off[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
    // resident B operands: K^T and V^T fragments of this wave's two key groups (row reads of the staged rows), pinned
    static_for<2 * C::KS>([&](auto i_) __attribute__((always_inline)) {
      constexpr int i = decltype(i_)::value, g = i / C::KS, ks = i % C::KS;
      vec8 kk = as_vec8<T>(lds_read16(smem + C::KS_OFF + (2 * wave + g) * C::KG_BYTES + row_off[ks]));
      if (FOLD && !p.q_prescaled) kk = scale_frag<T>(kk, c2);  // K * softmax_scale * log2(e)
      pin_write<C::F_K + 4 * g + ks>(__builtin_bit_cast(u32x4, kk));
      pin_write<C::F_V + 4 * g + ks>(lds_read16(smem + (g ? C::DO_BASE : 0) + b2 * C::TILE_BYTES + wave * C::KG_BYTES + row_off[ks]));
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([](auto i_) __attribute__((always_inline)) { acc_zero16<16 * decltype(i_)::value>(); });

    // ---- loop-invariant per-lane LDS offsets ----
    int tr_off[2][C::DB];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int db = 0; db < C::DB; ++db) tr_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);

    // ---- the pipeline: state carried from block to block, tile to tile ----
    f32x16 S_[2], P_[2];   // [set]: score / dP accumulators of the block in flight; the OTHER set holds the previous block's
                           // exponent arguments -> P and dP - delta -> dS
    f32x16 NL[FOLD ? 1 : 2], ND;   // row constants of a query block ([parity] for NL: the exact-fma path (fp16) needs the
                                   // previous block's -LSE*log2e while the next block's is being read)
    f32x16 CD;             // chain start of a diagonal block: dead ? -inf : (-LSE*log2e | 0)
    u32x4 RF[8];           // row fragments of the current query block: Q k-steps 0-3, dO k-steps 0-3
    vec8 TF[8];            // transposed fragments of the previous block's query block: dO^T (db, e) 0-3, Q^T 4-7
    u32x4 pk[2][2], sk[2][2];   // [set][k-step]: packed P and dS of a block

    // VALU work of ONE block at pipeline time tau (family 3's timetable: one exp, one dS multiply, one pack per slot):
    //   exp e at tau = 5 + e, pack P pair j at 7 + 2j, dS e at tau = 10 + e, pack dS pair j at 12 + 2j
    auto block_valu = [&](int tau, f32x16& X, f32x16& Y, u32x4 (&pkb)[2], u32x4 (&skb)[2], const f32x16& nl) __attribute__((always_inline)) {
      const int e = tau - 5, m = tau - 10;
      if (e >= 0 && e < 16) {
        // every exp behind its slot's MFMA (fa_common.h here): left free, the instruction selector moved exps of the diagonal
        // phase to four wait states behind the chain that writes their argument (tools/mfma_lint.py R3; wrong results on the
        // GPU).  The s_nop hipcc pads behind each such statement costs 0.5 % here (one exp per slot).
        float x = here(X[e]);
        if constexpr (!FOLD) x = __builtin_fmaf(x, c2, nl[e]);
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
    // dV^T / dK^T MFMA n = 2 e + db of the previous block (key group PKG): (k-step e, d block db); pk[.][0] is complete first
    auto dv_mfma = [&](auto pkg_tag, int n, const u32x4 (&pkb)[2]) __attribute__((always_inline)) {
      constexpr int A = C::ACC_DV + 32 * decltype(pkg_tag)::value;
      const u32x4 a = __builtin_bit_cast(u32x4, TF[2 * (n & 1) + (n >> 1)]);
      if (n == 0) MfmaPin::template into<A>((T*)nullptr, a, pkb[0]);
      else if (n == 1) MfmaPin::template into<A + 16>((T*)nullptr, a, pkb[0]);
      else if (n == 2) MfmaPin::template into<A>((T*)nullptr, a, pkb[1]);
      else MfmaPin::template into<A + 16>((T*)nullptr, a, pkb[1]);
    };
    auto dk_mfma = [&](auto pkg_tag, int n, const u32x4 (&skb)[2]) __attribute__((always_inline)) {
      constexpr int A = C::ACC_DK + 32 * decltype(pkg_tag)::value;
      const u32x4 a = __builtin_bit_cast(u32x4, TF[4 + 2 * (n & 1) + (n >> 1)]);
      if (n == 0) MfmaPin::template into<A>((T*)nullptr, a, skb[0]);
      else if (n == 1) MfmaPin::template into<A + 16>((T*)nullptr, a, skb[0]);
      else if (n == 2) MfmaPin::template into<A>((T*)nullptr, a, skb[1]);
      else MfmaPin::template into<A + 16>((T*)nullptr, a, skb[1]);
    };
    // One block iteration.  G: accumulator set of this block (the previous block's is G ^ 1); KG: its key group; PKG: the
    // previous block's; QPAR: parity of its query block (the fp16 path's NL set).  What it reloads, each into registers whose
    // last use is just over:  RD_DOT  dO^T fragments of THIS query block (`cur` + DO_BASE) under slots 12-15 -- a pair's first
    //   iteration -- or, SOLO, under 8-11;   RD_QT  Q^T fragments of this query block under slots 0-3 -- a pair's second
    //   iteration -- or, SOLO, under 12-15;   RD_NEXT  row fragments and row constants of the NEXT query block (`nxt`, `rcn`)
    //   under slots 4-11.  (`trb` + t_imm, `nxt` + n_imm: per-lane base registers + immediates.)  DIAG: the score chain starts
    //   from CD.  hook(s, 0) runs before the slot's MFMA, hook(s, 1) after it.
    auto block_iter = [&](auto g_tag, auto kg_tag, auto pkg_tag, auto qpar_tag, auto solo_tag, auto next_tag, auto diag_tag,
                          const int (&trb)[2][C::DB], int t_imm, const int (&nxt)[C::KS], int n_imm, int rcn,
                          auto&& hook) __attribute__((always_inline)) {
      constexpr int G = decltype(g_tag)::value, PG = G ^ 1, KG = decltype(kg_tag)::value, QPAR = decltype(qpar_tag)::value;
      constexpr bool SOLO = decltype(solo_tag)::value, RD_NEXT = decltype(next_tag)::value, DIAG = decltype(diag_tag)::value;
      constexpr bool RD_DOT = SOLO || KG == 0, RD_QT = SOLO || KG == 1;
      constexpr int NLc = FOLD ? 0 : QPAR, NLn = FOLD ? 0 : (QPAR ^ 1);
      // the previous block's query block has the parity of this one when this is a pair's second iteration, else the other
      constexpr int NLp = FOLD ? 0 : ((!SOLO && KG == 1) ? QPAR : (QPAR ^ 1));
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        hook(s, 0);
        // ---- the MFMA of this slot ----
        if (s == 0) {
          if constexpr (DIAG) MfmaPin::template first<C::F_K + 4 * KG>((T*)nullptr, S_[G], RF[0], CD);
          else if constexpr (FOLD) MfmaPin::template first<C::F_K + 4 * KG>((T*)nullptr, S_[G], RF[0], NL[NLc]);
          else MfmaPin::template first0<C::F_K + 4 * KG>((T*)nullptr, S_[G], RF[0]);
        } else if (s == 1) { MfmaPin::template acc<C::F_K + 4 * KG + 1>((T*)nullptr, S_[G], RF[1]);
        } else if (s == 2) { MfmaPin::template acc<C::F_K + 4 * KG + 2>((T*)nullptr, S_[G], RF[2]);
        } else if (s == 3) { MfmaPin::template acc<C::F_K + 4 * KG + 3>((T*)nullptr, S_[G], RF[3]);
        } else if (s == 4) { MfmaPin::template first<C::F_V + 4 * KG>((T*)nullptr, P_[G], RF[4], ND);
        } else if (s == 5) { MfmaPin::template acc<C::F_V + 4 * KG + 1>((T*)nullptr, P_[G], RF[5]);
        } else if (s == 6) { MfmaPin::template acc<C::F_V + 4 * KG + 2>((T*)nullptr, P_[G], RF[6]);
        } else if (s == 7) { MfmaPin::template acc<C::F_V + 4 * KG + 3>((T*)nullptr, P_[G], RF[7]);
        } else if (s < 12) { dv_mfma(pkg_tag, s - 8, pk[PG]);
        } else { dk_mfma(pkg_tag, s - 12, sk[PG]);
        }
        hook(s, 1);
        // ---- LDS reads into registers whose last use is just over ----
        if constexpr (RD_QT) {
          const int s0 = SOLO ? 12 : 0;
          if (s >= s0 && s < s0 + 4) {
            // (solo: the fragment the dK MFMA of this very slot has just consumed -- MFMA n reads fragment 2 (n & 1) + (n >> 1))
            const int n = SOLO ? 2 * ((s - s0) & 1) + ((s - s0) >> 1) : s - s0, db = n >> 1, e = n & 1;
            TF[4 + n] = lds_read_tr_frag<T>(lds_at(trb[0][db] + t_imm + 16 * e * C::ROWB), lds_at(trb[1][db] + t_imm + 16 * e * C::ROWB));
          }
        }
        if constexpr (RD_DOT) {
          const int s0 = SOLO ? 8 : 12;
          if (s >= s0 && s < s0 + 4) {
            const int n = SOLO ? 2 * ((s - s0) & 1) + ((s - s0) >> 1) : s - s0, db = n >> 1, e = n & 1;
            TF[n] = lds_read_tr_frag<T>(lds_at(trb[0][db] + t_imm + C::DO_BASE + 16 * e * C::ROWB),
                                        lds_at(trb[1][db] + t_imm + C::DO_BASE + 16 * e * C::ROWB));
          }
        }
        if constexpr (RD_NEXT) {
          if (s >= 4 && s < 12) {   // both uses of the old content are over (-LSE*log2e last used as C at slot 0, -delta at slot 4)
            const int f = s - 4;
            RF[f] = lds_read16(lds_at(nxt[f & 3] + n_imm + (f < 4 ? 0 : C::DO_BASE)));
            const f32x4 v = *(const FA_LDS f32x4*)lds_at(rcn + ((s < 8 ? 0 : C::BQ) + 8 * (f & 3)) * 4);   // (rcn: this lane's 4 h folded in)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (s < 8) NL[NLn][4 * (f & 3) + j] = v[j];
              else ND[4 * (f & 3) + j] = v[j];
            }
          }
        }
        // ---- VALU: the previous block at tau = 16 + s, this block at tau = s ----
        block_valu(16 + s, S_[PG], P_[PG], pk[PG], sk[PG], NL[NLp]);
        block_valu(s, S_[G], P_[G], pk[G], sk[G], NL[NLc]);
        // An MFMA reads its C operand over its passes (hipcc pads that hazard for its own MFMAs only): a chain-start block that
        // is dead after this use would have its registers reused at once (fa_bwd_dkv_v3.hip) -- live one more slot
        if (s == 1) {
          if constexpr (DIAG) keep_live(CD);
          else if constexpr (FOLD) keep_live(NL[0]);
        }
        if (s == 5) keep_live(ND);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the last block's exp / dS / dV^T / dK^T once nothing follows it
    auto pipe_drain = [&](auto g_tag, auto kg_tag, auto qpar_tag) __attribute__((always_inline)) {
      constexpr int G = decltype(g_tag)::value, QPAR = decltype(qpar_tag)::value;
#pragma unroll
      for (int s = 0; s < C::NS; ++s) {
        if (s >= 8 && s < 12) dv_mfma(kg_tag, s - 8, pk[G]);
        else if (s >= 12) dk_mfma(kg_tag, s - 12, sk[G]);
        block_valu(16 + s, S_[G], P_[G], pk[G], sk[G], NL[FOLD ? 0 : QPAR]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto no_hook = [](int, int) __attribute__((always_inline)) {};
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Yes = std::true_type;
    using No = std::false_type;
    // per-lane read bases of the query block at LDS byte offset `off` (opaque: fa_fwd_v4.hip)
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
    // row fragments and row constants of query block `off` / `rc_off` straight into RF / NL[par] / ND (pipeline fill)
    auto load_block = [&](int off, int rc_off, int par) __attribute__((always_inline)) {
      par &= 1;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        RF[ks] = lds_read16(smem + off + row_off[ks]);
        RF[4 + ks] = lds_read16(smem + C::DO_BASE + off + row_off[ks]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a = *(const FA_LDS f32x4*)(smem + rc_off + (8 * q + 4 * h) * 4);
        const f32x4 d = *(const FA_LDS f32x4*)(smem + rc_off + (C::BQ + 8 * q + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          NL[FOLD ? 0 : par][4 * q + j] = a[j];
          ND[4 * q + j] = d[j];
        }
      }
    };
    // the neutral "previous block" of the first iteration (P = 0, dS = 0 * 0, packed P / dS and fragments 0) in set PG
    auto pipe_fill = [&](auto pg_tag, int prev_par) __attribute__((always_inline)) {
      constexpr int PG = decltype(pg_tag)::value;
      // a zero hipcc cannot see through (the lane id is below 64): with constants it folds the neutral block's arithmetic
      // wherever the first tile step is a written-out copy, keeps the zero fragments in a scalar register and splats them
      // into place right in front of the asm MFMA that reads them (tools/mfma_lint.py R1; fa_bwd_dq_v4.hip met it)
      const unsigned lz = (unsigned)lane_id_now() >> 6;
      const float lzf = __builtin_bit_cast(float, lz);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        S_[PG][i] = i < 11 ? lzf : -INFINITY + lzf;   // exps 0-10 of a block are over when its iteration ends; 11-15 follow
        P_[PG][i] = lzf;
        if (!FOLD) NL[FOLD ? 0 : (prev_par & 1)][i] = lzf;   // the neutral block's -LSE*log2e (fp16 path)
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        pk[PG][e] = u32x4{lz, lz, lz, lz};
        sk[PG][e] = u32x4{lz, lz, lz, lz};
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) TF[n] = as_vec8<T>(u32x4{lz, lz, lz, lz});
    };

    // ---- stream position 0 is current: positions 0 and 1 and the first row constant have landed ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rc_publish(RC0{}, b0);
    rc_request(RC0{}, 1);   // the steady state keeps ONE row-constant request in flight, in a[192]
    // The V rows are consumed (every read above has fed a register write): this wave's part of slot b2 takes position 2 --
    // requested AFTER the row constant of position 1, the order of the steady state (a commit requests the next row
    // constant, then the Q pairs, then the dO pairs): the first commit's vmcnt(8) covers the eight newest requests, and with
    // the Q pairs of position 2 in front of it the row constant was one of them -- published from a[192] before it had
    // landed once in ~30 launches under load (tools/race_stress.py; round 4), a whole dK / dV tile wrong.
    dma_group(2, b2, 0);
    dma_group(2, b2, 1);
    asm volatile("s_nop 4");  // v_accvgpr_write -> MFMA operand wait states (hipcc pads nothing around asm)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the row constants are written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    load_block(b0 * C::TILE_BYTES, C::ROWC_OFF + b0 * C::ROWC_BYTES, 0);
    pipe_fill(I1{}, 1);
    __builtin_amdgcn_sched_barrier(0);

    FA4K_STAMP(0);
    // ---- the unmasked tiles: eight block iterations per tile, ring slots rotate ----
    // Non-causal launches are persistent too; with no diagonal phase to stage the next item's K / V rows from, they ride in
    // the LDS-DMA slots of a pass's last tile steps, which have no tile left to fetch (fa_bwd_dq_v4.hip does the same) --
    // SEL 1 (position n_main - 3): the V rows into the ring slot of the position that does not exist (free since this step's
    // commit; three rotations and one more make it the next pass's b2), SEL 2: the K rows into their staging area, SEL 3:
    // nothing.  SEL 2's commit counts the sixteen staging pieces in flight instead of the eight tile pieces.
    const bool nc_stage = !CAUSAL && item + (int)gridDim.x < n_items && n_main >= 3;
    auto tile_step = [&](int i, auto sel_tag) __attribute__((always_inline)) {
      constexpr int SEL = decltype(sel_tag)::value;
      // (defined HERE: a descriptor captured through two levels of closures goes through memory and comes back as a vector)
      const Stage est = stage_of(nwk_item, 0, SEL != 0);
#ifdef FA_STAMPS
      ++ntile_;
#endif
      int tA[2][C::DB], kq[C::KS], kN[C::KS];
      tr_bases(tA, b0 * C::TILE_BYTES);
      row_bases(kq, b0 * C::TILE_BYTES);
      row_bases(kN, b1 * C::TILE_BYTES);
      // row constants: per-lane base (its 4 h registers' worth of offset folded in) + immediates
      const int rcA = opaque(lds0 + C::ROWC_OFF + b0 * C::ROWC_BYTES + 16 * h), rcN = opaque(lds0 + C::ROWC_OFF + b1 * C::ROWC_BYTES + 16 * h);
      // position i + 2's dO pairs ride in the first iteration (its Q pairs went out in the previous tile's last one)
      auto hook_first = [&](int s, int phase) __attribute__((always_inline)) {
        if constexpr (SEL == 0 || SEL == 1) {
          if (phase == 1 && s == 9) dma_group(i + 2, b2, 2);
          if (phase == 1 && s == 13) dma_group(i + 2, b2, 3);
        } else if constexpr (SEL == 2) {
          if (phase == 1 && s == 9) stage_group(est, b2, 0);
          if (phase == 1 && s == 13) stage_group(est, b2, 1);
        }
      };
      // the commit: position i + 1 and its row constant have landed for every wave (vmcnt(8): the eight pieces of position
      // i + 2 may still fly), every read of position i is issued (slot 4 of the last iteration) and complete (lgkmcnt(0)); the
      // row constants of position i + 1 are published, and the ring slot of position i takes position i + 3
      auto hook_last = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 0 && s == 4) {
          if constexpr (SEL == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // 8 + 8 staging pieces, younger than the row constant
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          rc_publish(RC0{}, b1);
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          rc_request(RC0{}, i + 2);
        }
        if constexpr (SEL == 0) {
          if (phase == 1 && s == 9) dma_group(i + 3, b0, 0);
          if (phase == 1 && s == 13) dma_group(i + 3, b0, 1);
        } else if constexpr (SEL == 1) {   // slot b0 (position i) is free from the commit above on: the V rows of the next item
          if (phase == 1 && s == 9) stage_group(est, b0, 2);
          if (phase == 1 && s == 13) stage_group(est, b0, 3);
        }
      };
      block_iter(I0{}, I0{}, I1{}, I0{}, No{}, No{}, No{}, tA, 0 * C::QBLK, kq, 0, 0, hook_first);
      block_iter(I1{}, I1{}, I0{}, I0{}, No{}, Yes{}, No{}, tA, 0 * C::QBLK, kq, 1 * C::QBLK, rcA + 1 * 32 * 4, no_hook);
      block_iter(I0{}, I0{}, I1{}, I1{}, No{}, No{}, No{}, tA, 1 * C::QBLK, kq, 0, 0, no_hook);
      block_iter(I1{}, I1{}, I0{}, I1{}, No{}, Yes{}, No{}, tA, 1 * C::QBLK, kq, 2 * C::QBLK, rcA + 2 * 32 * 4, no_hook);
      block_iter(I0{}, I0{}, I1{}, I0{}, No{}, No{}, No{}, tA, 2 * C::QBLK, kq, 0, 0, no_hook);
      block_iter(I1{}, I1{}, I0{}, I0{}, No{}, Yes{}, No{}, tA, 2 * C::QBLK, kq, 3 * C::QBLK, rcA + 3 * 32 * 4, no_hook);
      block_iter(I0{}, I0{}, I1{}, I1{}, No{}, No{}, No{}, tA, 3 * C::QBLK, kq, 0, 0, no_hook);
      block_iter(I1{}, I1{}, I0{}, I1{}, No{}, Yes{}, No{}, tA, 3 * C::QBLK, kN, 0, rcN, hook_last);
      const int bt = b0;
      b0 = b1;
      b1 = b2;
      b2 = bt;
    };
    {
      int i = 0;
      const int nmain = nc_stage ? n_main - 3 : n_main;
      for (; i < nmain; ++i) tile_step(i, I0{});
      if constexpr (!CAUSAL) {
        if (nc_stage) {
          tile_step(i, I1{});
          tile_step(i + 1, std::integral_constant<int, 2>{});
          tile_step(i + 2, std::integral_constant<int, 3>{});
        }
      }
    }

    FA4K_STAMP(1);
    if constexpr (!CAUSAL) {
      pipe_drain(I1{}, I1{}, I1{});
      if (nc_stage) {   // the V rows sit in SEL 1's b0, now b0 again: one more rotation makes it b2, where the next pass looks
        const int bt = b0;
        b0 = b1;
        b1 = b2;
        b2 = bt;
      }
      staged = nc_stage;
    } else {
      // ---- the 256 query rows level with the key tile: slot b0 (landed, published) and b1 (requested two tile steps ago) ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      rc_publish(RC0{}, b1);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // From here to the end of the pass every wave is on its own (see the header).  Query blocks q = 7 .. 0 of the region:
      // LDS byte offset of block q, LDS address of its row constants (this lane's 4 h folded in)
      auto qbase = [&](int q) __attribute__((always_inline)) {
        return (q < C::QB ? b0 * C::TILE_BYTES : b1 * C::TILE_BYTES - C::QB * C::QBLK) + q * C::QBLK;
      };
      auto rcaddr = [&](int q) __attribute__((always_inline)) {
        return lds0 + C::ROWC_OFF + (q < C::QB ? b0 : b1) * C::ROWC_BYTES + (q & (C::QB - 1)) * 32 * 4 + 16 * h;
      };
      // chain start of a diagonal block: score register i of lane (r, h) is key r against query c_i + 4h of the block
      auto diag_start = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) CD[i] = (r > (i & 3) + 8 * (i >> 2) + 4 * h) ? -INFINITY : (FOLD ? NL[0][i] : 0.f);
        (void)par;
        settle_mfma(CD);   // VALU write -> asm MFMA operand: hipcc pads nothing in front of an asm statement
      };
      // (only instantiated where the scale is folded -- bf16: the query-block parity, which selects the fp16 path's NL set,
      // plays no part, and every block below passes parity 0)
      static_assert(FOLD, "the causal instance exists for the folded-scale dtype only (launch_bwd_dkv_v4)");
      // No vmcnt wait from here to the end of the pass and ring slot b2 is free: the place to stage the NEXT pass's K / V rows
      // (the pair's second pass, or the first pass of the workgroup's next item).  The four piece groups ride in the three
      // block iterations below that every wave executes exactly once; with nothing to follow the descriptors are empty.
      const bool more_pass = pass + 1 < npass, more_item = item + (int)gridDim.x < n_items;
      const Stage nst = stage_of(more_pass ? wk : nwk_item, more_pass ? pass + 1 : 0,
                                 more_pass || more_item);
      auto hook_a = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 2) stage_group(nst, b2, 0);
        if (phase == 1 && s == 10) stage_group(nst, b2, 1);
      };
      auto hook_b = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 6) stage_group(nst, b2, 2);
      };
      auto hook_c = [&](int s, int phase) __attribute__((always_inline)) {
        if (phase == 1 && s == 4) stage_group(nst, b2, 3);
      };
      staged = more_pass || more_item;
      int trb[2][C::DB], nxt[C::KS];
      // the last unmasked iteration prefetched the fragments of the block that FOLLOWS in memory; the phase starts at block 7
      load_block(qbase(7), rcaddr(7) - lds0 - 16 * h, 1);
      __builtin_amdgcn_sched_barrier(0);
      // both key groups: query blocks 7 .. 7 - wave, key group 1 (group 7 - wave) last on its diagonal
      int q = 7;
      for (; q > 7 - wave; --q) {
        tr_bases(trb, qbase(q));
        row_bases(nxt, qbase(q - 1));
        block_iter(I0{}, I0{}, I1{}, I0{}, No{}, No{}, No{}, trb, 0, nxt, 0, 0, no_hook);
        block_iter(I1{}, I1{}, I0{}, I0{}, No{}, Yes{}, No{}, trb, 0, nxt, 0, rcaddr(q - 1), no_hook);
      }
      // q = 7 - wave
      tr_bases(trb, qbase(q));
      row_bases(nxt, qbase(max(q - 1, 0)));
      block_iter(I0{}, I0{}, I1{}, I0{}, No{}, No{}, No{}, trb, 0, nxt, 0, 0, hook_a);
      diag_start(q);
      block_iter(I1{}, I1{}, I0{}, I0{}, No{}, Yes{}, Yes{}, trb, 0, nxt, 0, rcaddr(max(q - 1, 0)), hook_b);
      // key group 0 (group `wave`) alone: query blocks 6 - wave .. wave (an odd number: sets 0, 1, .., 0), the last on its
      // diagonal.  The first solo follows the pair's second iteration (key group 1), the others follow a solo (key group 0).
      // (offsets evaluated by the caller: a lambda that captures the qbase / rcaddr closures puts the ring slots they refer to
      // into memory, and hipcc then reads b0 / b1 back from scratch as per-lane values)
      auto solo = [&](auto g_tag, auto pkg_tag, auto next_tag, auto diag_tag, int off, int off_next, int rc_next, auto&& hook) __attribute__((always_inline)) {
        tr_bases(trb, off);
        if constexpr (decltype(next_tag)::value) row_bases(nxt, off_next);
        if constexpr (decltype(diag_tag)::value) diag_start(0);
        block_iter(g_tag, I0{}, pkg_tag, I0{}, Yes{}, next_tag, diag_tag, trb, 0, nxt, 0, rc_next, hook);
      };
      if (6 - wave <= wave) {   // wave 3: its only solo
        solo(I0{}, I1{}, No{}, Yes{}, qbase(wave), 0, 0, hook_c);
      } else {
        q = 6 - wave;
        solo(I0{}, I1{}, Yes{}, No{}, qbase(q), qbase(q - 1), rcaddr(q - 1), no_hook);
        for (q = 5 - wave; q > wave + 1; q -= 2) {
          solo(I1{}, I0{}, Yes{}, No{}, qbase(q), qbase(q - 1), rcaddr(q - 1), no_hook);
          solo(I0{}, I0{}, Yes{}, No{}, qbase(q - 1), qbase(q - 2), rcaddr(q - 2), no_hook);
        }
        solo(I1{}, I0{}, Yes{}, No{}, qbase(wave + 1), qbase(wave), rcaddr(wave), no_hook);
        solo(I0{}, I0{}, No{}, Yes{}, qbase(wave), 0, 0, hook_c);
      }
      FA4K_STAMP(2);
      pipe_drain(I0{}, I0{}, I0{});
    }
    FA4K_STAMP(3);
    // non-causal: the (out-of-range) fetches past the last tile are over before the ring is reused (causal: they were, at
    // the start of the diagonal phase -- what is in flight now are the NEXT pass's staged rows, which land outside the staging area)
    if constexpr (!CAUSAL) {
      if (!staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    __syncthreads();  // every wave is done with the tile buffers: they become the staging area
    {
      if (staged) {   // something follows: its first two tiles and its first row constant are requested NOW -- the ring is free
                      // since the barrier above -- and land while dK / dV are written out
        const bool more_pass = pass + 1 < npass;
        const Work nw = more_pass ? wk : nwk_item;
        const int t0n = ktile_of(nw, more_pass ? pass + 1 : 0) * C::BK / C::BQ, n_mainn = ntiles - t0n - 2;
        // (non-causal: the stream of a pass is tiles 0, 1, 2, ...)
        const int tn0 = !CAUSAL ? 0 : (n_mainn > 0 ? t0n + 2 : t0n), tn1 = !CAUSAL ? 1 : (n_mainn > 1 ? t0n + 3 : (n_mainn == 1 ? t0n : t0n + 1));
        const __amdgpu_buffer_rsrc_t nrq = make_rsrc((const char*)p.q + nw.b * p.lq.sb + nw.h * p.lq.sh, view_bytes(Sq, q_rs, C::ROWB));
        const __amdgpu_buffer_rsrc_t nrdo = make_rsrc((const char*)p.dout + nw.b * p.ldo.sb + nw.h * p.ldo.sh, view_bytes(Sq, do_rs, C::ROWB));
        const __amdgpu_buffer_rsrc_t nrrc = make_rsrc((rc_lse ? p.lse : p.delta) + nw.b * p.lse_sb + nw.h * p.lse_sh, (unsigned)Sq * 4);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dma_tile(nrq, nrdo, tn0, b0, g4);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dma_tile(nrq, nrdo, tn1, b1, g4);
        rc_tile(nrrc, RC0{}, tn0);
      }
    }
    FA_LDS char* stage = smem + C::EPI_OFF + wave * C::KG_BYTES;
    // dK = dS^T Q * scale; with the pre-scaled Q (= Q * scale * log2e) in LDS that is dS^T Q' * ln 2
    const float dk_mul = (FOLD && p.q_prescaled) ? kLn2 : p.scale;
    {
      const int dk_rs = p.ldk.rs, dv_rs = p.ldv.rs;
      const __amdgpu_buffer_rsrc_t rdk = make_rsrc((char*)p.dk + b_ * p.ldk.sb + h_ * p.ldk.sh, view_bytes(Sk, dk_rs, C::ROWB));
      const __amdgpu_buffer_rsrc_t rdv = make_rsrc((char*)p.dv + b_ * p.ldv.sb + h_ * p.ldv.sh, view_bytes(Sk, dv_rs, C::ROWB));
      const f32x16 dk0[C::DB] = {acc_read16<C::ACC_DK>(), acc_read16<C::ACC_DK + 16>()};
      store_tile_rows<D, T>(dk0, dk_mul, stage, rdk, kw[0] * dk_rs, lane, dk_rs);
      const f32x16 dv0[C::DB] = {acc_read16<C::ACC_DV>(), acc_read16<C::ACC_DV + 16>()};
      store_tile_rows<D, T>(dv0, 1.0f, stage, rdv, kw[0] * dv_rs, lane, dv_rs);
      const f32x16 dk1[C::DB] = {acc_read16<C::ACC_DK + 32>(), acc_read16<C::ACC_DK + 48>()};
      store_tile_rows<D, T>(dk1, dk_mul, stage, rdk, kw[1] * dk_rs, lane, dk_rs);
      const f32x16 dv1[C::DB] = {acc_read16<C::ACC_DV + 32>(), acc_read16<C::ACC_DV + 48>()};
      store_tile_rows<D, T>(dv1, 1.0f, stage, rdv, kw[1] * dv_rs, lane, dv_rs);
    }
    FA4K_STAMP(4);
#ifdef FA_STAMPS
    ++npass_;
#endif
  }  // pass
  }  // item
#ifdef FA_STAMPS
  if (p.dbg && (threadIdx.x & 63) == 0) {
    unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 4 + wave) * 32;
    for (int i = 0; i < 8; ++i) d[i] = seg[i];
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
  using C = Dkv4Cfg;
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
  auto kern = fa_bwd_dkv4_kernel<T, CAUSAL>;
  static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
  if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dkv_v4(BwdParams p, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sk + Dkv4Cfg::BK - 1) / Dkv4Cfg::BK;
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
  p.div_per_bh = make_fastdiv(p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles);
  p.div_h = make_fastdiv(p.H);
  if (dtype == 1) return causal ? launch4<BF16, true>(p, s) : launch4<BF16, false>(p, s);
  // (fp16 causal stays with family 3: the exact-fma path keeps two sets of -LSE*log2e blocks, and with the diagonal phase's
  // chain start on top the kernel spills 500 registers; fa_kernels.h pick_dkv_impl never sends it here)
  return causal ? hipErrorInvalidValue : launch4<FP16, false>(p, s);
}

}  // namespace fa
