// FlashAttention forward for gfx950: O = softmax(Q K^T / sqrt(D) [+causal]) V, LSE = logsumexp.
//
// Replaces the reference's flash_attention_forward_kernel
// (code/_flash_attention_kernel_optimized.py:35-129); semantics kept: fp32 scores and
// softmax state, l sums the un-rounded p (K:111), P is rounded to the input dtype for
// P@V (K:115), O = o / l cast on store (K:120-123), LSE = m + ln(l) (K:126),
// top-left aligned causal mask (K:102), keys >= S_k masked (K:94).
// Deviations inside the stated tolerance: the scale multiplies the fp32 accumulator once (fp16) or is folded into the
// resident Q fragments (bf16, fa_common.h kFoldScale); the rescale is deferred / lazy (tile(), tile_lazy()).
//
// Work decomposition (CDNA4-first, not the reference's 64x64 Triton tiles):
//   workgroup = 4 waves = 128 query rows of one (batch, head); wave = 32 query rows.
//   K/V stream through LDS in 64-key tiles (double buffered, swizzled image, one
//   barrier per tile).  Scores are computed TRANSPOSED, S^T = K Q^T, so the query
//   index sits on the MFMA lane: the online-softmax row max / row sum are in-lane
//   reductions plus one lane<->lane+32 exchange, the rescale is a per-lane scalar,
//   and the fp32 P^T accumulator is, after rounding, directly the B operand of
//   O^T += V^T P^T (V^T fetched with ds_read_b64_tr_b16) -- P never touches LDS.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

// A/B hooks (tools/build_variant.sh): FA_FWD_PRIO 1 = raise the wave's priority over its MFMA chains, 2 = over its
// softmax (VALU) phase; FA_FWD_OCC = workgroups per CU the D = 64 register allocation is held to.
#ifndef FA_FWD_PRIO
#define FA_FWD_PRIO 1
#endif
#ifndef FA_FWD_OCC
#define FA_FWD_OCC 3
#endif
#define FA_PRIO_MFMA(on) do { if (FA_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(on); } while (0)
#define FA_PRIO_VALU(on) do { if (FA_FWD_PRIO == 2) __builtin_amdgcn_s_setprio(on); } while (0)

template <int D>
struct FwdCfg {
  static constexpr int BM = 128;           // query rows per workgroup
  static constexpr int BN = 64;            // keys per LDS tile
  static constexpr int NT = 256;           // threads
  static constexpr int ROWB = D * 2;       // bytes per row
  static constexpr int CPR = D / 8;        // 16-byte chunks per row
  static constexpr int KS = D / 16;        // k-steps of S^T = K Q^T
  static constexpr int DB = D / 32;        // 32-wide d blocks of O^T
  static constexpr int TILE_BYTES = BN * ROWB;
  static constexpr int DMA_PER_MAT = TILE_BYTES / (4 * 1024);  // 1-KiB LDS-DMA instructions per wave per matrix
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;  // K[2], V[2]
};

// Online-softmax rescale is deferred until a row max grows by more than 2^kDeferLog2 (see tile()).
constexpr float kDeferLog2 = 6.0f;
// Lazy running max: largest partial row sum accepted without recomputing the true max (see tile_lazy).
constexpr float kLazySumMax = 8192.0f;

// DROP: attention dropout (fa_common.h `Dropout`): P is masked before P @ V (exact and lazy tiles alike), l keeps summing
// the undropped p (the softmax normalisation is not affected by dropout), 1 / (1 - p) joins the normalisation of O.
template <int D, typename T, bool CAUSAL, bool DROP = false>
__global__ __launch_bounds__(256, (D == 64 ? FA_FWD_OCC : 2)) void fa_fwd_kernel(FwdParams p) {
  using C = FwdCfg<D>;
  using vec8 = typename T::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // ---- which (batch*head, q tile) ----
  // Work list: non-causal -> one 128-row query tile per workgroup.  Causal -> query tile i streams i+1
  // K/V tiles, so a workgroup takes the PAIR (nq-1-i, i): every workgroup then does the same work and the
  // grid is balanced whatever the number of CUs (heavy tile first).
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.nq_tiles + 1) / 2 : p.nq_tiles;
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const BatchHead ix = batch_head(bh, p.B, p.H, p.vl.cu_q != nullptr);
  const int b_ = ix.b, h_ = ix.h;
  // variable-length launch: this sequence's rows and lengths come from cu_seqlens; the grid was sized for the longest
  // sequence, so workgroups past this one's own tile count have nothing to do
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
  if (pass) __syncthreads();  // the previous pass staged its O tile in the K/V buffers

  // inputs may be strided views with a contiguous head dim (e.g. a [B,S,H,D] buffer seen as [B,H,S,D]): per-tensor
  // batch / head byte strides, one row stride for Q and one shared by K and V; O has its own layout (contiguous for the
  // reference's launch, packed rows for varlen), LSE rows of one (batch, head) are contiguous
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


  // ---- Q^T fragments (B operand), resident for the whole kernel ----
  vec8 qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
    qf[ks] = as_vec8<T>(buf_load16(rq, (qw0 + r) * q_rs + (2 * ks + h) * 16));

  // ---- tile schedule ----
  const int kv_end = CAUSAL ? min(Sk, q0_wg + C::BM) : Sk;
  const int ntiles = (kv_end + C::BN - 1) / C::BN;
  // tiles [0, nfull) need no mask for this wave
  const int nfull = CAUSAL ? min(Sk / C::BN, qw0 / C::BN) : Sk / C::BN;

  // ---- staging addresses ----
  // ---- LDS-DMA: wave w fills rows [16w, 16w+16) of each K / V tile, 1 KiB (1024 / ROWB rows) per instruction;
  // lane p of instruction i lands on LDS row 16w + i*RPI + p/CPR, physical chunk p%CPR, so it fetches the
  // logical chunk swz(row, p%CPR) of that row (swizzle on the SOURCE address, the destination is wave-linear)
  constexpr int RPI = 1024 / C::ROWB;  // rows per DMA instruction
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = 16 * wave + RPI * i + lane / C::CPR;
    dma_src[i] = row * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16;
#ifndef FA_DMA_LEGACY
    dma_src[i] -= 1024 * i;  // dma_pieces: the immediate offset of piece i also moves the global address
#endif
  }
  // ---- fragment read addresses (loop invariant) ----
  int k_off[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) k_off[ks] = lds_off<D>(r, 2 * ks + h);
  int v_off[2][C::DB];  // [e][dblk]; key-block kb and k-step s add (32*kb + 16*s) rows
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) v_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);

  const float c2 = p.scale * kLog2e;  // exp(x*scale) = exp2(x*c2)
  // FOLD (bf16, fa_common.h): Q carries c2, the MFMA delivers scores in log2 units (cs = 1) and a lazy tile's
  // score chain starts from a block holding -m, so its exponent argument needs no VALU op at all.
  constexpr bool FOLD = T::kFoldScale;
  const float cs = FOLD ? 1.0f : c2;  // accumulator units -> log2 units
  if constexpr (FOLD) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = scale_frag<T>(qf[ks], c2);
  }
  const float defer_raw = kDeferLog2 / cs;  // rescale threshold in accumulator units
  float m = -INFINITY;                // running row max in accumulator units (raw scores, or log2 units if FOLD)
  f32x16 negm;                        // FOLD: -m in every register (this lane's query row)
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = INFINITY;
  float l = 0.f;                      // this lane's partial row sum (its 16 of every 32 keys)
  f32x16 oacc[C::DB];
#pragma unroll
  for (int db = 0; db < C::DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;

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
  // the tile fetched during this step has landed (vmcnt(0)); every wave is done with the current one
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

  // One 64-key tile for this wave.  MASKED = false: every key visible to every row.
  // BUF = 0/1: LDS buffer known at compile time (offsets fold into the ds_read immediates);
  // BUF = -1: taken from t at run time (the few masked tiles).
  // DROP: zero the dropped weights of one 32 x 32 block.  Registers 4g..4g+3 of a lane are four consecutive keys of its row =
  // the four bytes of word (row & 3) of patch g, which lane g of the quad generated (fa_common.h quad_bcast).  The factor
  // 1 / (1 - p) of the kept weights is linear in O and applied once, with the normalisation in the epilogue.
  auto drop_weights = [&](f32x16& w16, const u32x4& mine) __attribute__((always_inline)) {
    const unsigned thresh = p.drop.thresh;
    const int qsel = (qw0 + r) & 3;
    auto apply = [&](auto g_tag) __attribute__((always_inline)) {
      constexpr int g = decltype(g_tag)::value;
      const unsigned w = select_word(quad_bcast4<g>(mine), qsel);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool keep = ((w >> (8 * j)) & 255u) >= thresh;
        w16[4 * g + j] = keep ? w16[4 * g + j] : 0.f;
      }
    };
    apply(std::integral_constant<int, 0>{});
    apply(std::integral_constant<int, 1>{});
    apply(std::integral_constant<int, 2>{});
    apply(std::integral_constant<int, 3>{});
  };

  auto tile = [&](int t, auto buf_tag, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int BUF = decltype(buf_tag)::value;
    const int buf = BUF >= 0 ? BUF : (t & 1);
    const FA_LDS char* kt = smem + buf * C::TILE_BYTES;
    const FA_LDS char* vt = smem + (2 + buf) * C::TILE_BYTES;
    const int s0 = t * C::BN;
    bool use[2] = {true, true};
    if constexpr (MASKED) {
      if (CAUSAL) {
        use[0] = s0 <= qw0;        // key block start <= first row of the wave
        use[1] = s0 + 32 <= qw0;
      }
      use[0] = use[0] && s0 < Sk;
      use[1] = use[1] && s0 + 32 < Sk;
      if (!use[0] && !use[1]) return;
    }
    f32x16 sacc[2];
    // DROP: one Philox call per lane and key block -- registers 4g..4g+3 are keys s0 + 32b + 8g + 4h + 0..3 of row qw0 + r,
    // i.e. word (row & 3) of patch g, and the quad's four lanes (four consecutive rows) need the same four patches: lane j
    // generates patch g = j (fa_common.h quad_bcast).  Issued here so that its integer ops run beside the MFMA chains.
    u32x4 mine[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if constexpr (DROP) {
      const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (!(MASKED && !use[b])) mine[b] = dropout_patch(dr, (qw0 + r) >> 2, ((s0 + 32 * b + 4 * h) >> 2) + 2 * (r & 3), b_ * p.H + h_);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (MASKED && !use[b]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[b][i] = -INFINITY;
        continue;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[b][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(kt + k_off[ks] + b * 32 * C::ROWB));
        sacc[b] = T::mfma(a, qf[ks], sacc[b]);
      }
      if constexpr (MASKED) {
        const int qrow = qw0 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = s0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool dead = (CAUSAL && key > qrow) || key >= Sk;
          sacc[b][i] = dead ? -INFINITY : sacc[b][i];
        }
      }
    }
    // ---- online softmax (query on the lane) ----
    float tm0 = sacc[0][0], tm1 = sacc[1][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      tm0 = __builtin_fmaxf(tm0, sacc[0][i]);
      tm1 = __builtin_fmaxf(tm1, sacc[1][i]);
    }
    const float tm = half_max(__builtin_fmaxf(tm0, tm1));
    // Deferred rescale: the running max is only raised when some row's tile max exceeds it by
    // more than kDefer (in log2 units), so P stays <= 2^kDefer (exact in fp32, same RELATIVE
    // rounding in 16 bit) and the O-wide multiply is rare.  m = -inf (first tile) always fires.
    if (__builtin_amdgcn_ballot_w64(tm > m + defer_raw) != 0) {
      const float mn = __builtin_fmaxf(m, tm);
      const float corr = __builtin_amdgcn_exp2f((m - mn) * cs);  // m = -inf -> 0
      l *= corr;
#pragma unroll
      for (int db = 0; db < C::DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] *= corr;
      m = mn;
      if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -mn;
      }
    }
    const float mc = m * cs;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pe = __builtin_amdgcn_exp2f(FOLD ? sacc[b][i] - mc : __builtin_fmaf(sacc[b][i], c2, -mc));
        sacc[b][i] = pe;
        ls[i & 3] += pe;
      }
    l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
    if constexpr (DROP) {  // keep / drop each weight (the row sum above is the undropped one)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (!(MASKED && !use[b])) drop_weights(sacc[b], mine[b]);
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (MASKED && !use[b]) continue;
      const vec8 pf0 = pack8<T, 0>(sacc[b]);
      const vec8 pf1 = pack8<T, 1>(sacc[b]);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        const FA_LDS char* base = vt + b * 32 * C::ROWB;
        vec8 a0 = lds_read_tr_frag<T>(base + v_off[0][db], base + v_off[1][db]);
        oacc[db] = T::mfma(a0, pf0, oacc[db]);
        vec8 a1 = lds_read_tr_frag<T>(base + 16 * C::ROWB + v_off[0][db], base + 16 * C::ROWB + v_off[1][db]);
        oacc[db] = T::mfma(a1, pf1, oacc[db]);
      }
    }
  };

  // Unmasked tile with a LAZY running max: exponentiate against the stale row max -- no max reduction, no
  // cross-half exchange, no rescale test.  All p >= 0, so a lane's partial row sum bounds every p it holds: if
  // no partial sum exceeds kLazySumMax nothing can overflow (fp32 sums, 16-bit P fragments) and the stale max
  // is exactly as good as the true one (softmax is shift invariant).  Otherwise (always the first tile: m = -inf
  // gives p = +inf; afterwards only if scores jump by more than ~2^8) NOTHING has been committed: return false
  // and the caller redoes the tile on the exact path.
  // MASKED = true: the same for a tile on the causal diagonal or the ragged tail -- dead scores become -inf (p = 0),
  // key blocks no row of the wave can see are skipped.  A wave whose first visible tile is masked arrives here with
  // m = -inf, overflows by construction and takes the exact path once.
  auto tile_lazy = [&](int t, auto masked_tag) __attribute__((always_inline)) -> bool {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const FA_LDS char* kt = smem + (t & 1) * C::TILE_BYTES;
    const FA_LDS char* vt = smem + (2 + (t & 1)) * C::TILE_BYTES;
    const int s0 = t * C::BN;
    bool use[2] = {true, true};
    if constexpr (MASKED) {
      if (CAUSAL) {
        use[0] = s0 <= qw0;
        use[1] = s0 + 32 <= qw0;
      }
      use[0] = use[0] && s0 < Sk;
      use[1] = use[1] && s0 + 32 < Sk;
      if (!use[0] && !use[1]) return true;  // nothing of this tile is visible to the wave
    }
    f32x16 sacc[2];
    u32x4 mine[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if constexpr (DROP) {  // as in tile(): one Philox call per lane and key block, issued ahead of the MFMA chains
      const Dropout dr{p.drop.thresh, p.drop.seed_lo, p.drop.seed_hi, p.drop.offset, p.drop.rp};
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (!(MASKED && !use[b])) mine[b] = dropout_patch(dr, (qw0 + r) >> 2, ((s0 + 32 * b + 4 * h) >> 2) + 2 * (r & 3), b_ * p.H + h_);
    }
    FA_PRIO_MFMA(1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (MASKED && !use[b]) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[b][i] = FOLD ? negm[i] : 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        vec8 a = as_vec8<T>(lds_read16(kt + k_off[ks] + b * 32 * C::ROWB));
        sacc[b] = T::mfma(a, qf[ks], sacc[b]);
      }
    }
    FA_PRIO_MFMA(0);
    FA_PRIO_VALU(1);
    const float mc = m * c2;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (MASKED && !use[b]) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? sacc[b][i] : __builtin_fmaf(sacc[b][i], c2, -mc);
        if constexpr (MASKED) {
          const int key = s0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool dead = (CAUSAL && key > qw0 + r) || key >= Sk;
          x = dead ? -INFINITY : x;
        }
        const float pe = __builtin_amdgcn_exp2f(x);
        sacc[b][i] = pe;
        ls[i & 3] += pe;
      }
    }
    const float lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    if (__builtin_amdgcn_ballot_w64(!(lsum <= kLazySumMax)) != 0) {
      FA_PRIO_VALU(0);
      return false;
    }
    l += lsum;
    if constexpr (DROP) {  // the row sum above is the undropped one; drop before P @ V
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (!(MASKED && !use[b])) drop_weights(sacc[b], mine[b]);
    }
    FA_PRIO_VALU(0);
    FA_PRIO_MFMA(1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (MASKED && !use[b]) continue;
      const vec8 pf0 = pack8<T, 0>(sacc[b]);
      const vec8 pf1 = pack8<T, 1>(sacc[b]);
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        const FA_LDS char* base = vt + b * 32 * C::ROWB;
        vec8 a0 = lds_read_tr_frag<T>(base + v_off[0][db], base + v_off[1][db]);
        oacc[db] = T::mfma(a0, pf0, oacc[db]);
        vec8 a1 = lds_read_tr_frag<T>(base + 16 * C::ROWB + v_off[0][db], base + 16 * C::ROWB + v_off[1][db]);
        oacc[db] = T::mfma(a1, pf1, oacc[db]);
      }
    }
    FA_PRIO_MFMA(0);
    return true;
  };

  using BR = std::integral_constant<int, -1>;

  // ---- main loop: one barrier per tile; every wave runs exactly ntiles iterations ----
  if (Sk % C::BN != 0) {  // a ragged last tile must not expose uninitialised LDS (out-of-range DMA may not write)
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }
  dma_tile(0, 0);
  tile_sync();  // tile 0 and the Q fragments landed
  // Unmasked tiles: one exact tile (the first one, or the one a lazy tile bailed out of -- its prefetch is
  // then already issued), followed by lazy tiles until one bails out.  Separate loops on purpose: bodies that
  // merge control flow get their accumulators copied at every join.
  int t = 0;
  bool prefetched = false;
  while (t < nfull) {
    if (!prefetched && t + 1 < ntiles) dma_tile(t + 1, (t + 1) & 1);
    tile(t, BR{}, std::false_type{});
    tile_sync();
    ++t;
    prefetched = false;
    for (; t < nfull; ++t) {
      if (t + 1 < ntiles) dma_tile(t + 1, (t + 1) & 1);
      if (!tile_lazy(t, std::false_type{})) {
        prefetched = true;
        break;
      }
      tile_sync();
    }
  }
  // masked tiles (causal diagonal, ragged tail): lazy first, the exact path only if the stale max cannot be used
  for (; t < ntiles; ++t) {
    if (!prefetched && t + 1 < ntiles) dma_tile(t + 1, (t + 1) & 1);
    prefetched = false;
    if (!tile_lazy(t, std::true_type{})) tile(t, BR{}, std::true_type{});
    tile_sync();
  }

  // ---- epilogue ----
  const float lt = half_sum(l);
  // lt = 0 only for a variable-length sequence with queries but no keys (S_k = 0: no tile was visited): O = 0, LSE = -inf
  // DROP: O = (1 / (1 - p)) * sum(mask o P) V / l -- the rescale is linear, so it joins the normalisation here
  const float inv = lt > 0.f ? (DROP ? p.drop.rp : 1.0f) / lt : 0.f;
  // all waves are past the last barrier: the K/V buffers are free; wave w stages in its own 32*ROWB bytes
  store_tile_rows<D, T>(oacc, inv, smem + wave * 32 * C::ROWB, ro, qw0 * o_rs, lane, o_rs);
  if (h == 0) buf_store_f32(rl, (qw0 + r) * 4, m * (FOLD ? kLn2 : p.scale) + __builtin_logf(lt));
  }  // pass
}

// ---- host launcher ----------------------------------------------------------
template <int D, typename T, bool CAUSAL, bool DROP = false>
static hipError_t launch(const FwdParams& p, hipStream_t s) {
  using C = FwdCfg<D>;
  const int grid = (CAUSAL && p.pair ? (p.nq_tiles + 1) / 2 : p.nq_tiles) * p.B * p.H;
  auto kern = fa_fwd_kernel<D, T, CAUSAL, DROP>;
  if (C::LDS_BYTES > 48 * 1024) {
    static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
    if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES, opted_in)) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_fwd_v2(FwdParams p, int dtype, int causal, hipStream_t s);  // fa_fwd_v2.hip
hipError_t launch_fwd_v3(FwdParams p, int dtype, int causal, hipStream_t s);  // fa_fwd_v3.hip
hipError_t launch_fwd_v4(FwdParams p, int D, int dtype, int causal, hipStream_t s);  // fa_fwd_v4.hip

hipError_t launch_fwd(FwdParams p, int D, int dtype, int causal, hipStream_t s) {
  const int impl = p.drop.thresh ? 1 : pick_fwd_impl(g_force_fwd, D, dtype, p.B, p.H, p.Sq, p.Sk, causal != 0, p.vl.cu_q == nullptr);
  if (impl == 2) return launch_fwd_v2(p, dtype, causal, s);
  if (impl == 3) return launch_fwd_v3(p, dtype, causal, s);
  if (impl == 4) return launch_fwd_v4(p, D, dtype, causal, s);
  p.nq_tiles = (p.Sq + 127) / 128;
  p.pair = want_pairs(causal != 0, p.nq_tiles, (long)p.B * p.H);
#define FA_GO(DD, TT)                                                                           \
  (p.drop.thresh ? (causal ? launch<DD, TT, true, true>(p, s) : launch<DD, TT, false, true>(p, s)) \
                 : (causal ? launch<DD, TT, true>(p, s) : launch<DD, TT, false>(p, s)))
  if (D == 64) return dtype == 1 ? FA_GO(64, BF16) : FA_GO(64, FP16);
  if (D == 128) return dtype == 1 ? FA_GO(128, BF16) : FA_GO(128, FP16);
#undef FA_GO
  return hipErrorInvalidValue;
}

}  // namespace fa
