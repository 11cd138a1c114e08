// FlashAttention backward dK / dV, second-generation schedule for gfx950 (head dim 64: two workgroups per CU;
// head dim 128: one, accumulators partly in AGPRs).
//
// Same maths and rounding points as fa_bwd_dkv.hip (reference kernel
// code/_flash_attention_kernel_optimized.py:292-386; runs after the dQ kernel and reads its delta).
// What changes against the first-generation kernel:
//   * Q and dO tiles (128 query rows) go L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), double
//     buffered: the copy of tile t+1 is issued before tile t is computed and retired (vmcnt(0)) after
//     it -- no staging VGPRs and no ds_write pass for the tile data; only the 2 x 128 row
//     constants (-LSE*log2e, -delta) pass through a register to be pre-scaled once per workgroup;
//   * the loop is unrolled by two so that every LDS address is base register + immediate;
//   * causal: a workgroup takes the key-tile PAIR (i, nk-1-i), so all workgroups stream the same
//     number of query tiles;
//   * masked (diagonal) tiles run in their own loop, before the unmasked ones.
#include <stdlib.h>

#include <type_traits>

#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

#ifndef FA_DKV_BQ
#define FA_DKV_BQ 128  // query rows per LDS tile
#endif
#ifndef FA_DKV_STAGGER
#define FA_DKV_STAGGER 2  // block iteration at which waves 2, 3 issue their share of the next tile's DMA (0 = tile start)
#endif

template <int D_>
struct Dkv2Cfg {
  static constexpr int D = D_;
  static constexpr int BK = 128, BQ = FA_DKV_BQ, NT = 256, NW = 4;
  static constexpr int QB = BQ / 32;                       // 32-row query blocks per tile
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  static constexpr int TILE_BYTES = BQ * ROWB;
  static constexpr int DO_BASE = 2 * TILE_BYTES;           // Q[2], then dO[2]
  static constexpr int ROWC_OFF = 4 * TILE_BYTES;          // then row constants: nl[64], nd[64] per buffer
  static constexpr int ROWC_BYTES = 2 * BQ * 4;
  static constexpr int LDS_BYTES = 4 * TILE_BYTES + 2 * ROWC_BYTES;  // 66 KiB at BQ = 128
  static constexpr int DMA_PER_MAT = TILE_BYTES / (NW * 1024);
  static constexpr int RPI = 1024 / ROWB;                  // tile rows per 1-KiB DMA instruction
};

#ifdef FA_STAMPS
#define FA_STAMP(slot)                                                            \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long now_;                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += now_ - last_;                                                    \
    last_ = now_;                                                                 \
  } while (0)
#else
#define FA_STAMP(slot) do {} while (0)
#endif

template <int D, typename T, bool CAUSAL>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void fa_bwd_dkv2_kernel(BwdParams p) {
  using C = Dkv2Cfg<D>;
  using vec8 = typename T::vec8;
#ifdef FA_STAMPS
  unsigned long long clk0_, rt0_;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_), "=s"(rt0_)::"memory");
#endif
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FA_LDS char* smem = (FA_LDS char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // D = 128 launches are persistent (launch2: one workgroup per CU walks items blockIdx.x, blockIdx.x + gridDim.x, ...):
  // 2048 one-per-CU workgroups of ~0.1 ms each otherwise pay their start-up eight times over per CU
  const bool paired = CAUSAL && p.pair;
  const int per_bh = paired ? (p.n_tiles + 1) / 2 : p.n_tiles;
  const int n_items = per_bh * p.B * p.H;
#ifdef FA_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0, nblk_ = 0;
#endif
  // (the D = 64 instances run the body once, visibly to the compiler: a loop they do not need costs them scalar registers)
  constexpr bool PERSIST = D == 128;
  int item = blockIdx.x;   // (launch2: the grid never exceeds the work list)
  do {
  if (PERSIST && item != (int)blockIdx.x) __syncthreads();   // the previous item staged dK / dV in the tile buffers
  const int w = xcd_remap(item, n_items);
  const int bh = w / per_bh;
  const int idx = w - bh * per_bh;
  const BatchHead ix = batch_head(bh, p.B, p.H, p.vl.cu_q != nullptr);
  const int b_ = ix.b, h_ = ix.h;
  // variable-length launch (fa_kernels.h VarLen): this sequence's rows and lengths; surplus items are skipped
  const SeqInfo si = seq_info(p.vl, b_, p.Sq, p.Sk);
  const int Sq = si.Sq, Sk = si.Sk;
  const int nk = (Sk + C::BK - 1) / C::BK;
  if (idx >= (paired ? (nk + 1) / 2 : nk)) continue;
  const int npass = (paired && idx != nk - 1 - idx) ? 2 : 1;

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
  // Row constants of a query tile: the first BQ/64 waves load its LSE rows, the next BQ/64 waves its delta rows,
  // through ONE wave-uniform descriptor and an unconditional load.  (A divergent `if` around the load makes
  // hipcc merge the loaded value with a copy, and the s_waitcnt vmcnt(0) it puts before that copy also waits for
  // the tile DMA issued just above it: the double buffer then hides nothing.)
  const bool rc_lse = wave < C::BQ / 64, rc_any = wave < 2 * C::BQ / 64;
  const __amdgpu_buffer_rsrc_t rrc =
      make_rsrc((rc_lse ? p.lse : p.delta) + rowc_off, rc_any ? (unsigned)Sq * 4 : 0u);
  // row of the tile this thread serves; recomputed where it is used (volatile asm, one VALU op): as a loop
  // invariant hipcc spills it, and a scratch reload is a vmcnt wait just like the one this design avoids
  auto rc_row_now = [&]() __attribute__((always_inline)) -> int {
    int x;
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "n"(C::BQ - 1), "v"(tid));
    return x;
  };

  // loop-invariant per-lane addresses
  int dma_src[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    const int row = (C::BQ / C::NW) * wave + C::RPI * i + lane / C::CPR;
    dma_src[i] = row * q_rs + swz_chunk<D>(row, lane % C::CPR) * 16;
  }
  // the dO tile has the same lane -> (row, chunk) map; only its row stride may differ (the difference can be
  // negative: it is added in the VGPR offset, whose sum row*do_rs + chunk is not; the scalar offset is unsigned)
  const int do_delta = ((C::BQ / C::NW) * wave + lane / C::CPR) * (do_rs - q_rs);
#ifndef FA_DMA_LEGACY
  // dma_pieces (fa_common.h): M0 once per group of up to four 1-KiB pieces; piece j of a group carries the immediate
  // offset 1024*j, which also moves the global address, so it is taken out of the per-lane source offset here
  constexpr int DMA_GRP = C::DMA_PER_MAT < 4 ? C::DMA_PER_MAT : 4;
  int dma_do[C::DMA_PER_MAT];
#pragma unroll
  for (int i = 0; i < C::DMA_PER_MAT; ++i) {
    dma_do[i] = dma_src[i] + do_delta + C::RPI * i * (do_rs - q_rs) - 1024 * (i % DMA_GRP);
    dma_src[i] -= 1024 * (i % DMA_GRP);
  }
#endif
  int row_off[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) row_off[ks] = lds_off<D>(r, 2 * ks + h);
  int tr_off[2][C::DB];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int db = 0; db < C::DB; ++db) tr_off[e][db] = tr_lane_off<D>(lane, 8 * e, db);
  const float c2 = p.scale * kLog2e;
#ifdef FA_DKV_NOFOLD   // A/B hook: exact fma for the exponent argument also at bf16
  constexpr bool FOLD = false;
#else
  constexpr bool FOLD = T::kFoldScale;  // fa_common.h: the score chain starts from -LSE*log2e and K carries c2
#endif
  const int ntiles = (Sq + C::BQ - 1) / C::BQ;

  // A ragged last query tile leaves its tail rows to an out-of-range DMA; make sure those LDS bytes
  // are finite (they are multiplied by P = 0).
  if (Sq % C::BQ != 0) {
    lds_zero_fill(smem, C::LDS_BYTES, C::NT, tid);
    __syncthreads();
  }

  for (int pass = 0; pass < npass; ++pass) {
    const int kt_idx = paired ? (pass == 0 ? idx : nk - 1 - idx) : idx;  // low key tiles are the heavy ones
    const int k0_wg = kt_idx * C::BK;
    const int kw0 = k0_wg + wave * 32;
    if (pass) __syncthreads();  // previous pass staged dK / dV in the tile buffers

    const int t_start = CAUSAL ? k0_wg / C::BQ : 0;
    const int t_full = CAUSAL ? kw0 / C::BQ + 1 : 0;  // tiles >= t_full lie entirely below the diagonal for this wave

    // ---- DMA of one Q/dO tile + the row-constant load (one float per thread < 128) ----
    float rc = 0.f;
    auto fetch_dma = [&](int t, int buf) __attribute__((always_inline)) {
#ifdef FA_ABLATE_DMA
      if (t > t_start + 1) return;  // keep real (random) data in both buffers: zeros would raise the clock
#endif
      const int soff_q = t * C::BQ * q_rs, soff_do = t * C::BQ * do_rs;
#ifndef FA_DMA_LEGACY
#pragma unroll
      for (int g = 0; g < C::DMA_PER_MAT; g += DMA_GRP) {
        const int dst = buf * C::TILE_BYTES + ((C::BQ / C::NW) * wave + C::RPI * g) * C::ROWB;
        dma_pieces<DMA_GRP>(rq, lds_addr_of(smem + dst), dma_src + g, soff_q);
        dma_pieces<DMA_GRP>(rdo, lds_addr_of(smem + C::DO_BASE + dst), dma_do + g, soff_do);
      }
#else
#pragma unroll
      for (int i = 0; i < C::DMA_PER_MAT; ++i) {
        const int dst = buf * C::TILE_BYTES + ((C::BQ / C::NW) * wave + C::RPI * i) * C::ROWB;
        dma16(rq, lds_addr_of(smem + dst), dma_src[i], soff_q);
        dma16(rdo, lds_addr_of(smem + C::DO_BASE + dst), dma_src[i] + do_delta + C::RPI * i * (do_rs - q_rs), soff_do);
      }
#endif
    };
    auto fetch_rc = [&](int t) __attribute__((always_inline)) { rc = buf_load_f32(rrc, (t * C::BQ + rc_row_now()) * 4); };
    auto fetch_tile = [&](int t, int buf) __attribute__((always_inline)) {
      fetch_dma(t, buf);
      fetch_rc(t);
    };
    // everything of the fetched tile has landed (vmcnt(0)): publish the scaled row constants, then meet
    auto commit_tile = [&](int t, int buf, bool fetched) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      FA_STAMP(6);
      if (fetched) {
        FA_LDS float* rcp = (FA_LDS float*)(smem + C::ROWC_OFF + buf * C::ROWC_BYTES);
        // rows past S_q must give P = 0 (K:355-356): exp2(-inf) = 0
        const float lse_c = (t * C::BQ + rc_row_now() < Sq) ? -rc * kLog2e : -INFINITY;
        if (rc_any) rcp[tid] = rc_lse ? lse_c : -rc;  // rcp[row] = -LSE*log2e, rcp[BQ + row] = -delta
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the ds_write above
      FA_STAMP(7);
#ifndef FA_ABLATE_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
    };

    // ---- resident B operands: K^T and V^T of this wave's 32 keys ----
    vec8 kf[C::KS], vf[C::KS];
    if constexpr (D == 128) {
      // D = 128: the 2 x 8 KiB of this wave's K and V rows come through LDS -- LDS-DMA into the wave's own rows of the IDLE
      // tile buffer (Q image: K, dO image: V; 16 coalesced 1-KiB pieces), read back as ds_read_b128 fragments -- instead of
      // 16 per-lane fragment loads that touch 32 cache lines each and stall the wave ~250 cycles apiece wherever they are
      // issued (fa_bwd_dq_v4.hip; profiles/r04_ab_lines.txt).  The first tile's requests follow, so a counted wait
      // covers exactly them; the idle buffer is not written again before the barrier of the first commit.
      const int ib = (t_start & 1) ^ 1;
      int voff[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + lane / C::CPR;   // row of the wave's 32-key block (4 rows per 1-KiB piece)
        voff[i] = (kw0 + row) * kv_rs + swz_chunk<D>(row, lane % C::CPR) * 16 - 1024 * (i & 3);
      }
      const unsigned dstk = lds_addr_of(smem + ib * C::TILE_BYTES + wave * 32 * C::ROWB);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        dma_pieces<4>(rk, dstk + 4096 * g, voff + 4 * g, 0);
        dma_pieces<4>(rv, dstk + C::DO_BASE + 4096 * g, voff + 4 * g, 0);
      }
      const bool first = t_start < ntiles;
      if (first) fetch_tile(t_start, t_start & 1);
      asm volatile("" ::: "memory");
      // the first tile's 2 x DMA_PER_MAT pieces and its row-constant load are younger: leave them in flight
      if (first) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * C::DMA_PER_MAT + 1) & 15) | (((2 * C::DMA_PER_MAT + 1) >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      const FA_LDS char* kb = smem + ib * C::TILE_BYTES + wave * 32 * C::ROWB;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        kf[ks] = as_vec8<T>(lds_read16(kb + row_off[ks]));
        if (FOLD && !p.q_prescaled) kf[ks] = scale_frag<T>(kf[ks], c2);  // K * softmax_scale * log2(e)
        vf[ks] = as_vec8<T>(lds_read16(kb + C::DO_BASE + row_off[ks]));
      }
    } else {
      if (t_start < ntiles) fetch_tile(t_start, t_start & 1);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = (kw0 + r) * kv_rs + (2 * ks + h) * 16;
        kf[ks] = as_vec8<T>(buf_load16(rk, off));
        if (FOLD && !p.q_prescaled) kf[ks] = scale_frag<T>(kf[ks], c2);  // K * softmax_scale * log2(e)
        vf[ks] = as_vec8<T>(buf_load16(rv, off));
      }
    }
    f32x16 dkacc[C::DB], dvacc[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        dkacc[db][i] = 0.f;
        dvacc[db][i] = 0.f;
      }

    // one 32-row query block of the tile in buffer `buf`
    auto q_block = [&](int buf, int b, int qb0, auto masked_tag) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      const FA_LDS char* qbp = smem + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* dbp = smem + C::DO_BASE + buf * C::TILE_BYTES + b * 32 * C::ROWB;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + buf * C::ROWC_BYTES;
      f32x16 nl, pacc, sacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // per-register row constants: reg i <-> row (i&3) + 8(i>>2) + 4h
        const f32x4 a = *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * g + 4 * h) * 4);
        const f32x4 d = *(const FA_LDS f32x4*)(rcp + (C::BQ + 32 * b + 8 * g + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nl[4 * g + j] = a[j];
          sacc[4 * g + j] = FOLD ? a[j] : 0.f;  // FOLD: the chains start from -LSE*log2e and -delta
          pacc[4 * g + j] = d[j];
        }
      }
      if constexpr (D == 128) {   // VGPR-form asm chains started from the row constants (tile_pipelined says why)
        f32x16 sv, pv;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const u32x4 a = lds_read16(qbp + row_off[ks]);
          if (ks == 0) {
            if constexpr (FOLD) T::mfma_v_first(sv, a, __builtin_bit_cast(u32x4, kf[0]), sacc);
            else T::mfma_v_first0(sv, a, __builtin_bit_cast(u32x4, kf[0]));
          } else {
            T::mfma_v_acc(sv, a, __builtin_bit_cast(u32x4, kf[ks]));
          }
          if (FOLD && ks == 1) keep_live(sacc);   // a C operand is read over the MFMA's passes (tools/mfma_lint.py R2)
        }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const u32x4 a = lds_read16(dbp + row_off[ks]);
          if (ks == 0) T::mfma_v_first(pv, a, __builtin_bit_cast(u32x4, vf[0]), pacc);
          else T::mfma_v_acc(pv, a, __builtin_bit_cast(u32x4, vf[ks]));
          if (ks == 1) keep_live(pacc);
        }
        settle_mfma(sv, pv);   // asm MFMA results -> VALU readers (hipcc pads nothing around asm)
        sacc = sv;
        pacc = pv;
      } else {
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
      }
      FA_STAMP(1);  // row-constant + row-fragment reads, S and dP MFMA chains
      // transposed fragments for dV^T / dK^T: issued BEFORE the exp / dS arithmetic (order pinned) so that
      // their LDS latency is covered by it; left alone hipcc reads each one right before its MFMA
      vec8 dof[C::DB][2], qtf[C::DB][2];
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        dof[db][0] = lds_read_tr_frag<T>(dbp + tr_off[0][db], dbp + tr_off[1][db]);
        dof[db][1] = lds_read_tr_frag<T>(dbp + 16 * C::ROWB + tr_off[0][db], dbp + 16 * C::ROWB + tr_off[1][db]);
      }
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        qtf[db][0] = lds_read_tr_frag<T>(qbp + tr_off[0][db], qbp + tr_off[1][db]);
        qtf[db][1] = lds_read_tr_frag<T>(qbp + 16 * C::ROWB + tr_off[0][db], qbp + 16 * C::ROWB + tr_off[1][db]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = FOLD ? sacc[i] : __builtin_fmaf(sacc[i], c2, nl[i]);
        if constexpr (MASKED) {
          const int qrow = qb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          x = (kw0 + r > qrow) ? -INFINITY : x;
        }
        const float pe = __builtin_amdgcn_exp2f(x);
        sacc[i] = pe;            // P
        pacc[i] = pe * pacc[i];  // dS = P o (dP - delta)
      }
      const vec8 p0 = pack8<T, 0>(sacc), p1 = pack8<T, 1>(sacc);
      const vec8 s0 = pack8<T, 0>(pacc), s1 = pack8<T, 1>(pacc);
      __builtin_amdgcn_sched_barrier(0);
      FA_STAMP(2);  // tr reads + exp / dS / pack
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        dvacc[db] = T::mfma(dof[db][0], p0, dvacc[db]);
        dvacc[db] = T::mfma(dof[db][1], p1, dvacc[db]);
      }
#pragma unroll
      for (int db = 0; db < C::DB; ++db) {
        dkacc[db] = T::mfma(qtf[db][0], s0, dkacc[db]);
        dkacc[db] = T::mfma(qtf[db][1], s1, dkacc[db]);
      }
      FA_STAMP(3);  // dV and dK MFMAs
#ifdef FA_STAMPS
      ++nblk_;
#endif
    };

    // ---- unmasked tile, hand-ordered: software pipeline over the tile's query blocks -------------------------
    // A block is 16 MFMA "slots": 0-3 S = Q K^T, 4-7 dP = dO V^T (block b), 8-11 dV^T += dO^T P, 12-15
    // dK^T += Q^T dS (block b-1).  Every slot is one group "MFMA, the LDS read of the operand FOUR slots ahead,
    // a few VALU ops of the other block", closed by sched_barrier(0): the compiler keeps exactly this order (it
    // still allocates registers, counts waits and pads hazards).  Left alone, hipcc reads each operand one or two
    // MFMAs before its use and every wave then sits on the LDS latency sixteen times per block; here the reads
    // are ~140 cycles ahead and the VALU work hides in the issue cycles the MFMAs leave free.
    auto tile_pipelined = [&](auto buf_tag, auto&& block_hook) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_tag)::value;
      constexpr int KS = C::KS, DB = C::DB;
      // slot map of one block: [0, P0) S, [P0, V0) dP (block b); [V0, K0) dV^T, [K0, NS) dK^T (block b-1)
      constexpr int P0 = KS, V0 = 2 * KS, K0 = 2 * KS + 2 * DB, NS = 2 * KS + 4 * DB;
      constexpr int EPS = 16 / V0;          // exps per slot under S / dP            (D=64: 2, D=128: 1)
      constexpr int MPS = 16 / (2 * DB);    // dS multiplies (or fmas) per dV (dK) slot (D=64: 4, D=128: 2)
      // LDS addresses are `per-lane base register (set once per tile, opaque to hipcc) + immediate`.  Left alone, hipcc
      // hoists every (lane offset + constant) pair out of the tile loop and parks the values in accumulator registers; at
      // D = 128, where the tile images span 128 KiB (past the 16-bit immediate of ds_read), that was 721 v_add_u32 and a
      // good part of 768 v_accvgpr_read per 640 MFMAs of the kernel -- 4.2 vector instructions per MFMA, MFMA pipe 49 %
      // busy (profiles/r04_pmc_summary_d128.txt).  (fa_fwd_v4.hip does the same.)
      // (D = 64 keeps plain pointer sums: its images fit the immediate, and two workgroups per CU at 256 registers have
      //  no room for 13 base registers -- they spilled)
      constexpr bool OB = D == 128;
      auto ob = [](int x) __attribute__((always_inline)) { return OB ? opaque(x) : x; };
      const FA_LDS char* qt = smem + BUF * C::TILE_BYTES;
      const FA_LDS char* dt = smem + C::DO_BASE + BUF * C::TILE_BYTES;
      const FA_LDS char* rcp = smem + C::ROWC_OFF + BUF * C::ROWC_BYTES;
      const int lds0 = (int)lds_addr_of(smem);
      int qb_[KS], db_[KS], tq_[2][DB], td_[2][DB];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qb_[ks] = ob(lds0 + row_off[ks] + BUF * C::TILE_BYTES);
        db_[ks] = ob(lds0 + row_off[ks] + C::DO_BASE + BUF * C::TILE_BYTES);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          tq_[e][db] = ob(lds0 + tr_off[e][db] + BUF * C::TILE_BYTES);
          td_[e][db] = ob(lds0 + tr_off[e][db] + C::DO_BASE + BUF * C::TILE_BYTES);
        }
      const int rcb = ob(lds0 + C::ROWC_OFF + BUF * C::ROWC_BYTES + 16 * h);
      // operand fragment of slot s of block b (block index QB = the drain pass: only the slots from V0 on exist)
      auto frag = [&](int b, int s) __attribute__((always_inline)) -> vec8 {
        if constexpr (!OB) {
          if (s < V0) {  // row fragments of block b: Q rows (k-steps 0..KS-1), then dO rows
            const FA_LDS char* base = (s < P0 ? qt : dt) + b * 32 * C::ROWB;
            return as_vec8<T>(lds_read16(base + row_off[s < P0 ? s : s - P0]));
          }
          const int n = s < K0 ? s - V0 : s - K0;
          const FA_LDS char* base = (s < K0 ? dt : qt) + (b - 1) * 32 * C::ROWB + (n & 1) * 16 * C::ROWB;
          return lds_read_tr_frag<T>(base + tr_off[0][n >> 1], base + tr_off[1][n >> 1]);
        }
        if (s < V0)  // row fragments of block b: Q rows (k-steps 0..KS-1), then dO rows
          return as_vec8<T>(lds_read16(lds_at((s < P0 ? qb_[s] : db_[s - P0]) + b * 32 * C::ROWB)));
        // transposed fragments of block b-1: dO^T (d block n>>1, k-step n&1), then Q^T
        const int n = s < K0 ? s - V0 : s - K0;
        const int imm = (b - 1) * 32 * C::ROWB + (n & 1) * 16 * C::ROWB;
        if (s < K0) return lds_read_tr_frag<T>(lds_at(td_[0][n >> 1] + imm), lds_at(td_[1][n >> 1] + imm));
        return lds_read_tr_frag<T>(lds_at(tq_[0][n >> 1] + imm), lds_at(tq_[1][n >> 1] + imm));
      };
      // row constants of block b, group g (registers 4g..4g+3 <-> rows 8g + 4h + 0..3): the accumulators START
      // from them, so the MFMA chains deliver  s*c2 - LSE*log2e  (K is pre-scaled by c2) and  dP - delta
      auto rowc = [&](int b, int g, f32x16& s0, f32x16& p0) __attribute__((always_inline)) {
        const f32x4 a = OB ? *(const FA_LDS f32x4*)lds_at(rcb + (32 * b + 8 * g) * 4) : *(const FA_LDS f32x4*)(rcp + (32 * b + 8 * g + 4 * h) * 4);
        const f32x4 d = OB ? *(const FA_LDS f32x4*)lds_at(rcb + (C::BQ + 32 * b + 8 * g) * 4)
                           : *(const FA_LDS f32x4*)(rcp + (C::BQ + 32 * b + 8 * g + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[4 * g + j] = a[j];
          p0[4 * g + j] = d[j];
        }
      };
      f32x16 xP, dP_;               // previous block: exponent argument -> P, and dP - delta -> dS
      u32x4 pk[2], sk[2];           // previous block: packed P and dS fragments (k-steps 0, 1), built dword by dword
#ifndef FA_DKV_RING
#define FA_DKV_RING 4
#endif
      constexpr int RD = FA_DKV_RING;  // operand ring depth (slots of read-ahead); must divide the slots per block
      static_assert(NS % RD == 0, "ring depth must divide the slot count");
      vec8 fr[RD];                  // operand ring
      f32x16 sacc, pacc;            // this block's accumulators
      // D = 128: the S and dP chains are VGPR-form asm MFMAs (fa_common.h mfma_v_*), started from the row constants as a
      // separate C operand.  hipcc's own MFMAs all accumulate in accumulator registers (one form per kernel), and every
      // score / dP value then costs a v_accvgpr_read before the exp / multiply can touch it: 32 per 32-MFMA block, one
      // more vector instruction per MFMA in a loop that issues 4.2 of them per MFMA (profiles/r04_pmc_summary_d128.txt).
      // dV^T / dK^T stay hipcc's: they are only read in the epilogue.  (D = 64: two workgroups per CU, other budget.)
      constexpr bool VCH = D == 128;
      f32x16 sC, pC;                // VCH: this block's chain starts (-LSE*log2e | unused, -delta)
      f32x16 nl;                    // exact mode (!FOLD): -LSE*log2e of this block, added by an fma under the dK slots
#pragma unroll
      for (int s = 0; s < RD; ++s) fr[s] = frag(0, s);
      if constexpr (FOLD) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if constexpr (VCH) rowc(0, g, sC, pC);
          else rowc(0, g, sacc, pacc);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b <= C::QB; ++b) {
        block_hook(b);
        const bool cur = b < C::QB;   // block b exists: S / dP slots
        const bool prev = b > 0;      // block b-1 exists: dV / dK slots and its VALU work
        f32x16 sn, pn;                // FOLD: next block's starting accumulators (read under the dK slots)
        if constexpr (!FOLD) {
          if (cur) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if constexpr (VCH) rowc(b, g, nl, pC);
              else rowc(b, g, nl, pacc);
            }
            if constexpr (!VCH) {
#pragma unroll
              for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bool active = s < V0 ? cur : prev;
          if (active) {
            const vec8 a = fr[s % RD];
            if (s < V0) {
              if constexpr (VCH) {
                const u32x4 a4 = __builtin_bit_cast(u32x4, a);
                if (s == 0) {
                  if constexpr (FOLD) T::mfma_v_first(sacc, a4, __builtin_bit_cast(u32x4, kf[0]), sC);
                  else T::mfma_v_first0(sacc, a4, __builtin_bit_cast(u32x4, kf[0]));
                } else if (s < P0) {
                  T::mfma_v_acc(sacc, a4, __builtin_bit_cast(u32x4, kf[s < P0 ? s : 0]));
                } else if (s == P0) {
                  T::mfma_v_first(pacc, a4, __builtin_bit_cast(u32x4, vf[0]), pC);
                } else {
                  T::mfma_v_acc(pacc, a4, __builtin_bit_cast(u32x4, vf[s >= P0 ? s - P0 : 0]));
                }
              } else {
                if (s < P0) sacc = T::mfma(a, kf[s < P0 ? s : 0], sacc);
                else pacc = T::mfma(a, vf[s >= P0 ? s - P0 : 0], pacc);
              }
            } else if (s < K0) {
              dvacc[(s - V0) >> 1] = T::mfma(a, as_vec8<T>(pk[(s - V0) & 1]), dvacc[(s - V0) >> 1]);
            } else {
              dkacc[(s - K0) >> 1] = T::mfma(a, as_vec8<T>(sk[(s - K0) & 1]), dkacc[(s - K0) >> 1]);
            }
          }
          // (an MFMA reads its C operand over its passes and hipcc pads that for its own MFMAs only: a chain-start block
          //  that is dead after its use would be reused at once -- live one more slot; tools/mfma_lint.py rule R2)
          if constexpr (VCH) {
            if (FOLD && cur && s == 1) keep_live(sC);
            if (cur && s == P0 + 1) keep_live(pC);
          }
          // operand four slots ahead (wraps into the next block's row fragments; none past the last block)
          {
            const int ns = (s + RD) % NS, nb = b + (s + RD) / NS;
            const bool exists = ns < V0 ? (nb < C::QB) : (nb >= 1 && nb <= C::QB);
            if (exists) fr[s % RD] = frag(nb, ns);
          }
          if constexpr (FOLD) {  // next block's row constants, one group per 2*DB/4 dK slots
            if (s >= K0 && (s - K0) % (2 * DB / 4) == 0 && b + 1 < C::QB) rowc(b + 1, (s - K0) / (2 * DB / 4), sn, pn);
          } else {
            if (cur && s >= K0) {
#pragma unroll
              for (int e = MPS * (s - K0); e < MPS * (s - K0) + MPS; ++e) sacc[e] = __builtin_fmaf(sacc[e], c2, nl[e]);
            }
          }
          // VALU of block b-1, a few ops per slot: exp under the S / dP slots with the pack of each pair one slot
          // after its second exp; dS = P * (dP - delta) under the dV slots with its packs one slot later
          if (prev && s < V0) {
#pragma unroll
            for (int e = EPS * s; e < EPS * s + EPS; ++e) xP[e] = __builtin_amdgcn_exp2f(xP[e]);
          }
          if (prev) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // pair j = registers 2j, 2j+1
              if ((2 * j + 1) / EPS + 1 == s) pk[j >> 2][j & 3] = pack2<T>(xP[2 * j], xP[2 * j + 1]);
            }
          }
          if (prev && s >= V0 && s < K0) {
#pragma unroll
            for (int e = MPS * (s - V0); e < MPS * (s - V0) + MPS; ++e) dP_[e] = xP[e] * dP_[e];
          }
          if (prev) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (V0 + (2 * j + 1) / MPS + 1 == s) sk[j >> 2][j & 3] = pack2<T>(dP_[2 * j], dP_[2 * j + 1]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#ifdef FA_STAMPS
          if (s == P0 - 1) FA_STAMP(1);
          if (s == V0 - 1) FA_STAMP(2);
          if (s == K0 - 1) FA_STAMP(3);
          if (s == NS - 1) { FA_STAMP(5); if (cur) ++nblk_; }
#endif
        }
        if (cur) {
          xP = sacc;
          dP_ = pacc;
        }
        if constexpr (FOLD) {
          if (b + 1 < C::QB) {
            if constexpr (VCH) {
              sC = sn;
              pC = pn;
            } else {
              sacc = sn;
              pacc = pn;
            }
          }
        }
      }
    };

    auto step_full = [&](int t, auto buf_tag) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_tag)::value;
      const bool more = t + 1 < ntiles;
      // Staggered DMA issue: the four waves of a workgroup run in lockstep (one barrier per tile), so issuing all 32
      // LDS-DMA pieces of the next tile at the tile start makes them queue behind one another in the CU's address unit
      // (stamps: ~100 cycles of wave time per 1-KiB piece).  Waves 0 and 1 issue theirs at the tile start, waves 2 and 3
      // two block iterations later (still ~1000 cycles before the tile's closing vmcnt(0)): +1.5-1.7 % (A/B, round 2).
      constexpr int kLateBlock = FA_DKV_STAGGER;
      const bool early = kLateBlock == 0 || wave < 2;
      if (more) {
        fetch_rc(t + 1);
        if (early) fetch_dma(t + 1, BUF ^ 1);
      }
      FA_STAMP(0);  // DMA issue
#ifdef FA_DKV_NO_PIPE
      if (more && !early) fetch_dma(t + 1, BUF ^ 1);
#pragma unroll
      for (int b = 0; b < C::QB; ++b) q_block(BUF, b, 0, std::false_type{});
#else
      tile_pipelined(buf_tag, [&](int b) __attribute__((always_inline)) {
        if (kLateBlock != 0 && b == kLateBlock && more && !early) {
          __builtin_amdgcn_sched_barrier(0);
          fetch_dma(t + 1, BUF ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
#endif
      commit_tile(t + 1, BUF ^ 1, more);
      FA_STAMP(4);  // vmcnt(0) + row constants + barrier
    };
    auto step_masked = [&](int t) __attribute__((always_inline)) {
      const int buf = t & 1;
      const bool more = t + 1 < ntiles;
      if (more) fetch_tile(t + 1, buf ^ 1);
#pragma unroll
      for (int b = 0; b < C::QB; ++b) {
        const int qb0 = t * C::BQ + 32 * b;
        if (qb0 < kw0) continue;  // every row of the block is above the diagonal
        q_block(buf, b, qb0, std::true_type{});
      }
      commit_tile(t + 1, buf ^ 1, more);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

#ifdef FA_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    commit_tile(t_start, t_start & 1, t_start < ntiles);  // first tile landed (and K/V fragments)
#ifdef FA_STAMPS
    seg[4] = seg[6] = seg[7] = 0;
#endif

    int t = t_start;
    const int t_masked_end = min(ntiles, t_full);
    for (; t < t_masked_end; ++t) step_masked(t);
    if (t < ntiles && (t & 1)) {  // align to an even tile so that the unrolled loop sees constant buffers
      step_full(t, B1{});
      ++t;
    }
    for (; t + 2 <= ntiles; t += 2) {
      step_full(t, B0{});
      step_full(t + 1, B1{});
    }
    if (t < ntiles) step_full(t, B0{});

    FA_LDS char* stage = smem + wave * 32 * C::ROWB;
    // dK = dS^T Q * scale; with the pre-scaled Q (= Q * scale * log2e) in LDS that is dS^T Q' * ln 2
    store_tile_rows<D, T>(dkacc, (FOLD && p.q_prescaled) ? kLn2 : p.scale, stage, rdk, kw0 * dk_rs, lane, dk_rs);
    store_tile_rows<D, T>(dvacc, 1.0f, stage, rdv, kw0 * dv_rs, lane, dv_rs);
  }  // pass
  } while (PERSIST && (item += gridDim.x) < n_items);
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

template <int D, typename T, bool CAUSAL>
static hipError_t launch2(const BwdParams& p, hipStream_t s) {
  using C = Dkv2Cfg<D>;
  int grid = (CAUSAL && p.pair ? (p.n_tiles + 1) / 2 : p.n_tiles) * p.B * p.H;
  if (D == 128) {   // one workgroup per CU at D = 128: persistent (the kernel's item loop); a multiple of 8 keeps a workgroup on one XCD's items
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
  auto kern = fa_bwd_dkv2_kernel<D, T, CAUSAL>;
#ifdef FA_STAMPS   // diagnostic builds only: extra LDS per workgroup forces one workgroup per CU (tools/stamps_dkv.py)
  static const int pad = getenv("FA_LDS_PAD") ? atoi(getenv("FA_LDS_PAD")) : 0;
#else
  constexpr int pad = 0;
#endif
  if (C::LDS_BYTES + pad > 48 * 1024) {
    static std::atomic<unsigned long long> opted_in{0};   // per template instance: devices already opted in
    if (hipError_t e = opt_in_lds((const void*)kern, C::LDS_BYTES + pad, opted_in)) return e;
  }
  if (pad) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES + pad, s, p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_bwd_dkv_v2(BwdParams p, int D, int dtype, int causal, hipStream_t s) {
  p.n_tiles = (p.Sk + 127) / 128;  // Dkv2Cfg::BK
  p.pair = want_pairs(causal != 0, p.n_tiles, (long)p.B * p.H);
#define FA_GO(DD, TT) (causal ? launch2<DD, TT, true>(p, s) : launch2<DD, TT, false>(p, s))
  if (D == 64) return dtype == 1 ? FA_GO(64, BF16) : FA_GO(64, FP16);
  if (D == 128) return dtype == 1 ? FA_GO(128, BF16) : FA_GO(128, FP16);
#undef FA_GO
  return hipErrorInvalidValue;
}

}  // namespace fa
