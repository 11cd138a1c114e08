// Kernel parameter blocks and host-side launch entry points (internal to libmi355fa.so).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "fa_table.h"

namespace fa {

// Byte strides of one [B, H, S, D] operand whose D dimension is contiguous: batch, head, row.  Inputs and the 16-bit
// outputs (O, dQ, dK, dV) each carry one; LSE / delta rows of one (batch, head) are always contiguous.
struct TensorLayout {
  long long sb, sh;
  int rs;
  bool contiguous(int H, int S, int D) const { return rs == 2 * D && sh == (long long)S * rs && sb == (long long)H * sh; }
};
inline TensorLayout contiguous_layout(int H, int S, int D) {
  return TensorLayout{(long long)H * S * D * 2, (long long)S * D * 2, D * 2};
}

// Division of a work-list index by a launch constant (slices per (batch, head), heads): the persistent kernels decode an
// item per pass, and hipcc's 32-bit division is ~40 instructions that bounce between the scalar and the vector unit
// (stamps: ~1k cycles per decoded item, fa_fwd_v4.hip).  Round-up multiply-shift, exact for 0 <= n < 2^31, 1 <= d < 2^31:
// l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, n / d = (mulhi(m, n) + n) >> l.
struct FastDiv {
  unsigned m;
  int l;
  __device__ __forceinline__ int div(int n) const { return (int)((__umulhi(m, (unsigned)n) + (unsigned)n) >> l); }
};
inline FastDiv make_fastdiv(int d) {
  int l = 0;
  while ((1ll << l) < d) ++l;
  return FastDiv{(unsigned)((((1ull << l) - (unsigned long long)d) << 32) / (unsigned long long)d + 1), l};
}

// Variable-length ("varlen") launches: Q/K/V/O and the gradients are PACKED [total tokens, H, D] tensors, sequence b
// owns rows [cu[b], cu[b+1]) (cu_seqlens_q / cu_seqlens_k, int32, batch + 1 entries, device memory); LSE and delta are
// [H, total_q].  The kernels then take S_q / S_k and every base pointer per (batch, head) from the cu arrays; the grid
// is sized for the longest sequence and workgroups beyond a sequence's own tile count exit at once.  cu_q == nullptr:
// the fixed-length [B, H, S, D] launch of the reference.
struct VarLen {
  const int* cu_q;
  const int* cu_k;
};

// Attention dropout parameters as the kernels see them (fa_common.h `Dropout` has the same members; kept separate so
// that host code does not need the device header).  thresh == 0: no dropout.
struct DropoutParams {
  unsigned thresh, seed_lo, seed_hi, offset;
  float rp;
};

// A sequence's first packed row and its lengths (varlen), or {0, 0, S_q, S_k} for the fixed-length launch.  `b` is
// workgroup-uniform, so these are four scalar loads.
struct SeqInfo {
  int q0, k0, Sq, Sk;
};
__device__ __forceinline__ SeqInfo seq_info(const VarLen& vl, int b, int Sq, int Sk) {
  if (!vl.cu_q) return SeqInfo{0, 0, Sq, Sk};
  const int q0 = vl.cu_q[b], q1 = vl.cu_q[b + 1], k0 = vl.cu_k[b], k1 = vl.cu_k[b + 1];
  return SeqInfo{q0, k0, q1 - q0, k1 - k0};
}

// Work-list slice -> (batch, head).  The XCD-aware work list (fa_common.h xcd_remap) gives every XCD a contiguous run of
// slices.  Fixed-length launches order them (batch, head): all slices cost the same.  Variable-length launches order
// them (head, batch): every XCD then gets a few heads of EVERY sequence -- with (batch, head) order the one long sequence
// of a ragged batch lands on a single XCD (measured: a 8192/4096/.../128 batch 3.9x slower than its FLOPs).
struct BatchHead {
  int b, h;
};
__device__ __forceinline__ BatchHead batch_head(int slice, int B, int H, bool varlen) {
  if (varlen) {
    const int h = slice / B;
    return BatchHead{slice - h * B, h};
  }
  const int b = slice / H;
  return BatchHead{b, slice - b * H};
}

struct FwdParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  int B, H, Sq, Sk;  // varlen: Sq / Sk = the LONGEST sequence (grid sizing); the kernel reads the real ones
  float scale;
  int nq_tiles;  // filled by the launcher
  void* dbg;     // diagnostic builds (-DFA_STAMPS) only: cycle-stamp buffer, else unused
  int pair;      // filled by the launcher: causal workgroups take tile pairs (i, n-1-i)
  FastDiv div_per_bh, div_h;   // filled by the family-4 launcher: work-list index -> slice -> (batch, head)
  TensorLayout lq, lk, lv;  // K and V share their row stride (checked by the C ABI)
  TensorLayout lo;          // output O: contiguous [B, H, S, D] for the reference's launch, strided (fa_fwd_strided), packed rows for varlen
  long long lse_sb, lse_sh; // LSE element strides per batch / head (rows of one (batch, head) are contiguous)
  VarLen vl;
  DropoutParams drop;
  bool all_contiguous(int D) const {
    return !vl.cu_q && lq.contiguous(H, Sq, D) && lk.contiguous(H, Sk, D) && lv.contiguous(H, Sk, D) && lo.contiguous(H, Sq, D);
  }
};

struct BwdParams {
  const void* q;
  const void* k;
  const void* v;
  const void* o;    // dQ kernel only (delta = rowsum(dO * O))
  const void* dout;
  const float* lse;
  float* delta;     // written by the dQ kernel, read by the dK/dV kernel
  void* dq;
  void* dk;
  void* dv;
  int B, H, Sq, Sk;
  float scale;
  int n_tiles;      // filled by the launcher
  void* dbg;        // diagnostic builds (-DFA_STAMPS) only
  int pair;         // filled by the launcher: causal workgroups take tile pairs (i, n-1-i)
  FastDiv div_per_bh, div_h;   // filled by the family-4 launchers: work-list index -> slice -> (batch, head)
  TensorLayout lq, lk, lv, ldo;  // K and V share their row stride (checked by the C ABI)
  TensorLayout lo, ldq, ldk, ldv;  // O (input of the dQ kernel) and the gradient outputs: contiguous, strided, or packed rows (varlen)
  // bf16 scale fold, optional workspace (include/mi355fa.h mi355fa_opts.q_scaled): the dQ kernel writes the Q it actually
  // multiplies -- bf16(Q * softmax_scale * log2e), the operand the forward's LSE was computed from -- to `qs` (layout lqs);
  // the dK/dV launch then gets that buffer as `q` (q_prescaled = 1) and leaves K alone, so its recomputed P is consistent
  // with LSE at any score magnitude.  qs == nullptr / q_prescaled == 0: dK/dV folds the scale into K (its own rounding).
  void* qs;
  TensorLayout lqs;
  int q_prescaled;
  long long lse_sb, lse_sh;        // LSE / delta element strides per batch / head
  VarLen vl;
  DropoutParams drop;
  bool all_contiguous(int D) const {
    // o / dq are unset (zero) in a dK/dV launch and dk / dv in a dQ launch: only the tensors a kernel touches count
    return !vl.cu_q && lq.contiguous(H, Sq, D) && ldo.contiguous(H, Sq, D) && lk.contiguous(H, Sk, D) && lv.contiguous(H, Sk, D) &&
           (!dq || (lo.contiguous(H, Sq, D) && ldq.contiguous(H, Sq, D))) &&
           (!dk || (ldk.contiguous(H, Sk, D) && ldv.contiguous(H, Sk, D)));
  }
};

// Opt a kernel in to a dynamic LDS carve above the 48 KiB default (160 KiB per CU on gfx950).  The attribute belongs to
// the CURRENT device's copy of the function, so it is cached per (kernel template instance, device): `done` is that
// instance's bit mask of devices already set.  A racing second call only repeats an idempotent setting.
inline hipError_t opt_in_lds(const void* kern, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = dev < 64 ? 1ull << dev : 0;   // devices >= 64: never cached, always set
  if (bit && (done.load(std::memory_order_relaxed) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
  return e;
}

// ---- schedule selection: the counterpart of the reference's autotune key (S_q, S_k, D, is_causal), K:18-32 --------
// A table generated offline (tools/tune.py -> fa_table.h) instead of a run-time search.  Schedule families:
//   forward   1 = fa_fwd.hip      128-row workgroups, 32 rows per wave, up to 3 waves per SIMD  (D = 64, 128)
//             2 = fa_fwd_v2.hip   256-row workgroups, 64 rows per wave sharing every K/V fragment (D = 64, fixed length)
//             3 = fa_fwd_v3.hip   128-row workgroups, per-wave three-stage software pipeline (D = 64)
//             4 = fa_fwd_v4.hip   256-row workgroups, ONE wave per SIMD with 64 rows, continuous hand-ordered pipeline,
//                                 row constant m fixed per pass and verified at its end (D = 64, 128; fixed length)
//   dQ        1 = fa_bwd_dq.hip   as forward 1;  2 = fa_bwd_dq_v2.hip as forward 2;
//             3 = fa_bwd_dq_v3.hip  128-row workgroups, per-wave three-stage software pipeline (D = 64)
//             4 = fa_bwd_dq_v4.hip  256-row workgroups, ONE wave per SIMD with 64 rows, every K / V / K^T fragment read
//                                   from LDS once for both row blocks, continuous hand-ordered pipeline (D = 64, fixed length)
//   dK/dV     1 = fa_bwd_dkv.hip  128-key workgroups, 64-row Q/dO tiles;
//             2 = fa_bwd_dkv_v2.hip  128-row Q/dO tiles, hand-ordered pipeline (D = 64, 128)
//             3 = fa_bwd_dkv_v3.hip  256-key workgroups, ONE wave per SIMD with 64 keys, every Q / dO fragment read from
//                                    LDS once for both key groups, continuous hand-ordered pipeline (D = 64)
//             4 = fa_bwd_dkv_v4.hip  family 3's pipeline on the pinned accumulator file: ring of three tiles, counted waits,
//                                    a barrier-free per-wave diagonal phase with the mask in the chain start (D = 64, fixed length)
// The table is keyed on (kernel, D, dtype, causal, B*H bucket, S bucket); a family the launch cannot use (strided
// views for the 64-rows-per-wave kernels, a head dim it does not exist for) falls back to family 1.
// fa_debug_force_impl() (not in the public header) overrides the table for tests, A/B runs and the tuner; 0 = table.
extern std::atomic<int> g_force_fwd, g_force_dq, g_force_dkv;

inline int nearest_log2_bucket(long v, const int* buckets, int n) {
  int best = 0;
  for (int i = 1; i < n; ++i)   // buckets are powers of two: nearest in log2 = compare against the geometric mean
    if ((double)v * v > (double)buckets[i - 1] * buckets[i]) best = i;
  return best;
}
enum { kKernelFwd = 0, kKernelDq = 1, kKernelDkv = 2 };
inline int table_family(int kernel, int D, int dtype, bool causal, long bh, long S) {
  if (D != 64 && D != 128) return 1;
  return table::kFamily[kernel][D == 128][dtype == 1][causal ? 1 : 0][nearest_log2_bucket(bh, table::kBH, table::kNumBH)]
                       [nearest_log2_bucket(S, table::kS, table::kNumS)];
}
// (launches with dropout always take family 1: the dropout variants exist for that family only)
// `fixed_length`: not a variable-length launch (the family-2 forward reads strided views, but not packed batches)
inline int pick_fwd_impl(int forced, int D, int dtype, int B, int H, int Sq, int Sk, bool causal, bool fixed_length) {
  int f = forced ? forced : table_family(kKernelFwd, D, dtype, causal, (long)B * H, Sq > Sk ? Sq : Sk);
  if (f == 2 && (D != 64 || !fixed_length)) f = 1;
  if (f == 3 && D != 64) f = 1;
  // family 4: fixed length; causal launches only when every 256-row query tile has all 256 keys level with it (its causal
  // kernel has no ragged-tile path: fa_fwd_v4.hip)
  if (f == 4 && (!fixed_length || (causal && !(Sk % 256 == 0 && Sk >= (Sq + 255) / 256 * 256)))) f = 1;
  return (f >= 2 && f <= 4) ? f : 1;
}
inline int pick_dq_impl(int forced, int D, int dtype, int B, int H, int Sq, int Sk, bool causal, bool contiguous,
                        bool fixed_length = true) {
  int f = forced ? forced : table_family(kKernelDq, D, dtype, causal, (long)B * H, Sq > Sk ? Sq : Sk);
  // family 4: fixed length, whole 128-key tiles; causal launches only when every 256-row query tile has all 256 keys level
  // with it (its diagonal phase has no ragged path: fa_bwd_dq_v4.hip); otherwise the pipelined family 3
  if (f == 4 && (D != 64 || !fixed_length || Sk % 128 != 0 || (causal && !(Sk % 256 == 0 && Sk >= (Sq + 255) / 256 * 256)))) f = 3;
  if (f == 2 && (D != 64 || !contiguous)) f = 1;
  if (f == 3 && D != 64) f = 1;
  return (f >= 2 && f <= 4) ? f : 1;
}
inline int pick_dkv_impl(int forced, int D, int dtype, int B, int H, int Sq, int Sk, bool causal, bool fixed_length = true) {
  if (D != 64 && D != 128) return 1;
  int f = forced ? forced : table_family(kKernelDkv, D, dtype, causal, (long)B * H, Sq > Sk ? Sq : Sk);
  // family 4: fixed length, whole 128-row query tiles and 256-key tiles; causal launches only when every key tile has its two
  // diagonal query tiles (S_q >= S_k) -- its diagonal phase has no ragged path (fa_bwd_dkv_v4.hip); otherwise family 3
  // (and bf16 only under the causal mask: the fp16 causal instance does not fit the register file)
  if (f == 4 && (D != 64 || !fixed_length || Sq % 128 != 0 || Sk % 256 != 0 || (causal && (Sq < Sk || dtype != 1)))) f = 3;
  if (f == 3) return D == 64 ? 3 : 2;
  return f == 4 ? 4 : (f == 2 ? 2 : 1);
}

// Causal tile pairing equalises the work per workgroup but halves the number of workgroups: worth it as
// long as the paired grid still gives every one of the 256 CUs a workgroup (measured: B4 H8 S2048 -> 256
// pairs: 0.030 ms paired vs 0.038 ms unpaired; S512 -> 64 pairs: 0.014 vs 0.010 ms).
#ifndef FA_NO_PAIRS
inline int want_pairs(bool causal, long tiles, long bh) { return causal && ((tiles + 1) / 2) * bh >= 256; }
#else   // A/B hook: heavy-first single tiles instead of pairs
inline int want_pairs(bool, long, long) { return 0; }
#endif

hipError_t launch_fwd(FwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dq(BwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dkv(BwdParams p, int D, int dtype, int causal, hipStream_t s);

}  // namespace fa
