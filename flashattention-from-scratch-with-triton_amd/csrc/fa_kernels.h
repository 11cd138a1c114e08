// Kernel parameter blocks and host-side launch entry points (internal to libmi355fa.so).
#pragma once
#include <hip/hip_runtime.h>

namespace fa {

// Byte strides of one [B, H, S, D] INPUT operand whose D dimension is contiguous: batch, head, row.
// Outputs (O, LSE, dQ, delta, dK, dV) are always contiguous.
struct TensorLayout {
  long long sb, sh;
  int rs;
  bool contiguous(int H, int S, int D) const { return rs == 2 * D && sh == (long long)S * rs && sb == (long long)H * sh; }
};
inline TensorLayout contiguous_layout(int H, int S, int D) {
  return TensorLayout{(long long)H * S * D * 2, (long long)S * D * 2, D * 2};
}

struct FwdParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  int B, H, Sq, Sk;
  float scale;
  int nq_tiles;  // filled by the launcher
  void* dbg;     // diagnostic builds (-DFA_STAMPS) only: cycle-stamp buffer, else unused
  int pair;      // filled by the launcher: causal workgroups take tile pairs (i, n-1-i)
  TensorLayout lq, lk, lv;  // K and V share their row stride (checked by the C ABI)
  bool all_contiguous(int D) const { return lq.contiguous(H, Sq, D) && lk.contiguous(H, Sk, D) && lv.contiguous(H, Sk, D); }
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
  TensorLayout lq, lk, lv, ldo;  // K and V share their row stride (checked by the C ABI); O is contiguous
  bool all_contiguous(int D) const {
    return lq.contiguous(H, Sq, D) && ldo.contiguous(H, Sq, D) && lk.contiguous(H, Sk, D) && lv.contiguous(H, Sk, D);
  }
};

// ---- schedule selection (the counterpart of the reference's autotune key (S_q, S_k, D, is_causal),
// K:18-32): a static rule per kernel instead of a run-time search.  Two schedule families exist for D = 64:
//   1 = 128-row (128-key) workgroups, 32 rows per wave, up to 3 waves per SIMD  (also the only D = 128 path)
//   2 = 256-row workgroups, 64 rows per wave sharing every K/V fragment (forward, dQ: D = 64 only); 128-row Q/dO
//       tiles and the hand-ordered pipeline (dK/dV: D = 64 and 128)
// Measured on MI355X (profiles/r01_schedule_selection.txt): family 1 wins on small grids and on causal
// forward / dQ; family 2 wins on large non-causal grids (fp16 forward, dQ) and for dK/dV from S_q = 256 up.
// fa_debug_force_impl() (not in the public header) overrides the rule for tests and A/B runs; 0 = rule.
extern int g_force_fwd, g_force_dq, g_force_dkv;
// `fold`: the bf16 kernels, whose family-1 hot loops start the score chain from the row constant (fa_common.h
// kFoldScale): family 1 then wins on every grid measured (non-causal B4 H32 S4096: forward 1040 vs 1007 TFLOPS,
// dQ at three workgroups per CU 1161 vs 1135; S8192: forward 1052 vs 1018).
inline int pick_fwd_dq_impl(int forced, int D, int B, int H, int Sq, bool causal, bool fold = false) {
  if (D != 64) return 1;
  if (forced) return forced;
  if (fold) return 1;
  const long tiles256 = (Sq + 255) / 256;
  const long wgs2 = (long)B * H * (causal ? (tiles256 + 1) / 2 : tiles256);
  return (!causal && wgs2 >= 512) ? 2 : 1;
}
// dQ family 3 (fa_bwd_dq_v3.hip, D = 64): the per-wave three-stage pipeline.
inline bool pick_dq3(int forced, int Sk) {
  if (forced) return forced == 3;
  return false;
}
inline int pick_dkv_impl(int forced, int D, int Sq) {
  if (D != 64 && D != 128) return 1;
  if (forced) return forced;
  return Sq >= 256 ? 2 : 1;  // profiles/r01_schedule_selection.txt: the pipelined family 2 wins from S = 256 up
}

// Causal tile pairing equalises the work per workgroup but halves the number of workgroups: worth it as
// long as the paired grid still gives every one of the 256 CUs a workgroup (measured: B4 H8 S2048 -> 256
// pairs: 0.030 ms paired vs 0.038 ms unpaired; S512 -> 64 pairs: 0.014 vs 0.010 ms).
inline int want_pairs(bool causal, long tiles, long bh) { return causal && ((tiles + 1) / 2) * bh >= 256; }

hipError_t launch_fwd(FwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dq(BwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dkv(BwdParams p, int D, int dtype, int causal, hipStream_t s);

}  // namespace fa
