// Kernel parameter blocks and host-side launch entry points (internal to libmi355fa.so).
#pragma once
#include <hip/hip_runtime.h>

namespace fa {

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
};

hipError_t launch_fwd(FwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dq(BwdParams p, int D, int dtype, int causal, hipStream_t s);
hipError_t launch_bwd_dkv(BwdParams p, int D, int dtype, int causal, hipStream_t s);

}  // namespace fa
