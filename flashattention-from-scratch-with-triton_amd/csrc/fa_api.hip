// C ABI of libmi355fa.so (declared in include/mi355fa.h): argument checks, then enqueue.
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/mi355fa.h"
#include "fa_kernels.h"

namespace fa {
// 0 = selection table of fa_kernels.h.  Written only by fa_debug_force_impl() (tests, A/B tools, the tuner); relaxed
// atomics so that a thread flipping them while another launches is a benign race, not undefined behaviour.
std::atomic<int> g_force_fwd{0}, g_force_dq{0}, g_force_dkv{0};
}

namespace {

thread_local char g_err[256] = "";
void* g_dbg = nullptr;  // set by fa_debug_set_buffer(); read by -DFA_STAMPS builds only

int fail(int code, const char* fmt, const char* what) {
  snprintf(g_err, sizeof(g_err), fmt, what);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

int check_common(const char* fn, int B, int H, int Sq, int Sk, int D, int dtype) {
  if (B < 1 || H < 1 || Sq < 1 || Sk < 1) return fail(MI355FA_ERR_SHAPE, "%s: B, H, S_q, S_k must be >= 1", fn);
  if (D != 64 && D != 128) return fail(MI355FA_ERR_HEAD_DIM, "%s: head dim must be 64 or 128", fn);
  if (dtype != MI355FA_FP16 && dtype != MI355FA_BF16) return fail(MI355FA_ERR_DTYPE, "%s: dtype must be 0 (fp16) or 1 (bf16)", fn);
  // one (batch, head) slice is addressed with 32-bit buffer offsets
  const long long lim = (1ll << 31) - 1;
  if ((long long)Sq * D * 2 > lim || (long long)Sk * D * 2 > lim)
    return fail(MI355FA_ERR_SHAPE, "%s: one (batch, head) slice exceeds 2^31 bytes", fn);
  if ((long long)B * H * ((Sq > Sk ? Sq : Sk) + 127) / 128 > lim)
    return fail(MI355FA_ERR_SHAPE, "%s: too many tiles for one launch", fn);
  return 0;
}

// element strides {batch, head, seq} of a [B, H, S, D] input (NULL = contiguous) -> byte layout
// An OUTPUT (B, extent of the batch dim, given): no broadcast strides -- every (batch, head, row) must be its own memory.
int make_layout(const char* fn, const long long* st, int H, int S, int D, fa::TensorLayout* out, int out_B = 0) {
  if (!st) {
    *out = fa::contiguous_layout(H, S, D);
    return 0;
  }
  if (out_B && ((out_B > 1 && st[0] == 0) || (H > 1 && st[1] == 0)))
    return fail(MI355FA_ERR_STRIDE, "%s: an output cannot have a zero batch / head stride", fn);
  // batch / head strides may be 0 (an expanded K/V shared by several heads, MQA / GQA style); rows must not overlap
  for (int i = 0; i < 3; ++i)
    if (st[i] < 0 || (st[i] & 7) != 0)
      return fail(MI355FA_ERR_STRIDE, "%s: strides must be non-negative multiples of 8 elements", fn);
  if (st[2] < D) return fail(MI355FA_ERR_STRIDE, "%s: the sequence stride must be at least D elements", fn);
  if (((long long)S - 1) * st[2] * 2 + 2ll * D > (1ll << 31) - 1)
    return fail(MI355FA_ERR_STRIDE, "%s: one strided (batch, head) slice exceeds 2^31 bytes", fn);
  *out = fa::TensorLayout{st[0] * 2, st[1] * 2, (int)(st[2] * 2)};
  return 0;
}

// packed [total tokens, H, D] rows (varlen): no batch stride (the cu_seqlens arrays place each sequence), head stride D
fa::TensorLayout packed_layout(int H, int D) { return fa::TensorLayout{0, (long long)D * 2, H * D * 2}; }

int check_varlen(const char* fn, const int* cu_q, const int* cu_k, int batch, int H, int total_q, int total_k, int max_q,
                 int max_k, int D, int dtype) {
  if (!cu_q || !cu_k) return fail(MI355FA_ERR_NULL, "%s: NULL cu_seqlens", fn);
  // max_seqlen is an UPPER bound of the sequence lengths (it sizes the grid; a static bound above the token count is fine)
  if (total_q < 1 || total_k < 1) return fail(MI355FA_ERR_SHAPE, "%s: total tokens must be >= 1", fn);
  if (max_q < 1 || max_k < 1 || batch < 1) return fail(MI355FA_ERR_SHAPE, "%s: batch and max_seqlen must be >= 1", fn);
  // 32-bit buffer offsets are used INSIDE one sequence (rows are H*D*2 bytes apart); the packed tensors may be larger
  if ((long long)max_q * H * D * 2 > (1ll << 31) - 1 || (long long)max_k * H * D * 2 > (1ll << 31) - 1)
    return fail(MI355FA_ERR_SHAPE, "%s: one packed sequence exceeds 2^31 bytes", fn);
  return check_common(fn, batch, H, max_q, max_k, D, dtype);
}

// dropout state of a launch: p is quantised to multiples of 1/256 (include/mi355fa.h); p = 0 -> thresh 0 = plain kernels
int make_dropout(const char* fn, float p_drop, unsigned long long seed, unsigned long long offset, fa::DropoutParams* out) {
  if (!(p_drop >= 0.f) || p_drop >= 1.f) return fail(MI355FA_ERR_SHAPE, "%s: dropout probability must be in [0, 1)", fn);
  unsigned thresh = (unsigned)(p_drop * 256.f + 0.5f);
  if (thresh > 255u) thresh = 255u;
  if (p_drop > 0.f && thresh == 0u)   // never a silent "no dropout"
    return fail(MI355FA_ERR_SHAPE, "%s: dropout probability below 1/512 quantises to 0 (p is kept in 1/256 steps): pass 0 or >= 1/512", fn);
  // the Philox counter has one free 32-bit word for the offset; folding offset[63:32] into the key would make
  // (seed, offset) pairs that differ only there collide
  if (offset >> 32) return fail(MI355FA_ERR_SHAPE, "%s: the dropout offset must be below 2^32", fn);
  out->thresh = thresh;
  out->seed_lo = (unsigned)seed;
  out->seed_hi = (unsigned)(seed >> 32);
  out->offset = (unsigned)offset;
  out->rp = 256.f / (256.f - (float)thresh);
  return 0;
}

}  // namespace

// (fa_debug_poison below)  One wave per SIMD, every register of the wave written, the workgroup's LDS filled.
__global__ __launch_bounds__(256, 1) void fa_poison_kernel() {
  extern __shared__ __attribute__((aligned(16))) char poison_smem[];
  const unsigned nan2 = 0x7FC07FC0u;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) ((unsigned*)poison_smem)[i] = nan2;
#define FA_P4(n) "v_mov_b32 v" #n ", %0\n\tv_accvgpr_write_b32 a" #n ", %0\n\t"
#define FA_P16(a, b, c, d) FA_P4(a) FA_P4(b) FA_P4(c) FA_P4(d)
  // v8 .. v255 and a0 .. a255 (v0 .. v7 stay with the compiler: the loop above and the kernel's few addresses)
  asm volatile(
      FA_P16(8, 9, 10, 11) FA_P16(12, 13, 14, 15) FA_P16(16, 17, 18, 19) FA_P16(20, 21, 22, 23) FA_P16(24, 25, 26, 27) FA_P16(28, 29, 30, 31)
      FA_P16(32, 33, 34, 35) FA_P16(36, 37, 38, 39) FA_P16(40, 41, 42, 43) FA_P16(44, 45, 46, 47) FA_P16(48, 49, 50, 51) FA_P16(52, 53, 54, 55)
      FA_P16(56, 57, 58, 59) FA_P16(60, 61, 62, 63) FA_P16(64, 65, 66, 67) FA_P16(68, 69, 70, 71) FA_P16(72, 73, 74, 75) FA_P16(76, 77, 78, 79)
      FA_P16(80, 81, 82, 83) FA_P16(84, 85, 86, 87) FA_P16(88, 89, 90, 91) FA_P16(92, 93, 94, 95) FA_P16(96, 97, 98, 99) FA_P16(100, 101, 102, 103)
      FA_P16(104, 105, 106, 107) FA_P16(108, 109, 110, 111) FA_P16(112, 113, 114, 115) FA_P16(116, 117, 118, 119) FA_P16(120, 121, 122, 123)
      FA_P16(124, 125, 126, 127) FA_P16(128, 129, 130, 131) FA_P16(132, 133, 134, 135) FA_P16(136, 137, 138, 139) FA_P16(140, 141, 142, 143)
      FA_P16(144, 145, 146, 147) FA_P16(148, 149, 150, 151) FA_P16(152, 153, 154, 155) FA_P16(156, 157, 158, 159) FA_P16(160, 161, 162, 163)
      FA_P16(164, 165, 166, 167) FA_P16(168, 169, 170, 171) FA_P16(172, 173, 174, 175) FA_P16(176, 177, 178, 179) FA_P16(180, 181, 182, 183)
      FA_P16(184, 185, 186, 187) FA_P16(188, 189, 190, 191) FA_P16(192, 193, 194, 195) FA_P16(196, 197, 198, 199) FA_P16(200, 201, 202, 203)
      FA_P16(204, 205, 206, 207) FA_P16(208, 209, 210, 211) FA_P16(212, 213, 214, 215) FA_P16(216, 217, 218, 219) FA_P16(220, 221, 222, 223)
      FA_P16(224, 225, 226, 227) FA_P16(228, 229, 230, 231) FA_P16(232, 233, 234, 235) FA_P16(236, 237, 238, 239) FA_P16(240, 241, 242, 243)
      FA_P16(244, 245, 246, 247) FA_P16(248, 249, 250, 251) FA_P16(252, 253, 254, 255)
      "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\t"
      "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0"
      :: "v"(nan2)
      : "memory"
#define FA_C(n) , "v" #n, "a" #n
#define FA_C8(n) FA_C(n##0) FA_C(n##1) FA_C(n##2) FA_C(n##3) FA_C(n##4) FA_C(n##5) FA_C(n##6) FA_C(n##7) FA_C(n##8) FA_C(n##9)
        , "v8", "v9", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9"
        FA_C8(1) FA_C8(2) FA_C8(3) FA_C8(4) FA_C8(5) FA_C8(6) FA_C8(7) FA_C8(8) FA_C8(9) FA_C8(10) FA_C8(11) FA_C8(12) FA_C8(13) FA_C8(14)
        FA_C8(15) FA_C8(16) FA_C8(17) FA_C8(18) FA_C8(19) FA_C8(20) FA_C8(21) FA_C8(22) FA_C8(23) FA_C8(24)
        FA_C(250) FA_C(251) FA_C(252) FA_C(253) FA_C(254) FA_C(255));
#undef FA_C8
#undef FA_C
#undef FA_P16
#undef FA_P4
  __syncthreads();
}

extern "C" {

int fa_abi_version(void) { return MI355FA_ABI_VERSION; }

// Not part of the public header: pin the schedule family per kernel (0 = automatic rule); tests and A/B tools.
void fa_debug_force_impl(int fwd, int dq, int dkv) {
  fa::g_force_fwd.store(fwd, std::memory_order_relaxed);
  fa::g_force_dq.store(dq, std::memory_order_relaxed);
  fa::g_force_dkv.store(dkv, std::memory_order_relaxed);
}

// Not part of the public header: which schedule family a contiguous launch of this shape takes (kernel 0 = forward,
// 1 = dQ, 2 = dK/dV), after the generated table (fa_table.h), the validity fallbacks and any forced override.
int fa_debug_pick(int kernel, int D, int dtype, int causal, int B, int H, int S_q, int S_k) {
  if (kernel == 0) return fa::pick_fwd_impl(fa::g_force_fwd, D, dtype, B, H, S_q, S_k, causal != 0, true);
  if (kernel == 1) return fa::pick_dq_impl(fa::g_force_dq, D, dtype, B, H, S_q, S_k, causal != 0, true);
  return fa::pick_dkv_impl(fa::g_force_dkv, D, dtype, B, H, S_q, S_k, causal != 0);
}

// Not part of the public header: the work-list division of the persistent kernels (fa_kernels.h FastDiv) evaluated on the
// host exactly as the device evaluates it -- multiplier and shift from make_fastdiv(d), quotient = (mulhi(m, n) + n) >> l --
// so that a CPU test can sweep it against n / d.
int fa_debug_fastdiv(int n, int d) {
  const fa::FastDiv f = fa::make_fastdiv(d);
  const unsigned hi = (unsigned)(((unsigned long long)f.m * (unsigned)n) >> 32);
  return (int)((hi + (unsigned)n) >> f.l);
}

// Not part of the public header: diagnostic hook used by tools/stamps*.py with -DFA_STAMPS builds; in the product library the
// family-4 forward counts, in the buffer's first word, the passes that took their exact second attempt (tests).
void fa_debug_set_buffer(void* p) { g_dbg = p; }

// Not part of the public header: fill every CU's LDS (160 KiB) and every vector / accumulator register a workgroup of the
// family-4 kernels can own with NaN patterns (0x7FC07FC0: a NaN as fp32, as two bf16 and as two fp16).  Tests and
// tools/race_check.py launch it between kernels: a kernel that reads LDS or a register it has not written -- a missing
// wait on an LDS-DMA piece, an accumulator that is not zeroed -- otherwise finds what the PREVIOUS launch left there,
// which in a test that repeats one launch is exactly the right data.
int fa_debug_poison(void* stream) {
  static std::atomic<unsigned long long> opted_in{0};
  if (hipError_t e = fa::opt_in_lds((const void*)fa_poison_kernel, 160 * 1024, opted_in)) return (int)e;
  hipLaunchKernelGGL(fa_poison_kernel, dim3(2048), dim3(256), 160 * 1024, (hipStream_t)stream);
  return (int)hipGetLastError();
}

const char* fa_last_error(void) { return g_err; }

int fa_supported(int D, int dtype) {
  return (D == 64 || D == 128) && (dtype == MI355FA_FP16 || dtype == MI355FA_BF16);
}

// ---- the one implementation: every public entry point below fills an mi355fa_opts and lands here -------------------
// `fn` = the public name, for the error text.  Fixed-length: [B, H, S, D] tensors, optional per-tensor strides.  Varlen
// (opts->cu_seqlens_q != NULL): packed [total, H, D] tensors, B = batch, S_q / S_k = max_seqlen_q / max_seqlen_k.
// Dropout (opts->p_drop > 0) composes with both.
static int read_opts(const char* fn, const mi355fa_opts* in, mi355fa_opts* o) {
  *o = mi355fa_opts{};
  if (!in) return 0;
  if (in->size != sizeof(mi355fa_opts))
    return fail(MI355FA_ERR_SHAPE, "%s: mi355fa_opts.size does not match this library (set it to sizeof(mi355fa_opts))", fn);
  *o = *in;
  if ((o->cu_seqlens_q == nullptr) != (o->cu_seqlens_k == nullptr))
    return fail(MI355FA_ERR_NULL, "%s: cu_seqlens_q and cu_seqlens_k must be given together", fn);
  if (o->cu_seqlens_q && (o->q_strides || o->k_strides || o->v_strides || o->o_strides || o->dout_strides || o->dq_strides ||
                          o->dk_strides || o->dv_strides))
    return fail(MI355FA_ERR_STRIDE, "%s: packed variable-length tensors take no strides", fn);
  return 0;
}

static int fwd_impl(const char* fn, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q,
                    int S_k, int D, int dtype, int causal, float scale, const mi355fa_opts* opts, void* stream) {
  if (!q || !k || !v || !o || !lse) return fail(MI355FA_ERR_NULL, "%s: NULL pointer", fn);
  mi355fa_opts x;
  if (int rc = read_opts(fn, opts, &x)) return rc;
  if (x.cu_seqlens_q) {
    if (int rc = check_varlen(fn, x.cu_seqlens_q, x.cu_seqlens_k, B, H, x.total_q, x.total_k, S_q, S_k, D, dtype)) return rc;
  } else if (int rc = check_common(fn, B, H, S_q, S_k, D, dtype)) {
    return rc;
  }
  if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(o) || misaligned(lse))
    return fail(MI355FA_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
  fa::FwdParams p{q, k, v, o, lse, B, H, S_q, S_k, scale, 0, g_dbg, 0};
  if (x.cu_seqlens_q) {
    p.lq = p.lk = p.lv = p.lo = packed_layout(H, D);
    p.lse_sb = 0;
    p.lse_sh = x.total_q;
    p.vl = fa::VarLen{x.cu_seqlens_q, x.cu_seqlens_k};
  } else {
    if (int rc = make_layout(fn, x.q_strides, H, S_q, D, &p.lq)) return rc;
    if (int rc = make_layout(fn, x.k_strides, H, S_k, D, &p.lk)) return rc;
    if (int rc = make_layout(fn, x.v_strides, H, S_k, D, &p.lv)) return rc;
    if (p.lk.rs != p.lv.rs) return fail(MI355FA_ERR_STRIDE, "%s: K and V must share their sequence stride", fn);
    if (int rc = make_layout(fn, x.o_strides, H, S_q, D, &p.lo, B)) return rc;
    p.lse_sb = (long long)H * S_q;
    p.lse_sh = S_q;
  }
  if (int rc = make_dropout(fn, x.p_drop, x.seed, x.offset, &p.drop)) return rc;
  hipError_t e = fa::launch_fwd(p, D, dtype, causal != 0, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, fn);
  return 0;
}

// the layouts, sequence table and dropout state shared by the two backward launches
static int bwd_fill(const char* fn, fa::BwdParams* p, const mi355fa_opts& x, int B, int H, int S_q, int S_k, int D, int dtype) {
  if (x.cu_seqlens_q) {
    if (int rc = check_varlen(fn, x.cu_seqlens_q, x.cu_seqlens_k, B, H, x.total_q, x.total_k, S_q, S_k, D, dtype)) return rc;
    p->lq = p->lk = p->lv = p->ldo = p->lo = p->ldq = p->ldk = p->ldv = packed_layout(H, D);
    p->lse_sb = 0;
    p->lse_sh = x.total_q;
    p->vl = fa::VarLen{x.cu_seqlens_q, x.cu_seqlens_k};
  } else {
    if (int rc = check_common(fn, B, H, S_q, S_k, D, dtype)) return rc;
    if (int rc = make_layout(fn, x.q_strides, H, S_q, D, &p->lq)) return rc;
    if (int rc = make_layout(fn, x.k_strides, H, S_k, D, &p->lk)) return rc;
    if (int rc = make_layout(fn, x.v_strides, H, S_k, D, &p->lv)) return rc;
    if (int rc = make_layout(fn, x.dout_strides, H, S_q, D, &p->ldo)) return rc;
    if (p->lk.rs != p->lv.rs) return fail(MI355FA_ERR_STRIDE, "%s: K and V must share their sequence stride", fn);
    if (int rc = make_layout(fn, x.o_strides, H, S_q, D, &p->lo)) return rc;
    if (int rc = make_layout(fn, x.dq_strides, H, S_q, D, &p->ldq, B)) return rc;
    if (int rc = make_layout(fn, x.dk_strides, H, S_k, D, &p->ldk, B)) return rc;
    if (int rc = make_layout(fn, x.dv_strides, H, S_k, D, &p->ldv, B)) return rc;
    p->lse_sb = (long long)H * S_q;
    p->lse_sh = S_q;
  }
  return make_dropout(fn, x.p_drop, x.seed, x.offset, &p->drop);
}

static int dq_impl(const char* fn, const void* q, const void* k, const void* v, const void* o, const void* dout,
                   const float* lse, void* dq, float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal,
                   float scale, const mi355fa_opts* opts, void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dq || !delta) return fail(MI355FA_ERR_NULL, "%s: NULL pointer", fn);
  mi355fa_opts x;
  if (int rc = read_opts(fn, opts, &x)) return rc;
  fa::BwdParams p{q, k, v, o, dout, lse, delta, dq, nullptr, nullptr, B, H, S_q, S_k, scale, 0, g_dbg, 0};
  if (int rc = bwd_fill(fn, &p, x, B, H, S_q, S_k, D, dtype)) return rc;
  if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(o) || misaligned(dout) || misaligned(lse) ||
      misaligned(dq) || misaligned(delta) || misaligned(x.q_scaled))
    return fail(MI355FA_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
  if (x.q_scaled && dtype == MI355FA_BF16) {  // workspace the dK/dV launch will read instead of Q (fa_kernels.h BwdParams::qs)
    p.qs = x.q_scaled;
    p.lqs = x.cu_seqlens_q ? packed_layout(H, D) : fa::contiguous_layout(H, S_q, D);
  }
  hipError_t e = fa::launch_bwd_dq(p, D, dtype, causal != 0, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, fn);
  return 0;
}

static int dkv_impl(const char* fn, const void* q, const void* k, const void* v, const void* dout, const float* lse,
                    const float* delta, void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal,
                    float scale, const mi355fa_opts* opts, void* stream) {
  if (!q || !k || !v || !dout || !lse || !delta || !dk || !dv) return fail(MI355FA_ERR_NULL, "%s: NULL pointer", fn);
  mi355fa_opts x;
  if (int rc = read_opts(fn, opts, &x)) return rc;
  fa::BwdParams p{q, k, v, nullptr, dout, lse, const_cast<float*>(delta), nullptr, dk, dv, B, H, S_q, S_k, scale, 0, g_dbg, 0};
  if (int rc = bwd_fill(fn, &p, x, B, H, S_q, S_k, D, dtype)) return rc;
  if (misaligned(q) || misaligned(k) || misaligned(v) || misaligned(dout) || misaligned(lse) || misaligned(delta) ||
      misaligned(dk) || misaligned(dv) || misaligned(x.q_scaled))
    return fail(MI355FA_ERR_ALIGN, "%s: pointers must be 16-byte aligned", fn);
  if (x.q_scaled && dtype == MI355FA_BF16) {  // the rows the dQ launch wrote: Q * scale * log2e as the forward rounded it
    p.q = x.q_scaled;
    p.lq = x.cu_seqlens_q ? packed_layout(H, D) : fa::contiguous_layout(H, S_q, D);
    p.q_prescaled = 1;
  }
  hipError_t e = fa::launch_bwd_dkv(p, D, dtype, causal != 0, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, fn);
  return 0;
}

static mi355fa_opts base_opts() {
  mi355fa_opts x{};
  x.size = sizeof(mi355fa_opts);
  return x;
}

// ---- general, composable entry points ------------------------------------------------------------------------------
int fa_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q, int S_k, int D,
              int dtype, int causal, float scale, const mi355fa_opts* opts, void* stream) {
  return fwd_impl("fa_fwd_ex", q, k, v, o, lse, B, H, S_q, S_k, D, dtype, causal, scale, opts, stream);
}
int fa_bwd_dq_ex(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                 float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                 const mi355fa_opts* opts, void* stream) {
  return dq_impl("fa_bwd_dq_ex", q, k, v, o, dout, lse, dq, delta, B, H, S_q, S_k, D, dtype, causal, scale, opts, stream);
}
int fa_bwd_dkv_ex(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                  void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                  const mi355fa_opts* opts, void* stream) {
  return dkv_impl("fa_bwd_dkv_ex", q, k, v, dout, lse, delta, dk, dv, B, H, S_q, S_k, D, dtype, causal, scale, opts, stream);
}

// ---- the reference's three launches (contiguous [B, H, S, D]) ------------------------------------------------------
int fa_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q, int S_k, int D,
           int dtype, int causal, float scale, void* stream) {
  return fwd_impl("fa_fwd", q, k, v, o, lse, B, H, S_q, S_k, D, dtype, causal, scale, nullptr, stream);
}
int fa_bwd_dq(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
              float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, void* stream) {
  return dq_impl("fa_bwd_dq", q, k, v, o, dout, lse, dq, delta, B, H, S_q, S_k, D, dtype, causal, scale, nullptr, stream);
}
int fa_bwd_dkv(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
               void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
               void* stream) {
  return dkv_impl("fa_bwd_dkv", q, k, v, dout, lse, delta, dk, dv, B, H, S_q, S_k, D, dtype, causal, scale, nullptr, stream);
}

// ---- strided tensors ------------------------------------------------------------------------------------------------
int fa_fwd_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides, const void* v,
                   const long long* v_strides, void* o, const long long* o_strides, float* lse, int B, int H, int S_q,
                   int S_k, int D, int dtype, int causal, float scale, void* stream) {
  mi355fa_opts x = base_opts();
  x.q_strides = q_strides;
  x.k_strides = k_strides;
  x.v_strides = v_strides;
  x.o_strides = o_strides;
  return fwd_impl("fa_fwd", q, k, v, o, lse, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}
int fa_bwd_dq_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides,
                      const void* v, const long long* v_strides, const void* o, const long long* o_strides,
                      const void* dout, const long long* dout_strides, const float* lse, void* dq,
                      const long long* dq_strides, float* delta, int B, int H, int S_q, int S_k, int D, int dtype,
                      int causal, float scale, void* stream) {
  mi355fa_opts x = base_opts();
  x.q_strides = q_strides;
  x.k_strides = k_strides;
  x.v_strides = v_strides;
  x.o_strides = o_strides;
  x.dout_strides = dout_strides;
  x.dq_strides = dq_strides;
  return dq_impl("fa_bwd_dq", q, k, v, o, dout, lse, dq, delta, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}
int fa_bwd_dkv_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides,
                       const void* v, const long long* v_strides, const void* dout, const long long* dout_strides,
                       const float* lse, const float* delta, void* dk, const long long* dk_strides, void* dv,
                       const long long* dv_strides, int B, int H, int S_q, int S_k, int D, int dtype, int causal,
                       float scale, void* stream) {
  mi355fa_opts x = base_opts();
  x.q_strides = q_strides;
  x.k_strides = k_strides;
  x.v_strides = v_strides;
  x.dout_strides = dout_strides;
  x.dk_strides = dk_strides;
  x.dv_strides = dv_strides;
  return dkv_impl("fa_bwd_dkv", q, k, v, dout, lse, delta, dk, dv, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}

// ---- variable-length ("varlen"): packed [total, H, D] tensors + cu_seqlens (include/mi355fa.h) ------------------------
static mi355fa_opts varlen_opts(const int* cu_q, const int* cu_k, int total_q, int total_k) {
  mi355fa_opts x = base_opts();
  x.cu_seqlens_q = cu_q;
  x.cu_seqlens_k = cu_k;
  x.total_q = total_q;
  x.total_k = total_k;
  return x;
}
int fa_fwd_varlen(const void* q, const void* k, const void* v, void* o, float* lse, const int* cu_seqlens_q,
                  const int* cu_seqlens_k, int batch, int H, int total_q, int total_k, int max_seqlen_q, int max_seqlen_k,
                  int D, int dtype, int causal, float scale, void* stream) {
  if (!cu_seqlens_q || !cu_seqlens_k) return fail(MI355FA_ERR_NULL, "%s: NULL cu_seqlens", "fa_fwd_varlen");
  const mi355fa_opts x = varlen_opts(cu_seqlens_q, cu_seqlens_k, total_q, total_k);
  return fwd_impl("fa_fwd_varlen", q, k, v, o, lse, batch, H, max_seqlen_q, max_seqlen_k, D, dtype, causal, scale, &x, stream);
}
int fa_bwd_dq_varlen(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                     float* delta, const int* cu_seqlens_q, const int* cu_seqlens_k, int batch, int H, int total_q,
                     int total_k, int max_seqlen_q, int max_seqlen_k, int D, int dtype, int causal, float scale,
                     void* stream) {
  if (!cu_seqlens_q || !cu_seqlens_k) return fail(MI355FA_ERR_NULL, "%s: NULL cu_seqlens", "fa_bwd_dq_varlen");
  const mi355fa_opts x = varlen_opts(cu_seqlens_q, cu_seqlens_k, total_q, total_k);
  return dq_impl("fa_bwd_dq_varlen", q, k, v, o, dout, lse, dq, delta, batch, H, max_seqlen_q, max_seqlen_k, D, dtype, causal,
                 scale, &x, stream);
}
int fa_bwd_dkv_varlen(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                      void* dk, void* dv, const int* cu_seqlens_q, const int* cu_seqlens_k, int batch, int H, int total_q,
                      int total_k, int max_seqlen_q, int max_seqlen_k, int D, int dtype, int causal, float scale,
                      void* stream) {
  if (!cu_seqlens_q || !cu_seqlens_k) return fail(MI355FA_ERR_NULL, "%s: NULL cu_seqlens", "fa_bwd_dkv_varlen");
  const mi355fa_opts x = varlen_opts(cu_seqlens_q, cu_seqlens_k, total_q, total_k);
  return dkv_impl("fa_bwd_dkv_varlen", q, k, v, dout, lse, delta, dk, dv, batch, H, max_seqlen_q, max_seqlen_k, D, dtype,
                  causal, scale, &x, stream);
}

// ---- attention dropout (include/mi355fa.h): contiguous [B, H, S, D] tensors as fa_fwd / fa_bwd_* ----------------------
float fa_dropout_keep_scale(float p_drop) {   // 1 / (1 - p) for the quantised p the kernels use
  fa::DropoutParams d;
  if (make_dropout("fa_dropout_keep_scale", p_drop, 0, 0, &d)) return 0.f;
  return d.rp;
}
static mi355fa_opts dropout_opts(float p_drop, unsigned long long seed, unsigned long long offset) {
  mi355fa_opts x = base_opts();
  x.p_drop = p_drop;
  x.seed = seed;
  x.offset = offset;
  return x;
}
int fa_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q, int S_k, int D,
                   int dtype, int causal, float scale, float p_drop, unsigned long long seed, unsigned long long offset,
                   void* stream) {
  const mi355fa_opts x = dropout_opts(p_drop, seed, offset);
  return fwd_impl("fa_fwd_dropout", q, k, v, o, lse, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}
int fa_bwd_dq_dropout(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                      float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, float p_drop,
                      unsigned long long seed, unsigned long long offset, void* stream) {
  const mi355fa_opts x = dropout_opts(p_drop, seed, offset);
  return dq_impl("fa_bwd_dq_dropout", q, k, v, o, dout, lse, dq, delta, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}
int fa_bwd_dkv_dropout(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                       void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                       float p_drop, unsigned long long seed, unsigned long long offset, void* stream) {
  const mi355fa_opts x = dropout_opts(p_drop, seed, offset);
  return dkv_impl("fa_bwd_dkv_dropout", q, k, v, dout, lse, delta, dk, dv, B, H, S_q, S_k, D, dtype, causal, scale, &x, stream);
}

}  // extern "C"
