// Host-side binding of libmi355fa.so for PyTorch-ROCm: the launchers and the autograd function of
// My_FlashAttention_optimized.py (reference: code/My_FlashAttention_optimized.py:14-170) written against the C ABI of
// include/mi355fa.h.  PyTorch supplies device memory, the current stream and the autograd graph -- nothing else.
//
// Why this exists beside the ctypes binding (_mi355fa.py, still used by tools and by the tests that drive the C ABI
// directly): at the reference's small benchmark shapes (B=4, H=8, S=512) a fwd+bwd step is ~35 us of kernels, and a
// Python autograd.Function costs ~125 us of host time per step (Function.apply, ctx bookkeeping, the backward running
// on the autograd thread under the GIL, ctypes argument marshalling) against ~80 us for torch's own C++ SDPA.  Here the
// same sequence -- checks, torch::empty outputs, fa_fwd_strided / fa_bwd_dq_strided / fa_bwd_dkv_strided on the current
// stream -- runs without the interpreter.
//
// Built by csrc/Makefile with g++ (host code only; no device code here) into _mi355fa_torch.so next to libmi355fa.so.
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include <tuple>

#include "../../include/mi355fa.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

// The reference raises AssertionError on bad arguments (M:133-136); keep the exception type.
[[noreturn]] void assertion(const char* msg) {
  PyErr_SetString(PyExc_AssertionError, msg);
  throw pybind11::error_already_set();
}
#define FA_ASSERT(cond, msg) \
  do {                       \
    if (!(cond)) assertion(msg); \
  } while (0)

void check_rc(int rc, const char* what) {
  if (rc != 0) {
    std::string m = std::string(what) + " failed (code " + std::to_string(rc) + "): " + fa_last_error();
    throw std::runtime_error(m);
  }
}

// _mi355fa.strided_ok: can the kernels read `t` in place?
bool strided_ok(const Tensor& t) {
  if (reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) return false;
  if (t.is_contiguous()) return true;
  if (t.stride(3) != 1) return false;
  if (t.stride(2) < t.size(3)) return false;
  if (t.size(2) > 1 && t.stride(2) % 8) return false;
  if ((t.size(2) - 1) * t.stride(2) * 2 + 2 * t.size(3) > ((1ll << 31) - 1)) return false;
  for (int i = 0; i < 2; ++i)
    if (t.size(i) != 1 && (t.stride(i) < 0 || t.stride(i) % 8 != 0)) return false;
  return true;
}
Tensor in_place(const Tensor& t) { return strided_ok(t) ? t : t.clone(at::MemoryFormat::Contiguous); }

// element strides {batch, head, seq} for the C ABI, or nullptr for a contiguous tensor (_mi355fa.strides3)
struct Strides3 {
  long long v[3];
  const long long* ptr;
  explicit Strides3(const Tensor& t) {
    if (t.is_contiguous()) {
      ptr = nullptr;
      return;
    }
    v[0] = t.size(0) > 1 ? t.stride(0) : 0;
    v[1] = t.size(1) > 1 ? t.stride(1) : 0;
    v[2] = t.size(2) > 1 ? t.stride(2) : t.size(3);
    ptr = v;
  }
};

// Output for an input the kernels read in place: the input's own memory order (what empty_like gives a dense view --
// the reference allocates with empty_like too, M:24,71-73, but only after its .contiguous() copies), so a model that keeps
// [B, S, H, D] activations gets O and the gradients back in that order and never transposes.  Layouts the kernels cannot
// write (a broadcast stride, a head dim that does not stay innermost) get the contiguous tensor.
Tensor out_like(const Tensor& t) {
  if (!t.is_contiguous()) {
    Tensor o = at::empty_like(t);
    if (o.stride(3) == 1 && o.is_non_overlapping_and_dense() && strided_ok(o)) return o;
  }
  return torch::empty(t.sizes(), t.options());
}

int dtype_code(const Tensor& t) {
  if (t.scalar_type() == at::kHalf) return MI355FA_FP16;
  if (t.scalar_type() == at::kBFloat16) return MI355FA_BF16;
  assertion("dtype must be float16 or bfloat16");
}

void check_qkv(const Tensor& Q, const Tensor& K, const Tensor& V) {
  FA_ASSERT(Q.dim() == 4 && K.dim() == 4 && V.dim() == 4, "Q, K, V must be [B, H, S, D]");
  FA_ASSERT(K.size(0) == Q.size(0) && K.size(1) == Q.size(1),
            "K must have Q's batch and head counts (expand shared K/V heads)");
  FA_ASSERT(V.sizes() == K.sizes(), "K and V must have the same shape");
  FA_ASSERT(Q.size(3) == K.size(3), "Q, K, V must share the head dim");
  FA_ASSERT(Q.device() == K.device() && Q.device() == V.device(), "Q, K, V must be on the same device");
  FA_ASSERT(Q.scalar_type() == K.scalar_type() && Q.scalar_type() == V.scalar_type(), "Q, K, V must share their dtype");
}

void* current_stream(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

// mi355fa_opts for the general entry points (fa_*_ex): zeroed, sized, with the caller's dropout triple
mi355fa_opts make_opts(double p_drop, int64_t seed, int64_t offset) {
  FA_ASSERT(p_drop >= 0.0 && p_drop < 1.0, "dropout_p must be in [0, 1)");
  mi355fa_opts x{};
  x.size = sizeof(mi355fa_opts);
  x.p_drop = (float)p_drop;
  x.seed = (unsigned long long)seed;
  x.offset = (unsigned long long)offset;
  return x;
}

// flash_attention_forward (M:14-60): allocate O / LSE, enqueue.  Inputs: contiguous or strided_ok views, K and V sharing
// their sequence stride.  dropout_p > 0: attention dropout with the Philox mask of (seed, offset) (include/mi355fa.h).
std::tuple<Tensor, Tensor> forward_launch(const Tensor& Q, const Tensor& K, const Tensor& V, bool causal, double p_drop,
                                          int64_t seed, int64_t offset) {
  check_qkv(Q, K, V);
  FA_ASSERT(Q.is_cuda(), "Q, K, V must be device tensors");
  const int64_t B = Q.size(0), H = Q.size(1), Sq = Q.size(2), D = Q.size(3), Sk = K.size(2);
  const int dt = dtype_code(Q);
  c10::OptionalDeviceGuard guard(Q.device());
  Tensor O = out_like(Q);
  Tensor LSE = torch::empty({B, H, Sq}, Q.options().dtype(at::kFloat));
  Strides3 sq(Q), sk(K), sv(V), so(O);
  mi355fa_opts x = make_opts(p_drop, seed, offset);
  x.q_strides = sq.ptr;
  x.k_strides = sk.ptr;
  x.v_strides = sv.ptr;
  x.o_strides = so.ptr;
  check_rc(fa_fwd_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), (float*)LSE.data_ptr(), (int)B, (int)H, (int)Sq,
                     (int)Sk, (int)D, dt, causal ? 1 : 0, (float)(1.0 / std::sqrt((double)D)), &x, current_stream(Q)),
           "fa_fwd");
  return {O, LSE};
}

// flash_attention_backward (M:62-128): allocate dQ / dK / dV / delta, enqueue dQ (+delta) then dK/dV on the same stream
// (the dK/dV kernel reads the delta the dQ kernel wrote, K:376).  Dropout: the triple the forward was given.
std::tuple<Tensor, Tensor, Tensor> backward_launch(const Tensor& Q, const Tensor& K, const Tensor& V, const Tensor& O_,
                                                   const Tensor& dO, const Tensor& LSE, bool causal, double p_drop,
                                                   int64_t seed, int64_t offset) {
  check_qkv(Q, K, V);
  FA_ASSERT(Q.is_cuda(), "Q, K, V must be device tensors");
  FA_ASSERT(O_.sizes() == Q.sizes() && dO.sizes() == Q.sizes(), "O and dO must have Q's shape");
  FA_ASSERT(LSE.dim() == 3 && LSE.size(0) == Q.size(0) && LSE.size(1) == Q.size(1) && LSE.size(2) == Q.size(2),
            "LSE must be [B, H, S_q]");
  FA_ASSERT(O_.device() == Q.device() && dO.device() == Q.device() && LSE.device() == Q.device(),
            "O, dO, LSE must be on Q's device");
  FA_ASSERT(LSE.scalar_type() == at::kFloat && LSE.is_contiguous(), "LSE must be contiguous float32");
  const int64_t B = Q.size(0), H = Q.size(1), Sq = Q.size(2), D = Q.size(3), Sk = K.size(2);
  const int dt = dtype_code(Q);
  c10::OptionalDeviceGuard guard(Q.device());
  Tensor O = in_place(O_);
  Tensor dQ, dK, dV;
  if (!Q.is_contiguous() || !K.is_contiguous() || !V.is_contiguous()) {  // each gradient in its input's memory order
    dQ = out_like(Q);
    dK = out_like(K);
    dV = out_like(V);
  } else if (Sq == Sk) {  // self-attention: one allocation for the three gradients (M:71-73 makes three)
    Tensor g = torch::empty({3, B, H, Sq, D}, Q.options());
    dQ = g.select(0, 0);
    dK = g.select(0, 1);
    dV = g.select(0, 2);
  } else {
    dQ = torch::empty({B, H, Sq, D}, Q.options());
    Tensor g = torch::empty({2, B, H, Sk, D}, Q.options());
    dK = g.select(0, 0);
    dV = g.select(0, 1);
  }
  // ONE scratch allocation: delta [B, H, S_q] fp32 and, bf16, behind it (256-byte aligned) the Q rows the dQ launch
  // multiplied, left for the dK/dV launch (mi355fa_opts.q_scaled) -- at the small end of the reference's grid a step is
  // host-bound and every allocation is ~1 us of it (profiles/r03_small_trace.txt)
  const int64_t delta_bytes = B * H * Sq * 4, qs_off = (delta_bytes + 255) & ~(int64_t)255;
  Tensor scratch = torch::empty({qs_off + (dt == MI355FA_BF16 ? B * H * Sq * D * 2 : 0)}, Q.options().dtype(at::kByte));
  float* delta = (float*)scratch.data_ptr();
  Strides3 sq(Q), sk(K), sv(V), so(O), sdo(dO), sdq(dQ), sdk(dK), sdv(dV);
  mi355fa_opts x = make_opts(p_drop, seed, offset);
  if (dt == MI355FA_BF16) x.q_scaled = (char*)scratch.data_ptr() + qs_off;
  x.q_strides = sq.ptr;
  x.k_strides = sk.ptr;
  x.v_strides = sv.ptr;
  x.o_strides = so.ptr;
  x.dout_strides = sdo.ptr;
  x.dq_strides = sdq.ptr;
  x.dk_strides = sdk.ptr;
  x.dv_strides = sdv.ptr;
  void* st = current_stream(Q);
  const float scale = (float)(1.0 / std::sqrt((double)D));
  check_rc(fa_bwd_dq_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), (const float*)LSE.data_ptr(),
                        dQ.data_ptr(), delta, (int)B, (int)H, (int)Sq, (int)Sk, (int)D, dt, causal ? 1 : 0,
                        scale, &x, st),
           "fa_bwd_dq");
  check_rc(fa_bwd_dkv_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dO.data_ptr(), (const float*)LSE.data_ptr(),
                         (const float*)delta, dK.data_ptr(), dV.data_ptr(), (int)B, (int)H, (int)Sq, (int)Sk, (int)D,
                         dt, causal ? 1 : 0, scale, &x, st),
           "fa_bwd_dkv");
  return {dQ, dK, dV};
}

// FlashAttentionFunction (M:130-166)
class FlashAttnFn : public torch::autograd::Function<FlashAttnFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& Q, const Tensor& K, const Tensor& V, bool is_causal) {
    FA_ASSERT(Q.is_cuda() && K.is_cuda() && V.is_cuda(), "Q, K, V must be device tensors");
    FA_ASSERT(Q.scalar_type() == at::kHalf || Q.scalar_type() == at::kBFloat16, "dtype must be float16 or bfloat16");
    check_qkv(Q, K, V);
    FA_ASSERT(Q.size(3) == 64 || Q.size(3) == 128, "head dim must be 64 or 128");
    // no copy for views the kernels can read in place (M:138-140 copies every non-contiguous input)
    Tensor Q_ = in_place(Q), K_ = in_place(K), V_ = in_place(V);
    if (K_.size(2) > 1 && K_.stride(2) != V_.stride(2)) {  // the kernels use one row stride for the K/V pair
      K_ = K_.contiguous();
      V_ = V_.contiguous();
    }
    auto out = forward_launch(Q_, K_, V_, is_causal, 0.0, 0, 0);
    ctx->save_for_backward({Q_, K_, V_, std::get<0>(out), std::get<1>(out)});
    ctx->saved_data["is_causal"] = is_causal;
    return std::get<0>(out);
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto s = ctx->get_saved_variables();
    const bool causal = ctx->saved_data["is_causal"].toBool();
    Tensor dO = in_place(grads[0]);
    auto g = backward_launch(s[0], s[1], s[2], s[3], dO, s[4], causal, 0.0, 0, 0);
    return {std::get<0>(g), std::get<1>(g), std::get<2>(g), Tensor()};
  }
};

Tensor flash_attention(const Tensor& Q, const Tensor& K, const Tensor& V, bool is_causal) {
  return FlashAttnFn::apply(Q, K, V, is_causal);
}

// ---- variable-length sequences: packed [total, H, D] tensors + cu_seqlens (include/mi355fa.h, fa_*_varlen) ---------
void check_varlen(const Tensor& Q, const Tensor& K, const Tensor& V, const Tensor& cu_q, const Tensor& cu_k) {
  FA_ASSERT(Q.dim() == 3 && K.dim() == 3 && V.dim() == 3, "varlen Q, K, V must be packed [total tokens, H, D]");
  FA_ASSERT(Q.is_cuda() && K.is_cuda() && V.is_cuda() && cu_q.is_cuda() && cu_k.is_cuda(), "varlen tensors must be device tensors");
  FA_ASSERT(K.sizes() == V.sizes(), "K and V must have the same shape");
  FA_ASSERT(Q.size(1) == K.size(1) && Q.size(2) == K.size(2), "Q, K, V must share heads and head dim");
  FA_ASSERT(Q.scalar_type() == K.scalar_type() && Q.scalar_type() == V.scalar_type(), "Q, K, V must share their dtype");
  FA_ASSERT(Q.device() == K.device() && Q.device() == V.device() && Q.device() == cu_q.device() && Q.device() == cu_k.device(),
            "all varlen tensors must be on the same device");
  FA_ASSERT(cu_q.scalar_type() == at::kInt && cu_k.scalar_type() == at::kInt && cu_q.dim() == 1 && cu_k.dim() == 1 &&
                cu_q.is_contiguous() && cu_k.is_contiguous() && cu_q.numel() == cu_k.numel() && cu_q.numel() >= 2,
            "cu_seqlens_q / cu_seqlens_k must be contiguous int32 vectors of batch + 1 entries");
  FA_ASSERT(Q.size(2) == 64 || Q.size(2) == 128, "head dim must be 64 or 128");
}
Tensor packed(const Tensor& t) {  // the varlen kernels read packed rows only: copy anything else
  return (t.is_contiguous() && reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0) ? t : t.clone(at::MemoryFormat::Contiguous);
}

std::tuple<Tensor, Tensor> varlen_forward_launch(const Tensor& Q_, const Tensor& K_, const Tensor& V_, const Tensor& cu_q,
                                                 const Tensor& cu_k, int64_t max_q, int64_t max_k, bool causal, double p_drop,
                                                 int64_t seed, int64_t offset) {
  check_varlen(Q_, K_, V_, cu_q, cu_k);
  Tensor Q = packed(Q_), K = packed(K_), V = packed(V_);
  const int64_t Tq = Q.size(0), Tk = K.size(0), H = Q.size(1), D = Q.size(2), B = cu_q.numel() - 1;
  c10::OptionalDeviceGuard guard(Q.device());
  Tensor O = torch::empty({Tq, H, D}, Q.options());
  Tensor LSE = torch::empty({H, Tq}, Q.options().dtype(at::kFloat));
  mi355fa_opts x = make_opts(p_drop, seed, offset);
  x.cu_seqlens_q = (const int*)cu_q.data_ptr();
  x.cu_seqlens_k = (const int*)cu_k.data_ptr();
  x.total_q = (int)Tq;
  x.total_k = (int)Tk;
  check_rc(fa_fwd_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), (float*)LSE.data_ptr(), (int)B, (int)H, (int)max_q,
                     (int)max_k, (int)D, dtype_code(Q), causal ? 1 : 0, (float)(1.0 / std::sqrt((double)D)), &x,
                     current_stream(Q)),
           "fa_fwd_varlen");
  return {O, LSE};
}

std::tuple<Tensor, Tensor, Tensor> varlen_backward_launch(const Tensor& Q_, const Tensor& K_, const Tensor& V_, const Tensor& O_,
                                                          const Tensor& dO_, const Tensor& LSE, const Tensor& cu_q,
                                                          const Tensor& cu_k, int64_t max_q, int64_t max_k, bool causal,
                                                          double p_drop, int64_t seed, int64_t offset) {
  check_varlen(Q_, K_, V_, cu_q, cu_k);
  FA_ASSERT(O_.sizes() == Q_.sizes() && dO_.sizes() == Q_.sizes(), "O and dO must have Q's shape");
  FA_ASSERT(LSE.dim() == 2 && LSE.size(0) == Q_.size(1) && LSE.size(1) == Q_.size(0) && LSE.scalar_type() == at::kFloat &&
                LSE.is_contiguous() && LSE.device() == Q_.device(),
            "LSE must be contiguous float32 [H, total_q]");
  Tensor Q = packed(Q_), K = packed(K_), V = packed(V_), O = packed(O_), dO = packed(dO_);
  const int64_t Tq = Q.size(0), Tk = K.size(0), H = Q.size(1), D = Q.size(2), B = cu_q.numel() - 1;
  c10::OptionalDeviceGuard guard(Q.device());
  Tensor dQ = torch::empty({Tq, H, D}, Q.options());
  Tensor g = torch::empty({2, Tk, H, D}, Q.options());
  Tensor dK = g.select(0, 0), dV = g.select(0, 1);
  Tensor delta = torch::empty({H, Tq}, Q.options().dtype(at::kFloat));
  void* st = current_stream(Q);
  const float scale = (float)(1.0 / std::sqrt((double)D));
  const int dt = dtype_code(Q);
  mi355fa_opts x = make_opts(p_drop, seed, offset);
  Tensor qs;  // bf16: Q rows as the dQ launch multiplied them, for the dK/dV launch (mi355fa_opts.q_scaled)
  if (dt == MI355FA_BF16) {
    qs = torch::empty({Tq, H, D}, Q.options());
    x.q_scaled = qs.data_ptr();
  }
  x.cu_seqlens_q = (const int*)cu_q.data_ptr();
  x.cu_seqlens_k = (const int*)cu_k.data_ptr();
  x.total_q = (int)Tq;
  x.total_k = (int)Tk;
  check_rc(fa_bwd_dq_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), (const float*)LSE.data_ptr(),
                        dQ.data_ptr(), (float*)delta.data_ptr(), (int)B, (int)H, (int)max_q, (int)max_k, (int)D, dt,
                        causal ? 1 : 0, scale, &x, st),
           "fa_bwd_dq_varlen");
  check_rc(fa_bwd_dkv_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dO.data_ptr(), (const float*)LSE.data_ptr(),
                         (const float*)delta.data_ptr(), dK.data_ptr(), dV.data_ptr(), (int)B, (int)H, (int)max_q, (int)max_k,
                         (int)D, dt, causal ? 1 : 0, scale, &x, st),
           "fa_bwd_dkv_varlen");
  return {dQ, dK, dV};
}

class FlashAttnVarlenFn : public torch::autograd::Function<FlashAttnVarlenFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& Q, const Tensor& K, const Tensor& V, const Tensor& cu_q,
                        const Tensor& cu_k, int64_t max_q, int64_t max_k, bool is_causal, double p_drop, int64_t seed,
                        int64_t offset) {
    FA_ASSERT(Q.scalar_type() == at::kHalf || Q.scalar_type() == at::kBFloat16, "dtype must be float16 or bfloat16");
    Tensor Q_ = packed(Q), K_ = packed(K), V_ = packed(V);
    auto out = varlen_forward_launch(Q_, K_, V_, cu_q, cu_k, max_q, max_k, is_causal, p_drop, seed, offset);
    ctx->save_for_backward({Q_, K_, V_, std::get<0>(out), std::get<1>(out), cu_q, cu_k});
    ctx->saved_data["is_causal"] = is_causal;
    ctx->saved_data["max_q"] = max_q;
    ctx->saved_data["max_k"] = max_k;
    ctx->saved_data["p"] = p_drop;
    ctx->saved_data["seed"] = seed;
    ctx->saved_data["offset"] = offset;
    return std::get<0>(out);
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto s = ctx->get_saved_variables();
    auto g = varlen_backward_launch(s[0], s[1], s[2], s[3], grads[0], s[4], s[5], s[6], ctx->saved_data["max_q"].toInt(),
                                    ctx->saved_data["max_k"].toInt(), ctx->saved_data["is_causal"].toBool(),
                                    ctx->saved_data["p"].toDouble(), ctx->saved_data["seed"].toInt(),
                                    ctx->saved_data["offset"].toInt());
    return {std::get<0>(g), std::get<1>(g), std::get<2>(g), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor()};
  }
};

Tensor flash_attention_varlen(const Tensor& Q, const Tensor& K, const Tensor& V, const Tensor& cu_q, const Tensor& cu_k,
                              int64_t max_q, int64_t max_k, bool is_causal, double p_drop, int64_t seed, int64_t offset) {
  return FlashAttnVarlenFn::apply(Q, K, V, cu_q, cu_k, max_q, max_k, is_causal, p_drop, seed, offset);
}

// ---- attention dropout (include/mi355fa.h): the plain launchers with a (p, seed, offset) triple; views read in place ----
class FlashAttnDropoutFn : public torch::autograd::Function<FlashAttnDropoutFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& Q, const Tensor& K, const Tensor& V, bool is_causal, double p_drop,
                        int64_t seed, int64_t offset) {
    FA_ASSERT(Q.scalar_type() == at::kHalf || Q.scalar_type() == at::kBFloat16, "dtype must be float16 or bfloat16");
    FA_ASSERT(Q.dim() == 4 && (Q.size(3) == 64 || Q.size(3) == 128), "head dim must be 64 or 128");
    check_qkv(Q, K, V);
    Tensor Q_ = in_place(Q), K_ = in_place(K), V_ = in_place(V);
    if (K_.size(2) > 1 && K_.stride(2) != V_.stride(2)) {  // the kernels use one row stride for the K/V pair
      K_ = K_.contiguous();
      V_ = V_.contiguous();
    }
    auto out = forward_launch(Q_, K_, V_, is_causal, p_drop, seed, offset);
    ctx->save_for_backward({Q_, K_, V_, std::get<0>(out), std::get<1>(out)});
    ctx->saved_data["is_causal"] = is_causal;
    ctx->saved_data["p"] = p_drop;
    ctx->saved_data["seed"] = seed;
    ctx->saved_data["offset"] = offset;
    return std::get<0>(out);
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto s = ctx->get_saved_variables();
    auto g = backward_launch(s[0], s[1], s[2], s[3], in_place(grads[0]), s[4], ctx->saved_data["is_causal"].toBool(),
                             ctx->saved_data["p"].toDouble(), ctx->saved_data["seed"].toInt(),
                             ctx->saved_data["offset"].toInt());
    return {std::get<0>(g), std::get<1>(g), std::get<2>(g), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor flash_attention_dropout(const Tensor& Q, const Tensor& K, const Tensor& V, bool is_causal, double p_drop, int64_t seed,
                               int64_t offset) {
  return FlashAttnDropoutFn::apply(Q, K, V, is_causal, p_drop, seed, offset);
}

}  // namespace

PYBIND11_MODULE(_mi355fa_torch, m) {
  m.doc() = "C++ launchers and autograd function over libmi355fa.so (see My_FlashAttention_optimized.py)";
  m.def("flash_attention", &flash_attention, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"),
        pybind11::arg("is_causal") = false);
  m.def("forward_launch", &forward_launch, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"), pybind11::arg("is_causal"),
        pybind11::arg("dropout_p") = 0.0, pybind11::arg("seed") = 0, pybind11::arg("offset") = 0);
  m.def("backward_launch", &backward_launch, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"), pybind11::arg("O"),
        pybind11::arg("dO"), pybind11::arg("LSE"), pybind11::arg("is_causal"), pybind11::arg("dropout_p") = 0.0,
        pybind11::arg("seed") = 0, pybind11::arg("offset") = 0);
  m.def("flash_attention_varlen", &flash_attention_varlen, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"),
        pybind11::arg("cu_seqlens_q"), pybind11::arg("cu_seqlens_k"), pybind11::arg("max_seqlen_q"),
        pybind11::arg("max_seqlen_k"), pybind11::arg("is_causal") = false, pybind11::arg("dropout_p") = 0.0,
        pybind11::arg("seed") = 0, pybind11::arg("offset") = 0);
  m.def("varlen_forward_launch", &varlen_forward_launch, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"),
        pybind11::arg("cu_seqlens_q"), pybind11::arg("cu_seqlens_k"), pybind11::arg("max_seqlen_q"),
        pybind11::arg("max_seqlen_k"), pybind11::arg("is_causal"), pybind11::arg("dropout_p") = 0.0, pybind11::arg("seed") = 0,
        pybind11::arg("offset") = 0);
  m.def("varlen_backward_launch", &varlen_backward_launch, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"),
        pybind11::arg("O"), pybind11::arg("dO"), pybind11::arg("LSE"), pybind11::arg("cu_seqlens_q"),
        pybind11::arg("cu_seqlens_k"), pybind11::arg("max_seqlen_q"), pybind11::arg("max_seqlen_k"), pybind11::arg("is_causal"),
        pybind11::arg("dropout_p") = 0.0, pybind11::arg("seed") = 0, pybind11::arg("offset") = 0);
  m.def("flash_attention_dropout", &flash_attention_dropout, pybind11::arg("Q"), pybind11::arg("K"), pybind11::arg("V"),
        pybind11::arg("is_causal"), pybind11::arg("dropout_p"), pybind11::arg("seed"), pybind11::arg("offset") = 0);
  m.def("dropout_forward_launch", &forward_launch);    // the general launchers under their round-2 names
  m.def("dropout_backward_launch", &backward_launch);
  m.def("dropout_keep_scale", [](double p) { return (double)fa_dropout_keep_scale((float)p); });
  m.def("abi_version", []() { return fa_abi_version(); });
}
