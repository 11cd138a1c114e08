"""FlashAttention forward/backward for MI355X behind the reference's call surface.

Counterpart of the reference's code/My_FlashAttention_optimized.py: same module name, same
public names, argument order and defaults --

    flash_attention(Q, K, V, is_causal=False) -> O                       (M:169)
    FlashAttentionFunction.forward / .backward                           (M:130-166)
    flash_attention_forward(Q, K, V, is_causal) -> (O, LSE)              (M:14-60)
    flash_attention_backward(Q, K, V, O, dO, LSE, is_causal) -> (dQ, dK, dV)   (M:62-128)
    compare_with_sdpa(Q, K, V, is_causal)                                (M:172-212)

-- but the three Triton launches are three calls into libmi355fa.so (hand-written gfx950
HIP kernels, C ABI in include/mi355fa.h) on PyTorch's current stream.  PyTorch only
provides device memory, the stream and autograd.  There is no Triton and no fallback: on
a machine without the built libraries the import fails.

Host path: the launchers and the autograd function that `flash_attention` uses live in
_mi355fa_torch.so (csrc/torch_binding.cpp: the same checks / allocations / C-ABI calls as the
Python code below, without the interpreter -- at the reference's S = 512 benchmark points a
Python autograd.Function costs more host time per step than the three kernels take).
`FlashAttentionFunction` keeps the reference's Python class (same forward / backward
signatures, M:130-166) on top of the same launchers; _mi355fa.py is the ctypes view of the
C ABI used by the tools and the tests that drive the library directly.
"""
import torch
import torch.nn.functional as F

import _mi355fa as _fa
import _mi355fa_torch as _ext   # raises if the binding was not built (make -C csrc)

_DTYPES = {torch.float16: _fa.FP16, torch.bfloat16: _fa.BF16}

def _in_place(*tensors):
    """The reference makes every input contiguous (M:138-140,156): a 64 MiB copy per tensor at the headline size
    whenever Q/K/V are transposed views of a fused projection ([B,S,H,D] seen as [B,H,S,D]).  The kernels read such
    views in place and write O / dQ / dK / dV in the same storage order (fa_*_strided); only layouts they cannot address (non-unit head-dim stride, rows not 16-byte
    multiples, a base pointer off a 16-byte boundary, K and V with different sequence strides) are still copied --
    into a fresh (hence aligned) allocation."""
    return tuple(t if _fa.strided_ok(t) else t.clone(memory_format=torch.contiguous_format) for t in tensors)


def _kv_in_place(K, V):
    K, V = _in_place(K, V)
    if K.stride(2) != V.stride(2) and K.shape[2] > 1:   # the kernels use one row stride for the K/V pair
        K, V = K.contiguous(), V.contiguous()
    return K, V


def _check_qkv(Q, K, V):
    """The kernels take B and H from Q and address K / V slices as b*stride_b + h*stride_h: a K or V with fewer batches
    or heads would be read past its allocation (the reference's descriptors cover the whole tensor instead).  MQA / GQA
    callers pass K / V expanded to Q's head count (a stride-0 view is read in place)."""
    assert Q.ndim == 4 and K.ndim == 4 and V.ndim == 4
    assert K.shape[:2] == Q.shape[:2], "K must have Q's batch and head counts (expand shared K/V heads)"
    assert V.shape == K.shape, "K and V must have the same shape"
    assert Q.shape[-1] == K.shape[-1], "Q, K, V must share the head dim"
    assert Q.device == K.device == V.device, "Q, K, V must be on the same device"
    assert Q.dtype == K.dtype == V.dtype


def flash_attention_forward(Q, K, V, is_causal):
    """Allocate O / LSE and enqueue the forward kernel (M:14-60).  Q, K, V: contiguous, or strided views accepted
    by _mi355fa.strided_ok with K and V sharing their sequence stride.  Runs in _mi355fa_torch.forward_launch."""
    return _ext.forward_launch(Q, K, V, bool(is_causal))


def flash_attention_backward(Q, K, V, O, dO, LSE, is_causal):
    """Allocate dQ/dK/dV/delta and enqueue dQ (+delta) then dK/dV (M:62-128): _mi355fa_torch.backward_launch."""
    return _ext.backward_launch(Q, K, V, O, dO, LSE, bool(is_causal))


class FlashAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, V, is_causal: bool):
        assert Q.is_cuda and K.is_cuda and V.is_cuda
        assert Q.dtype in (torch.float16, torch.bfloat16)
        assert Q.dtype == K.dtype == V.dtype
        assert Q.shape[-1] == K.shape[-1] == V.shape[-1]
        assert Q.ndim == 4 and K.ndim == 4 and V.ndim == 4
        assert Q.shape[-1] in (64, 128), "head dim must be 64 or 128"
        _check_qkv(Q, K, V)   # beyond M:133-136: the raw-pointer kernels cannot bound a smaller K / V themselves
        if Q.is_contiguous() and K.is_contiguous() and V.is_contiguous() and not (
                (Q.data_ptr() | K.data_ptr() | V.data_ptr()) & 15):
            Q_, K_, V_ = Q, K, V
        else:                         # no copy for views the kernels can read in place (M:138-140 copies them)
            (Q_,) = _in_place(Q)
            K_, V_ = _kv_in_place(K, V)
        O, LSE = flash_attention_forward(Q_, K_, V_, is_causal)
        ctx.save_for_backward(Q_, K_, V_, O, LSE)
        ctx.is_causal = is_causal
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, O, LSE = ctx.saved_tensors
        dO_ = dO if (dO.is_contiguous() and not dO.data_ptr() & 15) else _in_place(dO)[0]
        dQ, dK, dV = flash_attention_backward(Q, K, V, O, dO_, LSE, ctx.is_causal)
        return dQ, dK, dV, None


def flash_attention(Q, K, V, is_causal=False):
    """M:169-170.  The autograd function behind it is the C++ twin of FlashAttentionFunction (torch_binding.cpp)."""
    return _ext.flash_attention(Q, K, V, bool(is_causal))


def flash_attention_varlen(Q, K, V, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, is_causal=False,
                           dropout_p=0.0, seed=0, offset=0):
    """Variable-length attention over PACKED sequences -- the extension the reference leaves as an exercise
    (Phase_6.md:119-178: "concatenate the batch into one long sequence and record where each sequence starts").

    Q: [total_q, H, D], K, V: [total_k, H, D] (fp16 / bf16, device); cu_seqlens_*: int32 device vectors of batch + 1
    prefix sums starting at 0; max_seqlen_*: Python ints >= the longest sequence (they size the launch grid).  Returns
    O [total_q, H, D]; differentiable w.r.t. Q, K, V.  Each sequence attends to itself only; is_causal applies each
    sequence's own top-left aligned mask.  No padding is computed: workgroups beyond a sequence's length exit at once.
    dropout_p / seed / offset: attention dropout as in flash_attention_dropout; the mask of sequence b is the one the same
    sequence gets at batch index b of a padded [B, H, S, D] launch with the same (seed, offset)."""
    return _ext.flash_attention_varlen(Q, K, V, cu_seqlens_q, cu_seqlens_k, int(max_seqlen_q), int(max_seqlen_k),
                                       bool(is_causal), float(dropout_p), int(seed), int(offset))


def flash_attention_dropout(Q, K, V, is_causal=False, dropout_p=0.0, seed=0, offset=0):
    """Attention with dropout on the attention weights -- the other extension the reference leaves as an exercise
    (Phase_6.md:54-113): P is masked and rescaled by 1 / (1 - p) inside the tile loop, and the backward regenerates the
    SAME mask from (seed, offset) with Philox4x32-10 instead of storing it (include/mi355fa.h, fa_*_dropout, gives the
    exact counter layout).  Q, K, V: [B, H, S, D] fp16 / bf16 device tensors (strided views are read in place, as in
    flash_attention); dropout_p in [0, 1), quantised to 1/256;
    seed / offset: Python ints (the caller owns the RNG stream: pass a fresh offset per layer and step).  Differentiable
    w.r.t. Q, K, V.  dropout_p = 0 is flash_attention."""
    if dropout_p == 0.0:
        return flash_attention(Q, K, V, is_causal)
    return _ext.flash_attention_dropout(Q, K, V, bool(is_causal), float(dropout_p), int(seed), int(offset))


def sdpa_reference(Q, K, V, is_causal):
    """torch SDPA on the device, fp16/bf16 (the reference pins the FLASH backend, M:178;
    here whatever backend this PyTorch-ROCm build selects)."""
    return F.scaled_dot_product_attention(Q, K, V, attn_mask=None, dropout_p=0.0, is_causal=is_causal)


def compare_with_sdpa(Q, K, V, is_causal, verbose=True):
    """Fwd+bwd of SDPA and of flash_attention on the same inputs and dO; verifies O, dQ, dK, dV
    in that order (M:172-212).  Returns the four metric dicts."""
    Q_ref = Q.detach().clone().requires_grad_(True)
    K_ref = K.detach().clone().requires_grad_(True)
    V_ref = V.detach().clone().requires_grad_(True)
    O_ref = sdpa_reference(Q_ref, K_ref, V_ref, is_causal)
    dO = torch.randn_like(O_ref)
    O_ref.backward(dO)
    dQ_ref, dK_ref, dV_ref = Q_ref.grad, K_ref.grad, V_ref.grad

    Q_ = Q.detach().clone().requires_grad_(True)
    K_ = K.detach().clone().requires_grad_(True)
    V_ = V.detach().clone().requires_grad_(True)
    O = flash_attention(Q_, K_, V_, is_causal=is_causal)
    O.backward(dO)
    dQ, dK, dV = Q_.grad, K_.grad, V_.grad

    from _verify_func import verify_results
    out = {}
    for name, ref, got in (("O", O_ref, O), ("dQ", dQ_ref, dQ), ("dK", dK_ref, dK), ("dV", dV_ref, dV)):
        if verbose:
            print("=" * 30 + " " + name + " test " + "=" * 30)
        out[name] = verify_results(ref, got, name=name, verbose=verbose)
    return out


if __name__ == "__main__":
    DEVICE = torch.device(torch.cuda.current_device())
    B, H, S_q, S_k, D = 4, 8, 256, 256, 64
    Q = torch.randn((B, H, S_q, D), dtype=torch.float16, device=DEVICE)
    K = torch.randn((B, H, S_k, D), dtype=torch.float16, device=DEVICE)
    V = torch.randn((B, H, S_k, D), dtype=torch.float16, device=DEVICE)
    compare_with_sdpa(Q, K, V, is_causal=True)
