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
a machine without the built library the import fails.
"""
import torch
import torch.nn.functional as F

import _mi355fa as _fa

_DTYPES = {torch.float16: _fa.FP16, torch.bfloat16: _fa.BF16}
_fwd, _bwd_dq, _bwd_dkv = _fa.lib.fa_fwd_strided, _fa.lib.fa_bwd_dq_strided, _fa.lib.fa_bwd_dkv_strided


try:   # the raw hipStream_t of PyTorch's current stream without building a torch.cuda.Stream object (~3 us per call)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover - older / newer PyTorch without the private accessor
    _raw_stream = None


def _stream(index=None):
    if _raw_stream is not None and index is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream().cuda_stream


class _OnDevice:
    """`with torch.cuda.device(d)` only when `d` is not already current (the context manager costs ~5 us per call,
    which is most of the host time at the reference's small benchmark shapes)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _in_place(*tensors):
    """The reference makes every input contiguous (M:138-140,156): a 64 MiB copy per tensor at the headline size
    whenever Q/K/V are transposed views of a fused projection ([B,S,H,D] seen as [B,H,S,D]).  The kernels read such
    views in place (fa_*_strided); only layouts they cannot address (non-unit head-dim stride, rows not 16-byte
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
    by _mi355fa.strided_ok with K and V sharing their sequence stride."""
    _check_qkv(Q, K, V)
    B, H, S_q, D = Q.shape
    _, _, S_k, _ = K.shape
    O = torch.empty((B, H, S_q, D), dtype=Q.dtype, device=Q.device)
    LSE = torch.empty((B, H, S_q), dtype=torch.float32, device=Q.device)
    sq, sk, sv = _fa.strides3(Q), _fa.strides3(K), _fa.strides3(V)   # keep the ctypes arrays alive over the call
    with _OnDevice(Q.device):
        rc = _fwd(Q.data_ptr(), sq, K.data_ptr(), sk, V.data_ptr(), sv, O.data_ptr(), LSE.data_ptr(),
                  B, H, S_q, S_k, D, _DTYPES[Q.dtype], 1 if is_causal else 0, 1 / (D ** 0.5), _stream(Q.device.index))
    if rc:
        _fa.check(rc, "fa_fwd")
    return O, LSE


def flash_attention_backward(Q, K, V, O, dO, LSE, is_causal):
    """Allocate dQ/dK/dV/delta and enqueue dQ (+delta) then dK/dV (M:62-128)."""
    _check_qkv(Q, K, V)
    assert O.shape == Q.shape and dO.shape == Q.shape and LSE.shape == Q.shape[:3]
    assert O.device == dO.device == LSE.device == Q.device
    B, H, S_q, D = Q.shape
    _, _, S_k, _ = K.shape
    if S_q == S_k:   # self-attention: one allocation for the three gradients (M:71-73 makes three)
        dQ, dK, dV = torch.empty((3, B, H, S_q, D), dtype=Q.dtype, device=Q.device).unbind(0)
    else:
        dQ = torch.empty((B, H, S_q, D), dtype=Q.dtype, device=Q.device)
        dK, dV = torch.empty((2, B, H, S_k, D), dtype=Q.dtype, device=Q.device).unbind(0)
    delta = torch.empty((B, H, S_q), dtype=torch.float32, device=Q.device)
    dt, causal, scale = _DTYPES[Q.dtype], 1 if is_causal else 0, 1 / (D ** 0.5)
    if not O.is_contiguous() or O.data_ptr() & 15:   # normally the tensor flash_attention_forward returned
        O = O.clone(memory_format=torch.contiguous_format)
    sq, sk, sv, sdo = _fa.strides3(Q), _fa.strides3(K), _fa.strides3(V), _fa.strides3(dO)
    with _OnDevice(Q.device):
        s = _stream(Q.device.index)
        rc = _bwd_dq(Q.data_ptr(), sq, K.data_ptr(), sk, V.data_ptr(), sv, O.data_ptr(),
                     dO.data_ptr(), sdo, LSE.data_ptr(), dQ.data_ptr(), delta.data_ptr(),
                     B, H, S_q, S_k, D, dt, causal, scale, s)
        if rc:
            _fa.check(rc, "fa_bwd_dq")
        # same stream, after dQ: the dK/dV kernel reads the delta the dQ kernel wrote (K:376)
        rc = _bwd_dkv(Q.data_ptr(), sq, K.data_ptr(), sk, V.data_ptr(), sv, dO.data_ptr(), sdo,
                      LSE.data_ptr(), delta.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                      B, H, S_q, S_k, D, dt, causal, scale, s)
        if rc:
            _fa.check(rc, "fa_bwd_dkv")
    return dQ, dK, dV


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
    return FlashAttentionFunction.apply(Q, K, V, is_causal)


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
