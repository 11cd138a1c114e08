"""GPU parity tests (run with -m gpu on an MI355X): the HIP kernels, called through the C-ABI
library via the Python host, against (1) the reference's own kernel outputs (tests/golden),
(2) the fp64 oracle on the same seeded inputs, (3) torch SDPA on the device, and (4) at the
BASELINE.json sizes, size-independent properties.

Tolerances: fp16 relFro <= 1e-3 vs fp64 (BASELINE "within 1e-3 rel"; the reference
kernels themselves sit at 2.4-3.2e-4) and verify_results PASS at rtol=1e-2, atol=1e-3, cos>0.999;
bf16 relFro <= 2x that of PyTorch's own bf16 SDPA on the same inputs (bf16 output rounding alone
is ~2e-3) and cos > 0.999; |LSE - logsumexp| < 1e-3 (Phase_3.md:752).
"""
import pytest
import torch

import fa_oracle as fo
from _util import golden_names, load_golden, rand_inputs

pytestmark = pytest.mark.gpu

F16, BF16 = torch.float16, torch.bfloat16


def _host():
    import My_FlashAttention_optimized as M
    return M


@pytest.fixture(params=[0, 1, 2, 3, 4], ids=["auto", "family1", "family2", "family3", "family4"])
def impl(request):
    """Run a test under the automatic schedule rule and with each schedule family forced (a family a launch cannot use
    -- head dim 128 for the D = 64-only families, packed batches for the 64-rows-per-wave forward -- falls back)."""
    import ctypes
    import _mi355fa as fa
    fn = fa.lib.fa_debug_force_impl
    fn.argtypes = [ctypes.c_int] * 3
    fn.restype = None
    fn(request.param, request.param, request.param)
    yield request.param
    fn(0, 0, 0)


def run_gpu(Q, K, V, dO, causal):
    """fwd + bwd through the autograd binding; everything returned on the CPU."""
    M = _host()
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention(q, k, v, is_causal=causal)
    o.backward(dO.cuda())
    torch.cuda.synchronize()
    return {"O": o.detach().cpu(), "dQ": q.grad.cpu(), "dK": k.grad.cpu(), "dV": v.grad.cpu()}


def run_gpu_raw(Q, K, V, dO, causal, workspace=False):
    """The launchers directly: also returns LSE and delta (not visible through autograd).  workspace=True: the backward as
    INTEGRATION.md section B binds it -- fa_bwd_dq_ex / fa_bwd_dkv_ex with mi355fa_opts.q_scaled -- through raw ctypes."""
    M = _host()
    q, k, v, do = (x.cuda() for x in (Q, K, V, dO))
    O, LSE = M.flash_attention_forward(q, k, v, causal)
    import _mi355fa as fa
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    dQ, dK, dV = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(LSE)
    st = torch.cuda.current_stream().cuda_stream
    dt = M._DTYPES[q.dtype]
    if workspace:
        import ctypes
        ws = torch.empty_like(q)
        o = fa.Opts.make(q_scaled=ws.data_ptr())
        fa.check(fa.lib.fa_bwd_dq_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), O.data_ptr(), do.data_ptr(), LSE.data_ptr(),
                                     dQ.data_ptr(), delta.data_ptr(), B, H, Sq, Sk, D, dt, int(causal), D ** -0.5,
                                     ctypes.byref(o), st), "dq_ex")
        fa.check(fa.lib.fa_bwd_dkv_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(), LSE.data_ptr(), delta.data_ptr(),
                                      dK.data_ptr(), dV.data_ptr(), B, H, Sq, Sk, D, dt, int(causal), D ** -0.5,
                                      ctypes.byref(o), st), "dkv_ex")
        torch.cuda.synchronize()
        return {k_: t.cpu() for k_, t in dict(O=O, LSE=LSE, delta=delta, dQ=dQ, dK=dK, dV=dV).items()}
    fa.check(fa.lib.fa_bwd_dq(q.data_ptr(), k.data_ptr(), v.data_ptr(), O.data_ptr(), do.data_ptr(), LSE.data_ptr(),
                              dQ.data_ptr(), delta.data_ptr(), B, H, Sq, Sk, D, dt, int(causal), D ** -0.5, st), "dq")
    fa.check(fa.lib.fa_bwd_dkv(q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(), LSE.data_ptr(), delta.data_ptr(),
                               dK.data_ptr(), dV.data_ptr(), B, H, Sq, Sk, D, dt, int(causal), D ** -0.5, st), "dkv")
    torch.cuda.synchronize()
    return {k_: t.cpu() for k_, t in dict(O=O, LSE=LSE, delta=delta, dQ=dQ, dK=dK, dV=dV).items()}


# ---------------------------------------------------------------- (1) golden vectors
@pytest.mark.parametrize("name", golden_names())
def test_against_reference_kernel_outputs(name, impl):
    g = load_golden(name)
    m = g["meta"]
    r = run_gpu_raw(g["Q"], g["K"], g["V"], g["dO"], m["causal"])
    gt = fo.attention_fp64(g["Q"], g["K"], g["V"], g["dO"], m["causal"])
    for k in ("O", "dQ", "dK", "dV"):
        ours, ref = fo.rel_fro(gt[k], r[k]), fo.rel_fro(gt[k], g["ref_" + k])
        assert ours < 1e-3, (k, ours)
        assert ours < 1.25 * ref + 2e-5, (k, ours, ref)       # as accurate as the reference's kernels
        assert fo.rel_fro(g["ref_" + k], r[k]) < 6e-4, k       # and the same numbers up to fp16 rounding
        assert fo.verify_metrics(g["ref_" + k], r[k])["passed"], k
    assert (r["LSE"] - g["ref_LSE"]).abs().max() < 1e-5
    assert (r["delta"] - g["ref_delta"]).abs().max() < 4e-3     # delta comes from each side's own rounded O


# ---------------------------------------------------------------- (2) fp64 oracle, seeded inputs
SHAPES = [
    # B, H, Sq, Sk, D, causal
    (2, 4, 256, 256, 64, False),     # BASELINE configs[0]
    (2, 4, 256, 256, 64, True),
    (4, 8, 256, 256, 64, True),      # reference __main__ (M:216-220)
    (1, 2, 128, 320, 64, False),     # cross attention (Phase_3.md:263)
    (1, 2, 1024, 4096, 64, False),   # Sq=1024, Sk=4096 (Phase_3.md:263)
    (1, 2, 384, 128, 64, True),      # Sq > Sk, top-left causal
    (1, 2, 500, 500, 64, True),      # ragged (Phase_3.md:260): tails masked
    (1, 2, 500, 500, 64, False),
    (1, 1, 77, 333, 64, False),
    (1, 1, 1, 1, 64, True),          # minimum sizes
    (1, 1, 1, 700, 64, False),
    (1, 1, 129, 65, 64, True),
    (2, 3, 1024, 1024, 64, True),
    (1, 1, 2048, 2048, 64, False),
    (1, 1, 256, 256, 128, True),     # D = 128 (fixture iv shape)
    (1, 2, 500, 500, 128, True),
    (1, 2, 333, 600, 128, False),
    # causal with S_k > S_q: top-left aligned mask, keys >= S_q are invisible to every query -- their dK / dV
    # rows must come out exactly 0 (zero-tile workgroups, paired passes mixing empty and non-empty key tiles)
    (1, 2, 128, 320, 64, True),
    (1, 1, 77, 333, 64, True),
    (2, 3, 256, 1024, 64, True),
    (1, 2, 128, 320, 128, True),
    (1, 1, 77, 333, 128, True),
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [F16, BF16])
def test_against_fp64_oracle(shape, dtype, impl):
    # D = 128: forward and dQ have one schedule; the dK/dV kernel has both families
    B, H, Sq, Sk, D, causal = shape
    Q, K, V, dO = rand_inputs(B, H, Sq, Sk, D, dtype, seed=11)
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    r = run_gpu_raw(Q, K, V, dO, causal)
    # fp16: the reference's own LSE bound (Phase_3.md:753).  bf16: the kernels fold softmax_scale*log2e into the
    # resident 8-bit-mantissa operand (fa_common.h kFoldScale), which moves a score by ~2^-9 relative: the LSE
    # (values ~6 here) is then good to ~2e-3 absolute, 3e-4 relative -- inside BASELINE's "1e-3 rel".
    lse_err = (r["LSE"].double() - gt["LSE"]).abs()
    assert lse_err.max() < (1e-2 if dtype == BF16 else 1e-3)   # bf16: ~2^-9 * max|score| (early causal rows: LSE = the score)
    assert fo.rel_fro(gt["LSE"], r["LSE"]) < 1e-3
    # delta is computed from the ROUNDED O as in the reference (K:210-211): its error is |dO| * ulp(O) * sqrt(D)
    assert (r["delta"].double() - gt["delta"]).abs().max() < (1e-1 if dtype == BF16 else 1.5e-2)
    if causal and Sk > Sq:                  # keys no query can see: every element of their gradient rows is written, as 0
        assert (r["dK"][:, :, Sq:] == 0).all() and (r["dV"][:, :, Sq:] == 0).all()
    names = ["O", "dQ", "dK", "dV"]
    for k in list(names):
        if gt[k].abs().max() < 1e-9:        # a single visible key: softmax is constant, dQ = dK = 0 exactly
            assert r[k].float().abs().max() < 2e-3, k
            names.remove(k)
    if dtype == F16:
        for k in names:
            assert fo.rel_fro(gt[k], r[k]) < 1e-3, (k, fo.rel_fro(gt[k], r[k]))
            assert fo.verify_metrics(gt[k], r[k])["passed"], k
    else:
        peer = dict(zip(("O", "dQ", "dK", "dV"), fo.cpu_sdpa(Q, K, V, causal, dO)))  # PyTorch's own bf16 SDPA (CPU)
        for k in names:
            ours, theirs = fo.rel_fro(gt[k], r[k]), fo.rel_fro(gt[k], peer[k])
            assert ours < max(2 * theirs, 4e-3), (k, ours, theirs)
            assert fo.verify_metrics(gt[k], r[k], rtol=2e-2, atol=2e-2)["cos"] > 0.999


def test_autograd_path_and_strided_inputs():
    """flash_attention through autograd, fed [B,S,H,D]-strided views (made contiguous like M:138-140)."""
    B, H, S, D = 2, 4, 320, 64
    Q, K, V, dO = rand_inputs(B, H, S, S, D, F16, seed=5)
    gt = fo.attention_fp64(Q, K, V, dO, True)
    M = _host()
    mk = lambda x: x.cuda().transpose(1, 2).contiguous().transpose(1, 2).requires_grad_(True)  # non-contiguous view
    q, k, v = mk(Q), mk(K), mk(V)
    assert not q.is_contiguous()
    o = M.flash_attention(q, k, v, True)
    o.backward(dO.cuda().transpose(1, 2).contiguous().transpose(1, 2))
    for name, t in (("O", o), ("dQ", q.grad), ("dK", k.grad), ("dV", v.grad)):
        assert fo.rel_fro(gt[name], t.detach().cpu()) < 1e-3, name
    assert o.dtype == F16 and q.grad.shape == q.shape


def test_python_autograd_class_and_cpp_autograd_function_agree():
    """FlashAttentionFunction (the reference's Python class, M:130-166) and flash_attention (its C++ twin in
    _mi355fa_torch.so) run the same launchers: bit-identical outputs and gradients, contiguous and strided inputs."""
    M = _host()
    for causal in (False, True):
        Q, K, V, dO = (x.cuda() for x in rand_inputs(2, 3, 320, 448, 64, BF16, seed=13))
        for view in (False, True):
            outs = []
            for fn in (M.flash_attention, lambda q, k, v, c: M.FlashAttentionFunction.apply(q, k, v, c)):
                mk = (lambda x: x.transpose(1, 2).contiguous().transpose(1, 2)) if view else (lambda x: x.clone())
                q, k, v = (mk(x).requires_grad_(True) for x in (Q, K, V))
                o = fn(q, k, v, causal)
                o.backward(mk(dO))
                outs.append((o.detach(), q.grad, k.grad, v.grad))
            for a, b in zip(*outs):
                assert torch.equal(a, b)
    # no grad needed -> no graph, same numbers
    with torch.no_grad():
        assert torch.equal(M.flash_attention(Q, K, V, True), M.flash_attention_forward(Q, K, V, True)[0])
    # only some inputs need a gradient
    q, k, v = Q.clone().requires_grad_(True), K.clone(), V.clone()
    M.flash_attention(q, k, v, False).backward(dO)
    assert q.grad is not None and k.grad is None


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_rescale_branch_is_exercised_by_a_late_spike(impl, dtype):
    """Online-softmax max must jump in a LATE tile: one key far down the sequence dominates one query
    (cdna guide rule 26: bounded random data rarely takes the rescale branch late).  bf16: its forward
    score chain starts from the -m block, dQ / dK/dV fold the scale into Q / K (one more 2^-9 rounding of the exponent
    argument) -- the same spikes must hold there at the bf16 tolerances of test_against_fp64_oracle."""
    B, H, S, D = 1, 1, 640, 64
    Q, K, V, dO = rand_inputs(B, H, S, S, D, dtype, seed=9)
    K[0, 0, 517] = (Q[0, 0, 600].float() * 0.9).to(dtype)     # huge score for row 600 at key 517 (tile 8)
    K[0, 0, 70] = (Q[0, 0, 100].float() * 0.7).to(dtype)
    # a jump of > 2^13 in one step: forces the lazy-max forward tile to bail out to the exact path late
    # (raw score 1.7 * |q|^2 ~ 110 -> exp2(110 * 0.18) ~ 2^20 against the stale max)
    K[0, 0, 450] = (Q[0, 0, 520].float() * 1.7).to(dtype)
    for causal in (False, True):
        gt = fo.attention_fp64(Q, K, V, dO, causal)
        r = run_gpu_raw(Q, K, V, dO, causal)
        if dtype == F16:
            for k in ("O", "dQ", "dK", "dV"):
                assert fo.rel_fro(gt[k], r[k]) < 1e-3, (k, causal)
            assert (r["LSE"].double() - gt["LSE"]).abs().max() < 1e-3
        else:
            peer = dict(zip(("O", "dQ", "dK", "dV"), fo.cpu_sdpa(Q, K, V, causal, dO)))
            for k in ("O", "dQ", "dK", "dV"):
                ours, theirs = fo.rel_fro(gt[k], r[k]), fo.rel_fro(gt[k], peer[k])
                assert ours < max(2 * theirs, 4e-3), (k, causal, ours, theirs)
            # the spiked rows have |score * log2e| ~ 20: 2^-9 relative on that is ~4e-2 absolute
            assert fo.rel_fro(gt["LSE"], r["LSE"]) < 2e-3


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("D", [64, 128])
def test_family4_forward_overflow_takes_the_exact_second_attempt(dtype, D):
    """Round 3: the one-wave-per-SIMD forward (fa_fwd_v4.hip) keeps ONE row constant per pass (the maximum over the row's
    first 32 keys) and checks the row sums at the end of the pass; a key far down the sequence that beats that constant by
    more than the number format allows (2^15 for fp16, 2^100 for bf16) must send the pass through its exact second attempt
    -- same tolerances as everywhere else, forward only (the backward does not depend on the forward's schedule)."""
    import ctypes
    import _mi355fa as fa
    M = _host()
    fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    B, H, S = 1, 2, 768
    Q, K, V, dO = rand_inputs(B, H, S, S, D, dtype, seed=31)
    # row 600's score against key 517: |q|^2 * gain / sqrt(D) * log2(e) ~ 2^138+ (bf16) / ~ 2^37+ (fp16) above its first keys'
    K[0, 0, 517] = (Q[0, 0, 600].float() * (12.0 if dtype == BF16 else 3.2)).to(dtype)
    K[0, 1, 300] = (Q[0, 1, 700].float() * (12.0 if dtype == BF16 else 3.2)).to(dtype)
    counter = torch.zeros(4, dtype=torch.int32, device="cuda")   # the debug buffer: its first word counts exact second attempts
    setbuf = fa.lib.fa_debug_set_buffer
    setbuf.restype, setbuf.argtypes = None, [ctypes.c_void_p]

    def redo():
        torch.cuda.synchronize()
        return int(counter[0].item())
    for causal in (False, True):
        gt = fo.attention_fp64(Q, K, V, dO, causal)
        outs = {}
        for fam in (1, 4):
            fa.lib.fa_debug_force_impl(fam, 0, 0)
            before = redo()
            setbuf(counter.data_ptr())
            try:
                O, LSE = M.flash_attention_forward(Q.cuda(), K.cuda(), V.cuda(), causal)
                torch.cuda.synchronize()
            finally:
                setbuf(None)
                fa.lib.fa_debug_force_impl(0, 0, 0)
            took = redo() - before
            # the two spiked rows sit in two different (head, query tile) workgroups: exactly those redo their pass
            assert took == (2 if fam == 4 else 0), (fam, causal, took)
            outs[fam] = (O.cpu(), LSE.cpu())
            assert torch.isfinite(O.float()).all() and torch.isfinite(LSE).all(), (fam, causal)
        e1, e4 = fo.rel_fro(gt["O"], outs[1][0]), fo.rel_fro(gt["O"], outs[4][0])
        assert e4 < max(1.5 * e1, 1e-3 if dtype == F16 else 6e-3), (causal, e1, e4)
        assert fo.rel_fro(gt["LSE"], outs[4][1]) < max(2 * fo.rel_fro(gt["LSE"], outs[1][1]), 2e-3), causal


def test_dkv_family3_is_bit_identical_to_family2():
    """Round 3: the one-wave-per-SIMD dK/dV kernel (fa_bwd_dkv_v3.hip) keeps the maths, the rounding points and the
    accumulation order of family 2 -- identical bits, causal and full, both dtypes, ragged and cross-attention shapes
    (tools/check_family.py runs the long list, incl. the headline shape)."""
    import ctypes
    import _mi355fa as fa
    M = _host()
    fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    for dtype in (BF16, F16):
        for causal in (False, True):
            for (B, H, Sq, Sk) in ((2, 3, 640, 640), (1, 2, 500, 500), (1, 2, 77, 333), (1, 2, 333, 77), (1, 1, 1024, 256)):
                Q, K, V, dO = (x.cuda() for x in rand_inputs(B, H, Sq, Sk, 64, dtype, seed=Sq + Sk))
                O, LSE = M.flash_attention_forward(Q, K, V, causal)
                got = {}
                for fam in (2, 3):
                    fa.lib.fa_debug_force_impl(0, 0, fam)
                    try:
                        got[fam] = M.flash_attention_backward(Q, K, V, O, dO, LSE, causal)
                        torch.cuda.synchronize()
                    finally:
                        fa.lib.fa_debug_force_impl(0, 0, 0)
                for a, b in zip(got[2][1:], got[3][1:]):      # dK, dV
                    assert torch.equal(a, b), (dtype, causal, Sq, Sk)


def test_dkv_family4_against_family3():
    """Round 4: the dK/dV kernel on the pinned accumulator file (fa_bwd_dkv_v4.hip).  Without a mask it keeps family 3's
    accumulation order: identical bits.  Under the causal mask it takes the two query tiles level with the key tile LAST and
    visits only the 9 visible blocks of that region per wave: the fp32 sums run in another order, so dK / dV may differ from
    family 3's by one unit in the last place of the 16-bit output on a small share of the elements -- and by nothing else."""
    import ctypes
    import _mi355fa as fa
    M = _host()
    fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    pick = fa.lib.fa_debug_pick
    pick.argtypes = [ctypes.c_int] * 8
    took4 = 0
    for dtype in (BF16, F16):
        for causal in (False, True):
            for (B, H, Sq, Sk) in ((1, 2, 256, 256), (2, 3, 768, 768), (1, 2, 1024, 256), (1, 1, 128, 512), (2, 2, 1280, 1280),
                                   (1, 2, 512, 1024), (1, 2, 500, 500), (4, 32, 512, 512),
                                   (6, 48, 768, 768)):   # > one item per persistent workgroup, causal and full
                Q, K, V, dO = (x.cuda() for x in rand_inputs(B, H, Sq, Sk, 64, dtype, seed=Sq + Sk))
                O, LSE = M.flash_attention_forward(Q, K, V, causal)
                got = {}
                for fam in (3, 4):
                    fa.lib.fa_debug_force_impl(0, 0, fam)
                    try:
                        took4 += fam == 4 and pick(2, 64, int(dtype == BF16), int(causal), B, H, Sq, Sk) == 4
                        got[fam] = M.flash_attention_backward(Q, K, V, O, dO, LSE, causal)
                        torch.cuda.synchronize()
                    finally:
                        fa.lib.fa_debug_force_impl(0, 0, 0)
                for a, b in zip(got[3][1:], got[4][1:]):      # dK, dV
                    assert not torch.isnan(b.float()).any(), (dtype, causal, Sq, Sk)
                    if not causal:
                        assert torch.equal(a, b), (dtype, causal, Sq, Sk)
                    else:
                        # one unit in the last place of the element -- or, where the element is the small remainder of a
                        # cancellation, of the typical element: fp32 sums of another order differ relative to their terms
                        ulp = (a.float().abs() + a.float().abs().mean()) * (2.0 ** -7 if dtype == BF16 else 2.0 ** -10)
                        assert ((a.float() - b.float()).abs() <= 1.01 * ulp).all(), (dtype, causal, Sq, Sk)
                        assert (a != b).float().mean() < 0.02, (dtype, causal, Sq, Sk)
    assert took4 >= 12   # the shapes above are mostly ones family 4 really runs (fp16 causal and ragged ones fall back)


def test_dq_family4_is_bit_identical_to_family3():
    """Round 4: the one-wave-per-SIMD dQ kernel (fa_bwd_dq_v4.hip: 64 rows per wave, per-wave diagonal phase, diagonal mask in
    the chain-start operand) keeps the maths, the rounding points and the accumulation order of family 3 -- identical dQ and
    delta bits, causal and full, both dtypes; shapes it takes itself (whole 128-key tiles; causal: whole 256-row tiles covered
    by S_k, one and several query tiles, paired and unpaired grids, S_q < S_k, a ragged S_q) and shapes that must fall back."""
    import ctypes
    import _mi355fa as fa
    M = _host()
    fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    pick = fa.lib.fa_debug_pick
    pick.argtypes = [ctypes.c_int] * 8
    took4 = 0
    for dtype in (BF16, F16):
        for causal in (False, True):
            for (B, H, Sq, Sk) in ((1, 2, 256, 256), (2, 3, 768, 768), (1, 2, 256, 1024), (1, 1, 200, 512), (2, 2, 1280, 1280),
                                   (1, 2, 1024, 128), (1, 2, 500, 500), (4, 32, 512, 512),
                                   (6, 48, 768, 768)):   # > one item per persistent workgroup, causal and full
                Q, K, V, dO = (x.cuda() for x in rand_inputs(B, H, Sq, Sk, 64, dtype, seed=Sq + Sk))
                O, LSE = M.flash_attention_forward(Q, K, V, causal)
                got = {}
                for fam in (3, 4):
                    fa.lib.fa_debug_force_impl(0, fam, 0)
                    try:
                        took4 += fam == 4 and pick(1, 64, int(dtype == BF16), int(causal), B, H, Sq, Sk) == 4
                        dQ = torch.full_like(Q, float("nan"))
                        delta = torch.full_like(LSE, float("nan"))
                        st = torch.cuda.current_stream().cuda_stream
                        rc = fa.lib.fa_bwd_dq(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), LSE.data_ptr(),
                                              dQ.data_ptr(), delta.data_ptr(), B, H, Sq, Sk, 64, int(dtype == BF16), int(causal),
                                              64 ** -0.5, st)
                        assert rc == 0, fa.lib.fa_last_error()
                        torch.cuda.synchronize()
                        got[fam] = (dQ, delta)
                    finally:
                        fa.lib.fa_debug_force_impl(0, 0, 0)
                assert not torch.isnan(got[4][0].float()).any() and not torch.isnan(got[4][1]).any(), (dtype, causal, Sq, Sk)
                for a, b in zip(got[3], got[4]):
                    assert torch.equal(a, b), (dtype, causal, Sq, Sk)
    assert took4 >= 20   # the shapes above are mostly ones family 4 really runs


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
def test_large_magnitude_scores_bf16(impl, causal):
    """Scaled-up Q and K (|s * log2e / sqrt(D)| up to ~50-100): the folded-scale bf16 kernels carry the exponent
    argument through one extra 8-bit rounding, which randn inputs (|arg| < 10) never stress.  Softmax is nearly one-hot
    here.  O and dQ stay within 2.5x PyTorch's own bf16 SDPA (measured 2.0x): forward and dQ round the SAME operand (Q), so the P
    the dQ kernel recomputes is consistent with the forward's LSE.  The dK/dV kernel folds the scale into K instead
    (its Q streams through LDS), so its P differs from the forward's by the two independent roundings,
    measured ~4.5e-4 * max|arg| in relative Frobenius norm ("scale fold"): dK / dV get that
    wider, documented bound (2 % at |arg| = 44; randn inputs, |arg| < 10, stay at PyTorch's own bf16 level)."""
    B, H, S, D = 1, 2, 384, 64
    Q, K, V, dO = rand_inputs(B, H, S, S, D, BF16, seed=21)
    Q, K = (Q.float() * 2.5).to(BF16), (K.float() * 2.5).to(BF16)
    arg = (Q.float() @ K.float().transpose(-1, -2)).abs().max().item() * 1.4427 / 8
    assert 40 < arg < 160, arg
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    r = run_gpu_raw(Q, K, V, dO, causal)
    peer = dict(zip(("O", "dQ", "dK", "dV"), fo.cpu_sdpa(Q, K, V, causal, dO)))
    for k in ("O", "dQ", "dK", "dV"):
        ours, theirs = fo.rel_fro(gt[k], r[k]), fo.rel_fro(gt[k], peer[k])
        assert torch.isfinite(r[k].float()).all(), k
        bound = max(2.5 * theirs, 6e-3) if k in ("O", "dQ") else max(2.5 * theirs, 6.0e-4 * arg)
        assert ours < bound, (k, ours, theirs, bound)
    assert fo.rel_fro(gt["LSE"], r["LSE"]) < 2e-3


@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
def test_large_magnitude_scores_bf16_with_the_scaled_q_workspace(causal):
    """The same stress through flash_attention(): its backward hands the dQ and dK/dV launches one workspace
    (mi355fa_opts.q_scaled), the dK/dV kernel then recomputes P from exactly the rounded operand LSE came from, and dK / dV
    hold the bound of O and dQ (2.5x PyTorch's own bf16 SDPA) instead of the magnitude-dependent one above."""
    M = _host()
    B, H, S, D = 1, 2, 384, 64
    Q, K, V, dO = rand_inputs(B, H, S, S, D, BF16, seed=21)
    Q, K = (Q.float() * 2.5).to(BF16), (K.float() * 2.5).to(BF16)
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    peer = dict(zip(("O", "dQ", "dK", "dV"), fo.cpu_sdpa(Q, K, V, causal, dO)))
    q, k, v = (x.cuda().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention(q, k, v, causal)
    o.backward(dO.cuda())
    raw = run_gpu_raw(Q, K, V, dO, causal)               # no workspace: the C ABI's plain entry points
    for name, got in (("O", o.detach()), ("dQ", q.grad), ("dK", k.grad), ("dV", v.grad)):
        ours, theirs = fo.rel_fro(gt[name], got.cpu()), fo.rel_fro(gt[name], peer[name])
        assert ours < max(2.5 * theirs, 6e-3), (name, ours, theirs)
    for name, got in (("dK", k.grad), ("dV", v.grad)):   # and it is the workspace that does it
        assert fo.rel_fro(gt[name], got.cpu()) < 0.6 * fo.rel_fro(gt[name], raw[name]), name
    assert torch.equal(o.detach().cpu(), raw["O"]) and torch.equal(q.grad.cpu(), raw["dQ"])
    # the same bound through the C entry points INTEGRATION.md section B tells a maintainer to bind
    # (fa_bwd_dq_ex / fa_bwd_dkv_ex with q_scaled, raw ctypes) -- bit for bit what flash_attention() computed
    bound = run_gpu_raw(Q, K, V, dO, causal, workspace=True)
    for name, got in (("dQ", q.grad), ("dK", k.grad), ("dV", v.grad)):
        assert torch.equal(bound[name], got.cpu()), name
        assert fo.rel_fro(gt[name], bound[name]) < max(2.5 * fo.rel_fro(gt[name], peer[name]), 6e-3), name


# ---------------------------------------------------------------- (3) torch SDPA on the device
@pytest.mark.parametrize("dtype", [F16, BF16])
def test_compare_with_sdpa_on_device(dtype):
    """compare_with_sdpa (M:172-212) at the reference's __main__ shape."""
    M = _host()
    torch.manual_seed(0)
    Q, K, V = (torch.randn(4, 8, 256, 64, dtype=dtype, device="cuda") for _ in range(3))
    res = M.compare_with_sdpa(Q, K, V, is_causal=True, verbose=False)
    for name, m in res.items():
        assert m["cos"] > 0.999, name
        if dtype == F16:
            assert m["passed"], (name, m)


# ---------------------------------------------------------------- (4) full BASELINE sizes: properties
def _full(dtype=BF16, D=64):
    import _scaling as sc
    return sc.make_shard(0, 4, 32, 4096, 4096, D, dtype, torch.device("cuda"))


def test_full_size_matches_device_sdpa_and_is_deterministic(impl):
    """B=4,H=32,N=4096,D=64 causal bf16 (BASELINE configs[1],[2]) against torch SDPA on the GPU."""
    M = _host()
    Q, K, V, dO = _full()
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention(q, k, v, True)
    o.backward(dO)
    q2, k2, v2 = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o2 = torch.nn.functional.scaled_dot_product_attention(q2, k2, v2, is_causal=True)
    o2.backward(dO)
    for name, a, b in (("O", o, o2), ("dQ", q.grad, q2.grad), ("dK", k.grad, k2.grad), ("dV", v.grad, v2.grad)):
        err = ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert err < 6e-3, (name, err)   # two bf16 implementations, each ~2-3e-3 from exact
    # no atomics anywhere: bit-identical on a second run
    q3, k3, v3 = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o3 = M.flash_attention(q3, k3, v3, True)
    o3.backward(dO)
    assert torch.equal(o, o3) and torch.equal(q.grad, q3.grad) and torch.equal(k.grad, k3.grad) and torch.equal(v.grad, v3.grad)


def test_full_size_properties():
    M = _host()
    Q, K, V, dO = _full()
    O, LSE = M.flash_attention_forward(Q, K, V, True)
    # rows of P sum to one: V = 1 -> O = 1 (up to bf16 rounding of P)
    O1, _ = M.flash_attention_forward(Q, K, torch.ones_like(V), True)
    assert (O1.float() - 1).abs().max() < 2e-2
    # causal row 0 attends to key 0 only: O[0] = V[0], LSE[0] = q0.k0/sqrt(D)
    assert torch.equal(O[:, :, 0], V[:, :, 0])
    s00 = (Q[:, :, 0].float() * K[:, :, 0].float()).sum(-1) / 8.0
    # bf16 inputs: the score operand carries the folded scale (one more 2^-9 rounding, fa_common.h kFoldScale)
    assert ((LSE[:, :, 0] - s00).abs() <= 4e-3 + 4e-3 * s00.abs()).all()
    # (batch, head) slices are independent: permuting them permutes the outputs bit-exactly
    perm = torch.randperm(32, device="cuda")
    Op, LSEp = M.flash_attention_forward(Q[:, perm].contiguous(), K[:, perm].contiguous(), V[:, perm].contiguous(), True)
    nO, nL = int((Op != O[:, perm]).sum()), int((LSEp != LSE[:, perm]).sum())
    assert nO == 0 and nL == 0, ("head permutation changed %d O and %d LSE elements; first: %s %s" % (
        nO, nL, (Op != O[:, perm]).nonzero()[:4].tolist(), (LSEp != LSE[:, perm]).nonzero()[:4].tolist()))
    # linear in V
    V2 = torch.randn_like(V)
    O2, _ = M.flash_attention_forward(Q, K, V2, True)
    O12, _ = M.flash_attention_forward(Q, K, (V.float() + V2.float()).to(V.dtype), True)
    err = ((O12.float() - (O.float() + O2.float())).norm() / O12.float().norm()).item()
    assert err < 1e-2
    # backward: zero upstream gradient -> zero gradients; dV with dO = 1 has column sums = S (sum_q P = ...)
    dQ, dK, dV = M.flash_attention_backward(Q, K, V, O, torch.zeros_like(dO), LSE, True)
    assert dQ.abs().max() == 0 and dK.abs().max() == 0 and dV.abs().max() == 0
    # sum over keys of dV[., d] for dO = 1 equals the number of query rows (each row of P sums to 1)
    dQ, dK, dV = M.flash_attention_backward(Q, K, V, O, torch.ones_like(dO), LSE, True)
    tot = dV.float().sum(dim=2)
    assert (tot / 4096 - 1).abs().max() < 1e-2


def test_full_size_d128_matches_device_sdpa():
    """B=4,H=32,N=4096,D=128 causal bf16 (BASELINE configs[3])."""
    M = _host()
    Q, K, V, dO = _full(D=128)
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o = M.flash_attention(q, k, v, True)
    o.backward(dO)
    q2, k2, v2 = (x.clone().requires_grad_(True) for x in (Q, K, V))
    o2 = torch.nn.functional.scaled_dot_product_attention(q2, k2, v2, is_causal=True)
    o2.backward(dO)
    for name, a, b in (("O", o, o2), ("dQ", q.grad, q2.grad), ("dK", k.grad, k2.grad), ("dV", v.grad, v2.grad)):
        err = ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert err < 6e-3, (name, err)


def test_config5_shard_shape_runs():
    """One rank's shard of BASELINE configs[4] at 8 GPUs: B=8,H=32,N=8192,D=64 causal bf16 (256 MiB per tensor)."""
    import _scaling as sc
    M = _host()
    Q, K, V, dO = sc.make_shard(0, 8, 32, 8192, 8192, 64, BF16, torch.device("cuda"))
    O, LSE = M.flash_attention_forward(Q, K, V, True)
    dQ, dK, dV = M.flash_attention_backward(Q, K, V, O, dO, LSE, True)
    torch.cuda.synchronize()
    # spot-check one (b, h) slice against fp64 on the CPU
    b, h = 5, 17
    sl = lambda x: x[b:b + 1, h:h + 1].cpu()
    gt = fo.attention_fp64(sl(Q), sl(K), sl(V), sl(dO), True)
    for name, t in (("O", O), ("dQ", dQ), ("dK", dK), ("dV", dV)):
        assert fo.rel_fro(gt[name], sl(t)) < 8e-3, name
    assert torch.isfinite(O.float()).all() and torch.isfinite(dK.float()).all()


# ---------------------------------------------------------------- error behaviour
def test_errors_raise():
    M = _host()
    x = torch.randn(1, 1, 16, 96, dtype=F16, device="cuda")
    with pytest.raises(AssertionError):
        M.flash_attention(x, x, x)                       # head dim 96
    y = torch.randn(1, 1, 16, 64, dtype=torch.float32, device="cuda")
    with pytest.raises(AssertionError):
        M.flash_attention(y, y, y)                       # dtype (M:134)
    with pytest.raises(RuntimeError, match="head dim"):
        M.flash_attention_forward(x, x, x, False)        # the C ABI rejects it too, before any launch


# ---------------------------------------------------------------- (5) strided inputs
@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("causal", [False, True], ids=["full", "causal"])
@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_strided_views_are_read_in_place_and_bit_identical(D, causal, dtype, impl):
    """Q/K/V as transposed views of one fused [B, S, 3, H, D] projection output, dO as a view of a [B, S, H, D]
    buffer: the kernels read them in place (the reference copies, M:138-140,156) and must give exactly the bits
    they give on contiguous copies of the same data -- same arithmetic, different addressing."""
    M = _host()
    import _mi355fa as fa
    if impl == 2:   # strided views take family 1 for dQ (the family-2 forward reads views): pin it for the contiguous twin
        fa.lib.fa_debug_force_impl(2, 1, 2)
    B, H, Sq = 2, 3, 333
    torch.manual_seed(3)
    qkv = torch.randn(B, Sq, 3, H, D, device="cuda", dtype=dtype)
    Qv, Kv, Vv = (qkv[:, :, i].transpose(1, 2) for i in range(3))            # [B, H, S, D] views
    dOv = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype).transpose(1, 2)
    for t in (Qv, Kv, Vv, dOv):
        assert not t.is_contiguous() and fa.strided_ok(t)
    # launcher level: views straight in
    O1, L1 = M.flash_attention_forward(Qv, Kv, Vv, causal)
    O2, L2 = M.flash_attention_forward(Qv.contiguous(), Kv.contiguous(), Vv.contiguous(), causal)
    assert torch.equal(O1, O2) and torch.equal(L1, L2)
    g1 = M.flash_attention_backward(Qv, Kv, Vv, O1, dOv, L1, causal)
    g2 = M.flash_attention_backward(Qv.contiguous(), Kv.contiguous(), Vv.contiguous(), O2, dOv.contiguous(), L2, causal)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    # outputs come back in their input's memory order (what empty_like gives a view, M:24,71-73): dense [B, S, H, D]
    for t in (O1,) + tuple(g1):
        assert t.shape == Qv.shape and t.transpose(1, 2).is_contiguous()
    for t in (O2,) + tuple(g2):
        assert t.is_contiguous()
    # autograd level: gradients flow back to the fused buffer; no .contiguous() copy of the views is saved
    x = qkv.clone().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    o = M.FlashAttentionFunction.apply(q, k, v, causal)              # the Python class exposes what it saved
    saved = o.grad_fn.saved_tensors
    assert saved[0].data_ptr() == q.data_ptr() and saved[1].data_ptr() == k.data_ptr()   # views, not copies
    o.backward(dOv)
    ref = torch.stack([g.transpose(1, 2) for g in g1], dim=2)                # [B, S, 3, H, D]
    assert torch.equal(x.grad, ref)
    # the C++ autograd function behind flash_attention(): same gradients, and no 3 x |Q| of copies either
    x2 = qkv.clone().requires_grad_(True)
    q2, k2, v2 = (x2[:, :, i].transpose(1, 2) for i in range(3))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    o2 = M.flash_attention(q2, k2, v2, causal)
    grown = torch.cuda.memory_allocated() - before
    assert grown < 1.5 * (o2.numel() * o2.element_size() + 4 * B * H * Sq) + (1 << 20), grown   # O + LSE only
    o2.backward(dOv)
    assert torch.equal(x2.grad, ref)
    # and against the fp64 oracle, like every other case
    gt = fo.attention_fp64(Qv.cpu(), Kv.cpu(), Vv.cpu(), dOv.cpu(), causal)
    tol = 1e-3 if dtype == F16 else 5e-3
    for name, got in zip(("O", "dQ", "dK", "dV"), (O1,) + tuple(g1)):
        assert fo.rel_fro(gt[name], got.cpu()) < tol, name


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_bshd_model_never_transposes_or_copies(dtype):
    """A model that keeps [B, S, H, D] activations: q = x.transpose(1, 2) in, o.transpose(1, 2).reshape(B, S, H*D) out.
    O is written in q's memory order, so the reshape is a view of O; the gradients are written in their inputs' order, so
    autograd hands them to the [B, S, H, D] leaves as they are.  Bits equal the contiguous launch; an expanded K/V head
    (zero head stride: not a layout an output can take) gets the contiguous gradient and autograd's sum over heads."""
    M = _host()
    B, H, S, D = 2, 4, 384, 64
    torch.manual_seed(11)
    xs = [torch.randn(B, S, H, D, device="cuda", dtype=dtype) for _ in range(3)]
    dY = torch.randn(B, S, H * D, device="cuda", dtype=dtype)
    leaves = [x.clone().requires_grad_(True) for x in xs]
    q, k, v = (x.transpose(1, 2) for x in leaves)
    o = M.flash_attention(q, k, v, True)
    y = o.transpose(1, 2).reshape(B, S, H * D)
    assert y.data_ptr() == o.data_ptr() and y.is_contiguous()              # a view: nothing was transposed
    y.backward(dY)
    twins = [x.transpose(1, 2).contiguous().requires_grad_(True) for x in xs]
    o2 = M.flash_attention(*twins, True)
    assert o2.is_contiguous() and torch.equal(o, o2)
    o2.backward(dY.view(B, S, H, D).transpose(1, 2).contiguous())
    for x, t in zip(leaves, twins):
        assert x.grad.is_contiguous() and torch.equal(x.grad, t.grad.transpose(1, 2))
    # grouped-query style: one K/V head expanded over the query heads
    kv = [torch.randn(B, S, 1, D, device="cuda", dtype=dtype).requires_grad_(True) for _ in range(2)]
    ql = xs[0].clone().requires_grad_(True)
    o3 = M.flash_attention(ql.transpose(1, 2), *(t.transpose(1, 2).expand(B, H, S, D) for t in kv), True)
    o3.transpose(1, 2).reshape(B, S, H * D).backward(dY)
    kc = [t.detach().transpose(1, 2).expand(B, H, S, D).contiguous().requires_grad_(True) for t in kv]
    qc = xs[0].transpose(1, 2).contiguous().requires_grad_(True)
    o4 = M.flash_attention(qc, *kc, True)
    o4.backward(dY.view(B, S, H, D).transpose(1, 2).contiguous())
    assert torch.equal(o3, o4) and torch.equal(ql.grad, qc.grad.transpose(1, 2))
    for t, c in zip(kv, kc):
        want = c.grad.float().sum(1, keepdim=True).transpose(1, 2)           # [B, S, 1, D]
        assert torch.allclose(t.grad.float(), want, rtol=2e-2, atol=2e-2)


def test_layouts_the_kernels_cannot_address_are_copied_not_misread():
    """Unit head-dim stride missing, or K and V with different sequence strides: the host copies (like the
    reference); results equal the contiguous ones."""
    M = _host()
    torch.manual_seed(4)
    B, H, S, D = 1, 2, 200, 64
    Q = torch.randn(B, H, S, D, device="cuda", dtype=F16)
    K = torch.randn(B, S, H, D, device="cuda", dtype=F16).transpose(1, 2)            # bshd view
    V = torch.randn(B, H, S, D, device="cuda", dtype=F16)                             # contiguous: strides differ from K
    Qt = torch.randn(B, H, D, S, device="cuda", dtype=F16).transpose(2, 3)            # head dim not unit stride
    for q in (Q, Qt):
        o = M.flash_attention(q, K, V, True)
        o2 = M.flash_attention(q.contiguous(), K.contiguous(), V.contiguous(), True)
        assert torch.equal(o, o2)


def test_step_is_hip_graph_capturable():
    """The launchers only enqueue on the caller's stream (no allocation, no synchronisation, no host-side state):
    a whole fwd+bwd step can be captured into a hipGraph and replayed with identical results."""
    M = _host()
    torch.manual_seed(11)
    Q, K, V, dO = (torch.randn(2, 4, 512, 64, device="cuda", dtype=BF16) for _ in range(4))
    for t in (Q, K, V):
        t.requires_grad_(True)
    out = {}

    def step():
        o = M.flash_attention(Q, K, V, True)
        o.backward(dO)
        out["o"], out["dq"], out["dk"], out["dv"] = o.detach(), Q.grad, K.grad, V.grad
        Q.grad = None
        K.grad = None
        V.grad = None

    step()
    eager = {k: v.clone() for k, v in out.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    captured = dict(out)                      # static output tensors of the captured step
    for t in captured.values():
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(eager[k], captured[k]), k


@pytest.mark.parametrize("dtype", [F16, BF16], ids=["fp16", "bf16"])
def test_expanded_kv_heads_are_read_in_place(dtype):
    """One K/V head shared by all query heads (MQA-style `expand`, head stride 0): read in place, bit-identical to the
    materialised copy; autograd sums the per-head dK / dV back into the shared head."""
    M = _host()
    import _mi355fa as fa
    torch.manual_seed(8)
    B, H, S, D = 2, 4, 300, 64
    q = torch.randn(B, H, S, D, device="cuda", dtype=dtype, requires_grad=True)
    k1 = torch.randn(B, 1, S, D, device="cuda", dtype=dtype, requires_grad=True)
    v1 = torch.randn(B, 1, S, D, device="cuda", dtype=dtype, requires_grad=True)
    dO = torch.randn(B, H, S, D, device="cuda", dtype=dtype)
    k, v = k1.expand(B, H, S, D), v1.expand(B, H, S, D)
    assert k.stride(1) == 0 and fa.strided_ok(k) and fa.strided_ok(v)
    op = M.FlashAttentionFunction.apply(q, k, v, True)                      # the Python class exposes what it saved
    assert op.grad_fn.saved_tensors[1].data_ptr() == k1.data_ptr()         # the expanded view, not a copy
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    o = M.flash_attention(q, k, v, True)                                   # C++ autograd function: no K/V copy either
    assert torch.cuda.memory_allocated() - before < 1.5 * (o.numel() * o.element_size() + 4 * B * H * S) + (1 << 20)
    assert torch.equal(o, op)
    o.backward(dO)
    q2 = q.detach().clone().requires_grad_(True)
    k2 = k.detach().contiguous().requires_grad_(True)
    v2 = v.detach().contiguous().requires_grad_(True)
    o2 = M.flash_attention(q2, k2, v2, True)
    o2.backward(dO)
    assert torch.equal(o, o2) and torch.equal(q.grad, q2.grad)
    # the shared head receives the sum over query heads (fp32 sum of the 16-bit per-head gradients, rounded once)
    for g1, g2 in ((k1.grad, k2.grad), (v1.grad, v2.grad)):
        ref = g2.float().sum(dim=1, keepdim=True)
        assert (g1.float() - ref).abs().max() <= 2e-2 * ref.abs().max()


# ---------------------------------------------------------------- benchmark counterpart (a11 / a12, P:9-144)
@pytest.mark.parametrize("provider", ["triton", "pytorch", "naive"])
@pytest.mark.parametrize("mode", ["fwd", "fwd_bwd", "bwd"])
def test_benchmark_attention_counterpart(provider, mode):
    """benchmark_attention for every provider x mode on a small shape: returns (ms, tflops) with
    tflops * ms == counted FLOPs (P:101-107; the FLOP KATs of tests/golden/kat.json), timing() really runs
    warmup + repeat calls, and the 'pytorch' provider records which SDPA backend ran (P:53-57 pins FLASH)."""
    from _util import load_kat
    import Performance_Comparison as P
    dev = torch.device("cuda", torch.cuda.current_device())
    c = load_kat()["flops"][3]                        # B2 H4 S256 D64 non-causal (BASELINE configs[0])
    ms, tf = P.benchmark_attention(provider, mode, c["B"], c["H"], c["S"], c["S"], c["D"], c["causal"], dev,
                                   warmup=2, repeat=5)
    assert ms > 0 and tf > 0
    assert abs(tf * 1e12 * ms * 1e-3 - c[mode]) <= 1e-6 * c[mode]
    ms_c, tf_c = P.benchmark_attention(provider, "fwd", 1, 2, 256, 256, 64, True, dev, warmup=1, repeat=2)
    assert abs(tf_c * 1e12 * ms_c * 1e-3 - 4 * 1 * 2 * 256 * 256 * 64 // 2) <= 1e-6 * 4 * 2 * 256 * 256 * 64
    if provider == "pytorch":
        b = P.last_sdpa_backend()
        assert b == "flash" or b.startswith("default ("), b
    calls = []
    P.timing(lambda: calls.append(1), 3, 4)
    assert len(calls) == 7


def test_benchmark_bf16_and_naive_matches_kernels():
    """dtype=bf16 (added argument) runs, and the package's naive_attention agrees with the kernels on device."""
    import Performance_Comparison as P
    M = _host()
    dev = torch.device("cuda", torch.cuda.current_device())
    ms, tf = P.benchmark_attention("triton", "fwd_bwd", 1, 2, 512, 512, 64, True, dev, warmup=1, repeat=2, dtype=BF16)
    assert ms > 0 and tf > 0
    Q, K, V, _ = (x.cuda() for x in rand_inputs(1, 2, 256, 256, 64, F16, seed=3))
    for causal in (False, True):
        ref = P.naive_attention(Q.float(), K.float(), V.float(), causal)
        got = M.flash_attention(Q, K, V, causal)
        assert fo.rel_fro(ref.cpu(), got.cpu()) < 1e-3
