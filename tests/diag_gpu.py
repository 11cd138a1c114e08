"""GPU diagnostic (not a pytest file): runs the three kernels on a list of shapes, prints
relative errors against the fp64 oracle and, when something is off, WHERE (by row / column
class) so that a fragment-layout mistake can be read off one run.

    python tests/diag_gpu.py [quick|full] [bench]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402

import fa_oracle as fo  # noqa: E402
import My_FlashAttention_optimized as M  # noqa: E402


def where(name, ref, got):
    """Summarise an error tensor [B,H,S,D] by row-in-128 class and by column."""
    e = (got.double().cpu() - ref.double().cpu()).abs()
    S, D = e.shape[-2:]
    by_col = e.amax(dim=(0, 1, 2))
    by_row = e.amax(dim=(0, 1, 3))
    print("   %s: worst cols %s" % (name, [(int(i), round(float(by_col[i]), 4)) for i in by_col.argsort(descending=True)[:6]]))
    print("   %s: worst rows %s" % (name, [(int(i), round(float(by_row[i]), 4)) for i in by_row.argsort(descending=True)[:8]]))
    bad_rows = (by_row > 10 * by_row.median().clamp_min(1e-6)).nonzero().flatten().tolist()
    print("   %s: #rows >10x median: %d  first %s" % (name, len(bad_rows), bad_rows[:24]))
    bad_cols = (by_col > 10 * by_col.median().clamp_min(1e-6)).nonzero().flatten().tolist()
    print("   %s: #cols >10x median: %d  first %s" % (name, len(bad_cols), bad_cols[:24]))


def run_case(B, H, Sq, Sk, D, causal, dtype, seed=0, verbose_fail=True):
    torch.manual_seed(seed)
    Q = torch.randn(B, H, Sq, D, dtype=dtype)
    K = torch.randn(B, H, Sk, D, dtype=dtype)
    V = torch.randn(B, H, Sk, D, dtype=dtype)
    dO = torch.randn(B, H, Sq, D, dtype=dtype)
    gt = fo.attention_fp64(Q, K, V, dO, causal)
    dev = "cuda"
    Qd, Kd, Vd, dOd = (x.to(dev) for x in (Q, K, V, dO))
    O, LSE = M.flash_attention_forward(Qd, Kd, Vd, causal)
    torch.cuda.synchronize()
    res = {"O": fo.rel_fro(gt["O"], O.cpu()), "LSE": (LSE.cpu().double() - gt["LSE"]).abs().max().item()}
    # backward fed with the ORACLE's O/LSE rounded the way the forward stores them, so that the
    # backward kernels are judged on their own
    O_in = gt["O"].to(dtype).to(dev)
    LSE_in = gt["LSE"].float().to(dev)
    dQ, dK, dV = M.flash_attention_backward(Qd, Kd, Vd, O_in, dOd, LSE_in, causal)
    torch.cuda.synchronize()
    for n, t in (("dQ", dQ), ("dK", dK), ("dV", dV)):
        res[n] = fo.rel_fro(gt[n], t.cpu())
    tol = 1.5e-3 if dtype == torch.float16 else 1.2e-2
    bad = [k for k in ("O", "dQ", "dK", "dV") if not (res[k] < tol)] + (["LSE"] if not (res["LSE"] < (1e-2 if dtype == torch.bfloat16 else 1e-3)) else [])
    tag = "OK " if not bad else "BAD"
    print("%s B%d H%d Sq%d Sk%d D%d %s %s :: " % (tag, B, H, Sq, Sk, D, "causal" if causal else "full  ",
                                                 str(dtype).split(".")[-1]) +
          " ".join("%s=%.2e" % kv for kv in res.items()), flush=True)
    if bad and verbose_fail:
        outs = {"O": O, "dQ": dQ, "dK": dK, "dV": dV}
        for n in bad:
            if n in outs:
                where(n, gt[n], outs[n])
            else:
                e = (LSE.cpu().double() - gt["LSE"]).abs().amax(dim=(0, 1))
                print("   LSE worst rows", [(int(i), round(float(e[i]), 4)) for i in e.argsort(descending=True)[:8]])
    return not bad


def bench(B, H, N, D, causal, dtype, rep=20):
    dev = "cuda"
    torch.manual_seed(0)
    Q, K, V, dO = (torch.randn(B, H, N, D, dtype=dtype, device=dev) for _ in range(4))
    O, LSE = M.flash_attention_forward(Q, K, V, causal)

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(rep):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / rep

    f = fo.attention_flops(B, H, N, N, D, causal)
    tf = t(lambda: M.flash_attention_forward(Q, K, V, causal))
    tb = t(lambda: M.flash_attention_backward(Q, K, V, O, dO, LSE, causal))
    print("BENCH B%d H%d N%d D%d %s %s: fwd %.3f ms %.1f TF | bwd %.3f ms %.1f TF | fwd+bwd %.1f TF" % (
        B, H, N, D, "causal" if causal else "full", str(dtype).split(".")[-1],
        tf, f / tf / 1e9, tb, 2.5 * f / tb / 1e9, 3.5 * f / (tf + tb) / 1e9), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    for a in sys.argv:
        if a.startswith("--impl="):  # force the schedule family per kernel: --impl=fwd,dq,dkv
            import ctypes
            M._fa.lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
            M._fa.lib.fa_debug_force_impl(*[int(x) for x in a.split("=")[1].split(",")])
    print("device:", torch.cuda.get_device_name(0), flush=True)
    ok = True
    bf, hf = torch.bfloat16, torch.float16
    cases = [
        (1, 1, 128, 128, 64, False, hf), (1, 1, 128, 128, 64, True, hf),
        (1, 2, 256, 256, 64, False, bf), (2, 2, 256, 256, 64, True, bf),
        (1, 2, 128, 320, 64, False, hf), (1, 2, 384, 128, 64, True, hf),
        (1, 2, 500, 500, 64, True, hf), (1, 2, 500, 500, 64, False, bf), (1, 1, 77, 333, 64, False, hf),
        (1, 1, 1024, 1024, 64, True, bf),
    ]
    if mode == "full":
        cases += [
            (1, 1, 128, 128, 128, False, hf), (1, 1, 256, 256, 128, True, bf), (1, 2, 500, 500, 128, True, hf),
            (2, 3, 1024, 1024, 64, True, hf), (1, 1, 2048, 2048, 64, False, bf),
        ]
    for c in cases:
        t0 = time.time()
        ok &= run_case(*c)
    print("ALL OK" if ok else "SOME BAD", flush=True)
    if "bench" in sys.argv:
        bench(4, 32, 4096, 64, True, bf)
        bench(4, 32, 4096, 64, True, hf)
        bench(4, 32, 4096, 64, False, bf)
        bench(4, 8, 4096, 64, True, hf)
