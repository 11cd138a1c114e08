#!/usr/bin/env python3
"""Per-kernel micro-benchmark / A-B harness / profiling target.

Launches fa_fwd, fa_bwd_dq, fa_bwd_dkv through the C ABI of one or more builds of libmi355fa.so
on the headline shape.  With several --libs the builds are timed in INTERLEAVED rounds in one
process after a long warm-up (clock ramp), and the median / min per build are printed -- the only
kind of A/B that means anything on a DVFS-limited part.

    python tools/kbench.py [--libs a.so,b.so] [--rounds 7] [--reps 10] [--dim 64] [--dtype bf16]
                           [--seq 4096] [--kernels fwd,dq,dkv] [--non-causal] [--warm-ms 300]

An arm may pin its schedule families: `--libs libmi355fa.so@0,0,2,libmi355fa.so@0,0,3` compares two families of ONE build
(the forced families are set before every launch of that arm).
"""
import argparse
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd")
sys.path.insert(0, PKG)
import torch  # noqa: E402

import _mi355fa as host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", default=host.LIB_PATH)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--warm-ms", type=float, default=300.0)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--seq", type=int, default=4096)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--kernels", default="fwd,dq,dkv")
ap.add_argument("--non-causal", action="store_true")
ap.add_argument("--impl", default="", help="force schedule family per kernel: fwd,dq,dkv (0 = rule), e.g. 1,1,2")
a = ap.parse_args()


def load(spec):
    path, _, impl = spec.partition("@")
    lib = ctypes.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
    for name, (res, args) in host.SIGNATURES.items():
        if not hasattr(lib, name):   # an older A/B build (e.g. before the strided entry points)
            continue
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    forced = [int(x) for x in (impl or a.impl).split(",")] if (impl or a.impl) else None
    return lib, forced


def split_libs(text):   # "a.so@0,0,2,b.so" -> ["a.so@0,0,2", "b.so"]: a comma starts a new arm when a path follows it
    out = []
    for tok in text.split(","):
        if out and tok.strip().lstrip("-").isdigit():
            out[-1] += "," + tok
        else:
            out.append(tok)
    return out


libs = [(os.path.basename(p), load(p)) for p in split_libs(a.libs)]
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
code = host.BF16 if a.dtype == "bf16" else host.FP16
causal = not a.non_causal
torch.manual_seed(0)
B, H, S, D = a.batch, a.heads, a.seq, a.dim
Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(4))
O = torch.empty_like(Q)
LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dQ, dK, dV, delta = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
c, sc = int(causal), D ** -0.5
P = lambda t: t.data_ptr()


def fns(lib_forced):
    lib, forced = lib_forced
    if forced is not None:
        plain = fns((lib, None))

        def pinned(f):
            def go():
                lib.fa_debug_force_impl(*forced)
                return f()
            return go
        return {k: pinned(v) for k, v in plain.items()}
    return {
        "fwd": lambda: lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, code, c, sc, st),
        "dq": lambda: lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, code, c, sc, st),
        "dkv": lambda: lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dK), P(dV), B, H, S, S, D, code, c, sc, st),
    }


table = [(n, fns(l)) for n, l in libs]
table[0][1]["fwd"]()
table[0][1]["dq"]()  # O, LSE, delta valid for every later launch
torch.cuda.synchronize()
F = 4 * B * H * S * S * D // (2 if causal else 1)
alg = {"fwd": 1.0, "dq": 1.5, "dkv": 1.0}
for kname in a.kernels.split(","):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < a.warm_ms:  # clock ramp + caches
        for _, f in table:
            assert f[kname]() == 0
        torch.cuda.synchronize()
    res = {n: [] for n, _ in table}
    for _ in range(a.rounds):
        for n, f in table:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f[kname]()
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / a.reps)
    for n, _ in table:
        med, mn = statistics.median(res[n]), min(res[n])
        print("%-4s %-22s median %.4f ms (min %.4f)  %.1f TFLOPS (alg %.1fF)  %.1f%% of peak" % (
            kname, n, med, mn, alg[kname] * F / med / 1e9, alg[kname], 100 * alg[kname] * F / med / 1e9 / 2516.6), flush=True)
