#!/usr/bin/env python3
"""Bitwise comparison of the three kernels' outputs between two builds of libmi355fa.so (A/B variants that must not
change results).  usage: check_libs.py ab/a.so ab/b.so [--force F,Q,K]   (schedule families forced in both, 0 = the table)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch  # noqa: E402

import _mi355fa as host  # noqa: E402


def load(path):
    lib = ctypes.CDLL(os.path.join(ROOT, path))
    for name, (res, args) in host.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


A, B_ = load(sys.argv[1]), load(sys.argv[2])
if "--force" in sys.argv:
    fam = [int(x) for x in sys.argv[sys.argv.index("--force") + 1].split(",")]
    for lib in (A, B_):
        lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
        lib.fa_debug_force_impl(*fam)
P = lambda t: t.data_ptr()
bad = 0
for (B, H, Sq, Sk, D) in [(1, 2, 128, 128, 64), (2, 2, 320, 320, 64), (1, 2, 500, 500, 64), (1, 2, 333, 777, 64), (1, 2, 777, 333, 64),
                          (2, 3, 1024, 1024, 64), (4, 32, 4096, 4096, 64), (3, 5, 1280, 1280, 64), (1, 2, 640, 640, 128), (2, 16, 2048, 2048, 128),
                          (3, 7, 768, 1024, 128)]:
    for dt, code in ((torch.bfloat16, 1), (torch.float16, 0)):
        for causal in (0, 1):
            torch.manual_seed(Sq + Sk)
            Q, dO = (torch.randn(B, H, Sq, D, device="cuda", dtype=dt) for _ in range(2))
            K, V = (torch.randn(B, H, Sk, D, device="cuda", dtype=dt) for _ in range(2))
            st = torch.cuda.current_stream().cuda_stream
            sc = D ** -0.5
            outs = []
            for lib in (A, B_):
                o = torch.full_like(Q, float("nan"))
                lse = torch.full((B, H, Sq), float("nan"), device="cuda")
                dq, dk, dv = torch.full_like(Q, float("nan")), torch.full_like(K, float("nan")), torch.full_like(V, float("nan"))
                delta = torch.full_like(lse, float("nan"))
                assert lib.fa_fwd(P(Q), P(K), P(V), P(o), P(lse), B, H, Sq, Sk, D, code, causal, sc, st) == 0
                assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(o), P(dO), P(lse), P(dq), P(delta), B, H, Sq, Sk, D, code, causal, sc, st) == 0
                assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(lse), P(delta), P(dk), P(dv), B, H, Sq, Sk, D, code, causal, sc, st) == 0
                outs.append((o, lse, dq, delta, dk, dv))
            torch.cuda.synchronize()
            names = ("O", "LSE", "dQ", "delta", "dK", "dV")
            for n, a, b in zip(names, outs[0], outs[1]):
                if not torch.equal(a.view(torch.uint8), b.view(torch.uint8)) or torch.isnan(b.float()).any():
                    bad += 1
                    print("DIFF %s %s causal=%d B%d H%d %dx%d D%d  max|diff| %.3g" % (n, dt, causal, B, H, Sq, Sk, D, (a.float() - b.float()).abs().max().item()))
print("check_libs: %s" % ("ALL BIT-IDENTICAL" if not bad else "%d differences" % bad))
sys.exit(1 if bad else 0)
