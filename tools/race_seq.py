#!/usr/bin/env python3
"""dQ (table family) + dK/dV (forced family) repeated after ONE forward; the first family-4 run follows a family-3 run.
usage: race_seq.py lib.so ... [--runs N]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 10
B, H, S, D = 4, 32, 4096, 64
P = lambda t: t.data_ptr()
bits = lambda a: a.view(torch.int16 if a.dtype != torch.float32 else torch.int32)
for path in libs:
    lib = ctypes.CDLL(os.path.join(ROOT, path))
    for name, (res, args) in host.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
        for causal in (0, 1):
            torch.manual_seed(S)
            Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(4))
            O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            sc = D ** -0.5
            lib.fa_debug_force_impl(0, 0, 0)
            assert lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, code, causal, sc, st) == 0
            def run(fam):
                lib.fa_debug_force_impl(0, 0, fam)
                dq, delta = torch.full_like(Q, float("nan")), torch.full_like(LSE, float("nan"))
                assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dq), P(delta), B, H, S, S, D, code, causal, sc, st) == 0
                dk, dv = torch.full_like(K, float("nan")), torch.full_like(V, float("nan"))
                assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dk), P(dv), B, H, S, S, D, code, causal, sc, st) == 0
                torch.cuda.synchronize()
                return dq, delta, dk, dv
            ref3 = run(3)
            outs = [run(4) for _ in range(runs)]
            names = ("dQ", "delta", "dK", "dV")
            msg = []
            for i in range(1, runs):
                for n, a, b in zip(names, outs[0], outs[i]):
                    ne = bits(a) != bits(b)
                    if ne.any():
                        idx = ne.nonzero()
                        msg.append("run %d %s: %d el, (b,h) %s rows %d..%d" % (i, n, idx.shape[0], sorted(set((int(x), int(y)) for x, y in idx[:, :2].tolist()))[:3], idx[:, 2].min().item(), idx[:, 2].max().item()))
            d3 = [n for n, a, b in zip(names, ref3, outs[-1]) if not torch.equal(bits(a), bits(b))]
            print("%-10s %-8s causal=%d: last run vs family-3 run differs in %s; %s" % (os.path.basename(path), str(dt).split(".")[1], causal, d3 or "nothing", "; ".join(msg[:4]) or "all runs identical"))
