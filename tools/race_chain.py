#!/usr/bin/env python3
"""The whole fwd -> dQ -> dK/dV chain repeated on fixed inputs with fresh output buffers every run (the allocator hands
the previous run's blocks back in another order): every output of every run compared bitwise with the first run's.
usage: race_chain.py [lib.so] [--impl F,Q,K] [--runs N] [--shape B,H,S] [--poison]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
path = libs[0] if libs else "flashattention-from-scratch-with-triton_amd/libmi355fa.so"
runs = int(arg("--runs", "20"))
impl = [int(x) for x in arg("--impl", "0,0,0").split(",")]
B, H, S = (int(x) for x in arg("--shape", "4,32,4096").split(","))
D = 64
lib = ctypes.CDLL(os.path.join(ROOT, path))
for name, (res, args) in host.SIGNATURES.items():
    if hasattr(lib, name):
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
poison = "--poison" in sys.argv
if poison:
    lib.fa_debug_poison.argtypes = [ctypes.c_void_p]
lib.fa_debug_force_impl(*impl)
P = lambda t: t.data_ptr()
bits = lambda a: a.view(torch.int16 if a.dtype != torch.float32 else torch.int32)
names = ("O", "LSE", "dQ", "delta", "dK", "dV")
bad = 0
for dt, code in ((torch.bfloat16, 1), (torch.float16, 0)):
    for causal in (1, 0):
        torch.manual_seed(S)
        Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(4))
        st = torch.cuda.current_stream().cuda_stream
        sc = D ** -0.5
        def run():
            o, dq, dk, dv = (torch.full_like(Q, float("nan")) for _ in range(4))
            lse, delta = (torch.full((B, H, S), float("nan"), device="cuda") for _ in range(2))
            for f in (lambda: lib.fa_fwd(P(Q), P(K), P(V), P(o), P(lse), B, H, S, S, D, code, causal, sc, st),
                      lambda: lib.fa_bwd_dq(P(Q), P(K), P(V), P(o), P(dO), P(lse), P(dq), P(delta), B, H, S, S, D, code, causal, sc, st),
                      lambda: lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(lse), P(delta), P(dk), P(dv), B, H, S, S, D, code, causal, sc, st)):
                if poison:
                    assert lib.fa_debug_poison(st) == 0
                assert f() == 0
            torch.cuda.synchronize()
            return (o, lse, dq, delta, dk, dv)
        first = run()
        counts = dict.fromkeys(names, 0)
        where = {}
        for i in range(runs):
            out = run()
            for n, a, b in zip(names, first, out):
                ne = bits(a) != bits(b)
                if ne.any():
                    counts[n] += 1
                    if n not in where:
                        idx = ne.nonzero()
                        where[n] = "%s: run %d, %d elements, (b, h) %s, rows %d..%d" % (
                            n, i + 1, idx.shape[0], sorted(set((int(x), int(y)) for x, y in idx[:, :2].tolist()))[:4], idx[:, 2].min().item(), idx[:, 2].max().item())
        bad += sum(counts.values())
        print("%-8s causal=%d families %s: runs (of %d) that differ from the first, per output: %s  %s"
              % (str(dt).split(".")[1], causal, [lib.fa_debug_pick(k, D, code, causal, B, H, S, S) for k in range(3)], runs, counts, "; ".join(where.values())))
print("race_chain on %s: %s" % (os.path.basename(path), "clean" if not bad else "%d PROBLEMS" % bad))
sys.exit(1 if bad else 0)
