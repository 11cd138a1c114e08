#!/usr/bin/env python3
"""High-volume repeat of ONE kernel on fixed inputs, every run compared bitwise with the first (a changed result = a race).
usage: race_stress.py {fwd|dq|dkv} FAMILY [lib.so] [--runs N] [--poison] [--with-dq] [--shape B,H,S] [--only bf16|fp16] [--causal 0|1]
--with-dq (dkv only): the dQ launch (table family) that produces delta runs in front of every dK/dV launch."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
kern, fam = sys.argv[1], int(sys.argv[2])
libs = [a for a in sys.argv[3:] if a.endswith(".so")]
path = libs[0] if libs else "flashattention-from-scratch-with-triton_amd/libmi355fa.so"
runs = int(arg("--runs", "300"))
B, H, S = (int(x) for x in arg("--shape", "4,32,4096").split(","))
D = int(arg("--dim", "64"))
poison, with_dq = "--poison" in sys.argv, "--with-dq" in sys.argv
only, only_c = arg("--only", ""), arg("--causal", "")
lib = ctypes.CDLL(os.path.join(ROOT, path))
for name, (res, args) in host.SIGNATURES.items():
    if hasattr(lib, name):
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
if poison:
    lib.fa_debug_poison.argtypes = [ctypes.c_void_p]
P = lambda t: t.data_ptr()
bits = lambda a: a.view(torch.int16 if a.dtype != torch.float32 else torch.int32)
print("GPU: %s, uuid %s" % (torch.cuda.get_device_name(0), getattr(torch.cuda.get_device_properties(0), "uuid", "?")))
bad = 0
for dt, code in ((torch.bfloat16, 1), (torch.float16, 0)):
    if only and only != ("bf16" if code else "fp16"):
        continue
    for causal in (1, 0):
        if only_c and int(only_c) != causal:
            continue
        torch.manual_seed(S)
        Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(4))
        st = torch.cuda.current_stream().cuda_stream
        sc = D ** -0.5
        O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda")
        dq0, delta0 = torch.empty_like(Q), torch.empty_like(LSE)
        lib.fa_debug_force_impl(0, 0, 0)
        assert lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, code, causal, sc, st) == 0
        assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dq0), P(delta0), B, H, S, S, D, code, causal, sc, st) == 0
        lib.fa_debug_force_impl(fam if kern == "fwd" else 0, fam if kern == "dq" else 0, fam if kern == "dkv" else 0)
        if lib.fa_debug_pick({"fwd": 0, "dq": 1, "dkv": 2}[kern], D, code, causal, B, H, S, S) != fam:
            print("%s %s causal=%d: family %d does not take this launch" % (kern, dt, causal, fam)); continue
        a0, a1 = (torch.empty_like(Q), torch.empty_like(LSE)) if kern != "dkv" else (torch.empty_like(K), torch.empty_like(V))
        b0, b1 = torch.empty_like(a0), torch.empty_like(a1)
        dq1, delta1 = torch.empty_like(Q), torch.empty_like(LSE)
        def run(x0, x1):
            x0.fill_(float("nan")); x1.fill_(float("nan"))
            if poison:
                assert lib.fa_debug_poison(st) == 0
            if kern == "fwd":
                assert lib.fa_fwd(P(Q), P(K), P(V), P(x0), P(x1), B, H, S, S, D, code, causal, sc, st) == 0
            elif kern == "dq":
                assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(x0), P(x1), B, H, S, S, D, code, causal, sc, st) == 0
            else:
                d = delta0
                if with_dq:
                    delta1.fill_(float("nan"))
                    assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dq1), P(delta1), B, H, S, S, D, code, causal, sc, st) == 0
                    d = delta1
                assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(d), P(x0), P(x1), B, H, S, S, D, code, causal, sc, st) == 0
        run(a0, a1)
        torch.cuda.synchronize()
        n_bad, notes = 0, []
        t0 = time.time()
        for i in range(runs):
            run(b0, b1)
            ne0, ne1 = bits(a0) != bits(b0), bits(a1) != bits(b1)
            if ne0.any().item() or ne1.any().item():
                n_bad += 1
                if len(notes) < 5:
                    idx = (ne0 if ne0.any().item() else ne1).nonzero()
                    notes.append("run %d: %d el, (b,h) %s rows %d..%d" % (i, idx.shape[0], sorted(set((int(x), int(y)) for x, y in idx[:, :2].tolist()))[:4], idx[:, 2].min().item(), idx[:, 2].max().item()))
        bad += n_bad
        print("%s family %d %-8s causal=%d%s%s: %d of %d runs differ from the first (%.1f s)  %s"
              % (kern, fam, str(dt).split(".")[1], causal, " +poison" if poison else "", " +dq" if with_dq else "", n_bad, runs, time.time() - t0, "; ".join(notes)))
lib.fa_debug_force_impl(0, 0, 0)
print("race_stress %s family %d on %s: %s" % (kern, fam, os.path.basename(path), "clean" if not bad else "%d PROBLEMS" % bad))
sys.exit(1 if bad else 0)
