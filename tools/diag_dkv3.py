#!/usr/bin/env python3
"""Where do two dK/dV families differ?  usage: diag_dkv3.py Sq Sk causal [dtype]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
lib = host.lib
if os.environ.get("DIAG_LIB"):   # an A/B build (tools/build_variant.sh)
    lib = ctypes.CDLL(os.path.join(ROOT, os.environ["DIAG_LIB"]))
    for name, (res, args) in host.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
Sq, Sk, causal = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.float16 if len(sys.argv) > 4 and sys.argv[4] == "fp16" else torch.bfloat16
code = 0 if dt == torch.float16 else 1
B, H, D = 1, 1, 64
torch.manual_seed(1)
Q, dO = (torch.randn(B, H, Sq, D, device="cuda", dtype=dt) for _ in range(2))
K, V = (torch.randn(B, H, Sk, D, device="cuda", dtype=dt) for _ in range(2))
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
sc = D ** -0.5
o, lse = torch.empty_like(Q), torch.empty(B, H, Sq, device="cuda", dtype=torch.float32)
lib.fa_debug_force_impl(0, 0, 0)
assert lib.fa_fwd(P(Q), P(K), P(V), P(o), P(lse), B, H, Sq, Sk, D, code, causal, sc, st) == 0
dq, delta = torch.empty_like(Q), torch.empty_like(lse)
assert lib.fa_bwd_dq(P(Q), P(K), P(V), P(o), P(dO), P(lse), P(dq), P(delta), B, H, Sq, Sk, D, code, causal, sc, st) == 0
outs = {}
for fam in (2, 3, 3, 3, 3):
    lib.fa_debug_force_impl(0, 0, fam)
    dk, dv = torch.full_like(K, float("nan")), torch.full_like(V, float("nan"))
    assert lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(lse), P(delta), P(dk), P(dv), B, H, Sq, Sk, D, code, causal, sc, st) == 0
    torch.cuda.synchronize()
    outs.setdefault(fam, []).append((dk.float().cpu()[0, 0], dv.float().cpu()[0, 0]))
a, b, b2 = outs[2][0], outs[3][0], outs[3][1]
print("deterministic:", [torch.equal(b[0], x[0]) for x in outs[3][1:]], torch.equal(b[1], b2[1]))
for name, x, y in (("dK", a[0], b[0]), ("dV", a[1], b[1])):
    d = (x - y).abs()
    rows = (d.max(dim=1).values > 0).nonzero().flatten().tolist()
    cols = (d.max(dim=0).values > 0).nonzero().flatten().tolist()
    print(name, "max diff", d.max().item(), "n rows differing", len(rows), "rows", rows[:40], "cols", cols[:70])
# which query tile is responsible: recompute dK contributions per q tile in fp32
with torch.no_grad():
    Qf, Kf, Vf, dOf = (t.float().cpu()[0, 0] for t in (Q, K, V, dO))
    S = Qf @ Kf.T * sc
    if causal:
        S = S.masked_fill(torch.arange(Sk)[None, :] > torch.arange(Sq)[:, None], float("-inf"))
    Pm = torch.exp(S - lse.cpu()[0, 0][:, None])
    dP = dOf @ Vf.T
    dS = Pm * (dP - delta.cpu()[0, 0][:, None])
    ref = dS.T @ Qf * sc
    print("fam2 vs fp32 ref", (a[0] - ref).abs().max().item(), " fam3 vs ref", (b[0] - ref).abs().max().item())
    err = b[0] - a[0]
    # project the error onto per-32-row-block contributions: err ~ sum_blocks c_blk * contribution_blk ?
    for blk in range(0, Sq, 32):
        contrib = dS[blk:blk + 32].T @ Qf[blk:blk + 32] * sc
        num = (err * contrib).sum().item()
        den = (contrib * contrib).sum().item()
        print("  q block %4d: projection coeff %+.4f" % (blk, num / max(den, 1e-30)))
