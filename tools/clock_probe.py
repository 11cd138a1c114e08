#!/usr/bin/env python3
"""Shader clock / power while one kernel runs back to back (diagnostic): launches the kernel in a loop for a few
seconds and samples `rocm-smi` from the same process between synchronisations."""
import ctypes, os, subprocess, sys, time, argparse, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="fwd")
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--non-causal", action="store_true")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--zero", action="store_true", help="all-zero inputs (toggle-free data)")
a = ap.parse_args()
lib = host.lib
B, H, S, D = 4, a.heads, 4096, a.dim
torch.manual_seed(0)
mk = (lambda: torch.zeros(B, H, S, D, device="cuda", dtype=torch.bfloat16)) if a.zero else (lambda: torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16))
Q, K, V, dO = (mk() for _ in range(4))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
dQ, dK, dV, delta = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
c, sc = int(not a.non_causal), D ** -0.5
def fwd(): lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, c, sc, st)
def dq(): lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, c, sc, st)
def dkv(): lib.fa_bwd_dkv(P(Q), P(K), P(V), P(dO), P(LSE), P(delta), P(dK), P(dV), B, H, S, S, D, 1, c, sc, st)
fwd(); dq(); dkv(); torch.cuda.synchronize()
run = {"fwd": fwd, "dq": dq, "dkv": dkv}[a.kernel]
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return "rocm-smi failed: %r" % e
    keep = [l.strip() for l in out.splitlines() if re.search(r"sclk|Power|fclk|mclk", l)]
    return " | ".join(keep)
print("idle:", smi())
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
samples = []
while time.time() - t0 < a.seconds:
    e0.record()
    for _ in range(200): run()
    e1.record()
    # sample while the queue is still full
    samples.append(smi())
    torch.cuda.synchronize()
    n += 200
    ms = e0.elapsed_time(e1) / 200
print("%s: %.4f ms per launch (last batch)" % (a.kernel, ms))
for s_ in samples[-3:]: print("busy:", s_)
