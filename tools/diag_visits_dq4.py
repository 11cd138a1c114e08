import ctypes, os, sys
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch
import _mi355fa as host
lib = ctypes.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "ab/stamps.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
B, H, S, D = 4, 32, 4096, 64
torch.manual_seed(0)
Q, K, V, dO = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
O = torch.empty_like(Q); LSE = torch.empty(B, H, S, device="cuda")
dQ, delta = torch.empty_like(Q), torch.empty_like(LSE)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
lib.fa_debug_force_impl(1, 4, 4)
lib.fa_fwd(P(Q), P(K), P(V), P(O), P(LSE), B, H, S, S, D, 1, 1, D ** -0.5, st)
nwg = 1024
dbg = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device="cuda")
for i in range(6):
    lib.fa_debug_set_buffer(dbg.data_ptr())
    lib.fa_bwd_dq(P(Q), P(K), P(V), P(O), P(dO), P(LSE), P(dQ), P(delta), B, H, S, S, D, 1, 1, D ** -0.5, st)
torch.cuda.synchronize()
d = dbg.cpu().view(nwg, 4, 32).double()[:256]
passes = d[:, :, 20]
for w in range(4):
    p = passes[:, w].sum()
    s12, s13, s14, s2 = (d[:, w, i].sum() / p for i in (12, 13, 14, 2))
    print("wave %d: phase entry (landing wait, barrier, descriptors) %5.0f; whole phase %5.0f cycles per pass" % (w, d[:, w, 11].sum() / p, sum(d[:, w, i].sum() for i in (11, 12, 13, 14, 2)) / p))
    print("wave %d: below-both %5.0f cycles for %d visits (%s per visit); key block w pair %5.0f (2 visits: %.0f each); solo %5.0f for %d visits (%s per visit); last diagonal visit + rest %5.0f"
          % (w, s12, 2 * w, "%.0f" % (s12 / (2 * w)) if w else "-", s13, s13 / 2, s14, 6 - 2 * w, "%.0f" % (s14 / (6 - 2 * w)) if w < 3 else "-", s2))
