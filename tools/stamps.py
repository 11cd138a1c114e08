#!/usr/bin/env python3
"""Read the per-wave cycle stamps of a -DFA_STAMPS build of the forward kernel (diagnostic only;
a stamped build is slower and its absolute time means nothing -- read the SHARES)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import torch  # noqa: E402

import _mi355fa as host  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "ab/stamps.so"))
for name, (res, args) in host.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
lib.fa_debug_set_buffer.argtypes = [ctypes.c_void_p]
causal = "--non-causal" not in sys.argv
B, H, S, D = 4, 32, 4096, 64
torch.manual_seed(0)
Q, K, V = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
O = torch.empty_like(Q)
LSE = torch.empty(B, H, S, device="cuda", dtype=torch.float32)
nblk = (S // 128) * B * H
dbg = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(5):
    lib.fa_debug_set_buffer(dbg.data_ptr() if i == 4 else None)
    assert lib.fa_fwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), LSE.data_ptr(), B, H, S, S, D, 1, int(causal), D ** -0.5, st) == 0
torch.cuda.synchronize()
d = dbg.cpu().view(nblk, 4, 12).double()
names = ["prefetch issue", "S MFMA + max", "rescale/exp/sum", "pack + PV MFMA", "vmcnt + ds_write", "barrier"]
tiles = d[:, :, 8]
tot = d[:, :, :6].sum()
print("waves %d, tiles/wave mean %.1f" % (d.shape[0] * 4, tiles.mean()))
for i, n in enumerate(names):
    print("  %-18s %5.1f%%   %7.0f cycles per tile" % (n, 100 * d[:, :, i].sum() / tot, d[:, :, i].sum() / tiles.sum()))
print("  loop cycles per tile: %.0f ; epilogue per wave: %.0f cycles ; prologue+loop+epilogue per wave %.0f" % (
    d[:, :, 6].sum() / tiles.sum(), d[:, :, 7].mean(), (d[:, :, 11] - d[:, :, 10]).mean()))
span = (d[:, :, 11].max() - d[:, :, 10].min())
print("  kernel span (s_memtime ticks, 100 MHz?): %.0f" % span)
for w in range(4):
    print("  wave %d: " % w + " ".join("%s=%.0f" % (n.split()[0], d[:, w, i].sum() / tiles[:, w].sum()) for i, n in enumerate(names)))
