import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
torch.manual_seed(3)
B, H, Sq, D = 2, 3, 333, 64
dtype = torch.float16
for causal in (False, True):
  for which in ("all", "q", "k", "v", "do"):
    qkv = torch.randn(B, Sq, 3, H, D, device="cuda", dtype=dtype)
    Qv, Kv, Vv = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    dOv = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype).transpose(1, 2)
    Qc, Kc, Vc, dOc = Qv.contiguous(), Kv.contiguous(), Vv.contiguous(), dOv.contiguous()
    O, L = M.flash_attention_forward(Qc, Kc, Vc, causal)
    ref = M.flash_attention_backward(Qc, Kc, Vc, O, dOc, L, causal)
    args = dict(q=Qc, k=Kc, v=Vc, do=dOc)
    if which == "all": args = dict(q=Qv, k=Kv, v=Vv, do=dOv)
    elif which == "q": args["q"] = Qv
    elif which == "k": args["k"], args["v"] = Kv, Vv
    elif which == "v": args["k"], args["v"] = Kv, Vv
    elif which == "do": args["do"] = dOv
    got = M.flash_attention_backward(args["q"], args["k"], args["v"], O, args["do"], L, causal)
    msg = []
    for n, a, b in zip(("dQ", "dK", "dV"), got, ref):
        d = (a.float() - b.float()).abs()
        bad = (d > 0).nonzero()
        msg.append("%s: %s" % (n, "ok" if len(bad) == 0 else "BAD n=%d max=%.3g rows[%d..%d] first %s" % (len(bad), d.max().item(), bad[:, 2].min().item(), bad[:, 2].max().item(), bad[0].tolist())))
    print("causal=%d strided=%-3s  " % (causal, which) + " | ".join(msg))
