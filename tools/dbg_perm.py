import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))
import My_FlashAttention_optimized as M
torch.manual_seed(0)
B, H, S, D = 4, 32, 4096, 64
Q, K, V = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
O, LSE = M.flash_attention_forward(Q, K, V, True)
O2, LSE2 = M.flash_attention_forward(Q, K, V, True)
print("run-to-run equal:", torch.equal(O, O2), torch.equal(LSE, LSE2))
perm = torch.randperm(32, device="cuda")
Op, LSEp = M.flash_attention_forward(Q[:, perm].contiguous(), K[:, perm].contiguous(), V[:, perm].contiguous(), True)
dO = (Op.float() - O[:, perm].float()).abs()
dL = (LSEp - LSE[:, perm]).abs()
print("perm: O equal", torch.equal(Op, O[:, perm]), "LSE equal", torch.equal(LSEp, LSE[:, perm]))
print("max dO", dO.max().item(), "n diff", int((dO > 0).sum()), "max dL", dL.max().item(), "n diff", int((dL > 0).sum()))
idx = (dL > 0).nonzero()
print(idx[:20].tolist())
rows = idx[:, 2]
print("rows min/max", rows.min().item() if len(rows) else None, rows.max().item() if len(rows) else None)
print("row mod 128 histogram (first 16 bins of 8):", torch.bincount((rows % 128) // 8, minlength=16).tolist() if len(rows) else None)
import _scaling as sc
Q, K, V, dO = sc.make_shard(0, 4, 32, 4096, 4096, 64, torch.bfloat16, torch.device("cuda"))
O, LSE = M.flash_attention_forward(Q, K, V, True)
bad = 0
for it in range(40):
    if it % 2 == 0:
        O1, _ = M.flash_attention_forward(Q, K, torch.ones_like(V), True)
    perm = torch.randperm(32, device="cuda")
    Op, LSEp = M.flash_attention_forward(Q[:, perm].contiguous(), K[:, perm].contiguous(), V[:, perm].contiguous(), True)
    okO, okL = torch.equal(Op, O[:, perm]), torch.equal(LSEp, LSE[:, perm])
    if not (okO and okL):
        bad += 1
        dO_ = (Op.float() - O[:, perm].float()).abs(); dL = (LSEp - LSE[:, perm]).abs()
        io = (dO_ > 0).nonzero(); il = (dL > 0).nonzero()
        print("iter", it, "O diff", len(io), "max", dO_.max().item(), "LSE diff", len(il), "max", dL.max().item())
        if len(io): print("  O idx sample", io[:6].tolist(), "rows", io[:, 2].min().item(), io[:, 2].max().item())
        if len(il): print("  L idx sample", il[:6].tolist())
print("bad iterations:", bad, "of 40")
print("--- interleaved with backward launches and other shapes")
bad = 0
Qs, Ks, Vs = (torch.randn(1, 2, 500, 64, device="cuda", dtype=torch.float16) for _ in range(3))
for it in range(80):
    dQ, dK, dV = M.flash_attention_backward(Q, K, V, O, dO, LSE, True)
    M.flash_attention_forward(Qs, Ks, Vs, it % 2 == 0)
    perm = torch.randperm(32, device="cuda")
    Op, LSEp = M.flash_attention_forward(Q[:, perm].contiguous(), K[:, perm].contiguous(), V[:, perm].contiguous(), True)
    okO, okL = torch.equal(Op, O[:, perm]), torch.equal(LSEp, LSE[:, perm])
    if not (okO and okL):
        bad += 1
        dO_ = (Op.float() - O[:, perm].float()).abs(); dL = (LSEp - LSE[:, perm]).abs()
        io = (dO_ > 0).nonzero(); il = (dL > 0).nonzero()
        print("iter", it, "O diff", len(io), "max", dO_.max().item(), "LSE diff", len(il), "max", dL.max().item())
        if len(io): print("  O idx sample", io[:6].tolist(), "rows", io[:, 2].min().item(), io[:, 2].max().item(), "heads", sorted(set(io[:, 1].tolist()))[:8])
        if len(il): print("  L idx sample", il[:6].tolist())
print("bad iterations:", bad, "of 80")
