"""Shared helpers for the tests (data loading, seeded inputs)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    B, H, Sq, Sk, D, causal, BM, BN = (int(x) for x in z["meta"])
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    t["meta"] = dict(B=B, H=H, Sq=Sq, Sk=Sk, D=D, causal=bool(causal), BM=BM, BN=BN)
    return t


def load_kat():
    with open(os.path.join(GOLDEN, "kat.json")) as fh:
        return json.load(fh)


def rand_inputs(B, H, Sq, Sk, D, dtype, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32).to(dtype).to(device)
    return mk(B, H, Sq, D), mk(B, H, Sk, D), mk(B, H, Sk, D), mk(B, H, Sq, D)
