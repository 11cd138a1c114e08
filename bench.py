#!/usr/bin/env python3
"""Headline benchmark of the attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input: the forward kernel
plus the two backward kernels (dQ+delta, then dK/dV) through the flash_attention autograd
binding, exactly the fwd_bwd mode of the reference's benchmark (Performance_Comparison.py:70-77).
Workload at every N: BASELINE.json configs[2] per GPU (B=4,H=32,N=4096,D=64 causal bf16,
fwd+bwd; configs[1] is its forward half and is reported in `fwd_tflops`), i.e. weak scaling by
batch -- the path has no cross-(batch,head) dependency, so there is no data-path collective.
Rank 0 prints ONE JSON line.  TFLOPS use the reference's counted-FLOP convention
(Performance_Comparison.py:101-107): F = 4*B*H*Sq*Sk*D/(2 if causal), fwd = F, fwd+bwd = 3.5 F.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd"))

import torch  # noqa: E402

import _scaling as sc  # noqa: E402

PEAK_TFLOPS = 2516.6  # gfx950 dense bf16/fp16 MFMA: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz


def flops_fwd(B, H, Sq, Sk, D, causal):
    return 4 * B * H * Sq * Sk * D // (2 if causal else 1)


def kernel_times(M, Q, K, V, dO, causal, reps):
    """Average duration (ms) of each of the three kernels, HIP events on the launch stream."""
    O, LSE = M.flash_attention_forward(Q, K, V, causal)
    B, H, Sq, D = Q.shape
    dQ = torch.empty_like(Q)
    dK = torch.empty_like(K)
    dV = torch.empty_like(V)
    delta = torch.empty_like(LSE)
    fa, dt = M._fa, M._DTYPES[Q.dtype]
    Sk = K.shape[2]
    scale = 1 / (D ** 0.5)
    st = torch.cuda.current_stream().cuda_stream
    c = int(causal)
    launches = {
        "fa_fwd": lambda: fa.lib.fa_fwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), LSE.data_ptr(),
                                        B, H, Sq, Sk, D, dt, c, scale, st),
        "fa_bwd_dq": lambda: fa.lib.fa_bwd_dq(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(),
                                              LSE.data_ptr(), dQ.data_ptr(), delta.data_ptr(),
                                              B, H, Sq, Sk, D, dt, c, scale, st),
        "fa_bwd_dkv": lambda: fa.lib.fa_bwd_dkv(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dO.data_ptr(),
                                                LSE.data_ptr(), delta.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                                                B, H, Sq, Sk, D, dt, c, scale, st),
    }
    out = {}
    for name, fn in launches.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            rc = fn()
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, fa.lib.fa_last_error()
        out[name] = a.elapsed_time(b) / reps
    return out


def cpu_baseline(B, H, S, D, causal, dtype):
    """PyTorch CPU SDPA fwd+bwd on this box's host cores (reported baseline, not the target)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fa_oracle as fo
    torch.manual_seed(0)
    Q, K, V, dO = (torch.randn(B, H, S, D, dtype=dtype) for _ in range(4))
    fo.cpu_sdpa(Q, K, V, causal, dO)  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < 8.0 and n < 20):
        fo.cpu_sdpa(Q, K, V, causal, dO)
        n += 1
    sec = (time.perf_counter() - t0) / n
    f = 3.5 * flops_fwd(B, H, S, S, D, causal)
    return {"value": round(f / sec / 1e12, 4), "unit": "TFLOPS", "cores": torch.get_num_threads(),
            "kind": "port", "ms": round(sec * 1e3, 2),
            "sample": "torch CPU scaled_dot_product_attention fwd+bwd, the full workload "
                      "B=%d,H=%d,N=%d,D=%d causal=%s %s, 1 warm-up + %d timed" % (
                          B, H, S, D, causal, str(dtype).split(".")[-1], n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="batch PER GPU")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--non-causal", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", default="bhsd", choices=["bhsd", "bshd"],
                    help="bhsd: contiguous [B,H,S,D] inputs (the BASELINE config); bshd: Q/K/V are transposed views of "
                         "[B,S,H,D] buffers, read in place by the kernels (the reference would copy them)")
    args = ap.parse_args()

    rank, local_rank, world = sc.init()
    assert world == max(1, args.gpus) or world == 1, "launch with torchrun --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path is the only path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import My_FlashAttention_optimized as M  # raises if libmi355fa.so is missing

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    causal = not args.non_causal
    B, H, S, D = args.batch, args.heads, args.seq, args.dim
    lo, hi = rank * B, (rank + 1) * B  # weak scaling: B batches per GPU, global batch = B * world
    Q, K, V, dO = sc.make_shard(lo, hi, H, S, S, D, dtype, dev)
    if args.layout == "bshd":   # same values, stored [B,S,H,D]; the step sees [B,H,S,D] views
        Qb, Kb, Vb = (x.transpose(1, 2).contiguous() for x in (Q, K, V))
        Q, K, V = (x.transpose(1, 2) for x in (Qb, Kb, Vb))
    Q.requires_grad_(True)
    K.requires_grad_(True)
    V.requires_grad_(True)

    def step():  # Performance_Comparison.py:70-77
        O = M.flash_attention(Q, K, V, causal)
        O.backward(dO)
        Q.grad = None
        K.grad = None
        V.grad = None

    def step_fwd():
        with torch.no_grad():
            M.flash_attention(Q, K, V, causal)

    ms_total = sc.timed_steps(step, args.steps, args.warmup, dev)
    ms_step = ms_total / args.steps
    ms_fwd = sc.timed_steps(step_fwd, args.steps, min(args.warmup, 3), dev) / args.steps
    ms_fwd_copy = None
    if args.layout == "bshd":   # what the reference's binding does with such views (M:138-140): copy, then run
        def step_fwd_copy():
            with torch.no_grad():
                M.flash_attention(Q.contiguous(), K.contiguous(), V.contiguous(), causal)
        ms_fwd_copy = sc.timed_steps(step_fwd_copy, args.steps, min(args.warmup, 3), dev) / args.steps

    F = flops_fwd(B, H, S, S, D, causal)  # per GPU
    tf_step = world * 3.5 * F / (ms_step * 1e-3) / 1e12
    tf_fwd = world * F / (ms_fwd * 1e-3) / 1e12

    kt = kernel_times(M, Q.detach(), K.detach(), V.detach(), dO, causal, reps=20)
    # algorithmic FLOPs per launch (DESIGN.md "Measurement"): fwd = F (QK^T, PV); dQ kernel = 1.5 F
    # (S, dP, dQ); dK/dV kernel = 1.0 F (dK, dV; its recomputed S and dP are not credited) -> 3.5 F total
    alg = {"fa_fwd": 1.0 * F, "fa_bwd_dq": 1.5 * F, "fa_bwd_dkv": 1.0 * F}
    # MFMA FLOPs the kernel actually executes (information only): the dK/dV kernel recomputes S and dP (4 GEMMs)
    executed = {"fa_fwd": 1.0 * F, "fa_bwd_dq": 1.5 * F, "fa_bwd_dkv": 2.0 * F}
    kernels = {k: {"ms": round(v, 4), "tflops": round(alg[k] / (v * 1e-3) / 1e12, 1),
                   "frac": round(alg[k] / (v * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
                   "mfma_busy_frac_executed": round(executed[k] / (v * 1e-3) / 1e12 / PEAK_TFLOPS, 4)}
               for k, v in kt.items()}
    dom = max(kt, key=kt.get)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom, {}).get("bytes")  # HBM bytes per launch, PMC (profiles/)
        except Exception:
            traffic = None
    roofline = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"], "peak": PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": kernels[dom]["frac"], "traffic": traffic}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(B, H, S, D, causal, dtype)

    if rank == 0:
        line = {
            "metric": "achieved TFLOPS fwd and fwd+bwd (B=4,H=32,N=4096,D=64 causal); %MFMA peak",
            "value": round(tf_step, 2), "unit": "TFLOPS (fwd+bwd, counted FLOPs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "FlashAttention fwd+bwd, B=%d per GPU,H=%d,N=%d,D=%d %s %s (BASELINE configs[2]; "
                                   "fwd_* fields = configs[1])%s" % (B, H, S, D, "causal" if causal else "non-causal", args.dtype,
                                                                    "" if args.layout == "bhsd" else "; inputs are [B,S,H,D] views read in place"),
                       "global_batch": B * world, "seq_len": S, "parallelism": "batch-sharded x%d, no collective" % world},
            "fwd_bwd_tflops": round(tf_step, 2), "fwd_tflops": round(tf_fwd, 2), "fwd_ms": round(ms_fwd, 4),
            "pct_mfma_peak_fwd_bwd": round(100 * tf_step / world / PEAK_TFLOPS, 2),
            "pct_mfma_peak_fwd": round(100 * tf_fwd / world / PEAK_TFLOPS, 2),
            "kernels": kernels, "roofline": roofline, "cpu_baseline": cpu,
        }
        if ms_fwd_copy is not None:
            line["fwd_ms_if_views_were_copied_first"] = round(ms_fwd_copy, 4)
        print(json.dumps(line), flush=True)
    sc.finalize()


if __name__ == "__main__":
    main()
