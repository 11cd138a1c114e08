#!/usr/bin/env python3
"""Headline benchmark of the attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {3,4,5}]

One "step" = one pass of the hot path over one batch of synthetic input: the forward kernel plus the two backward
kernels (dQ+delta, then dK/dV) through the flash_attention autograd binding, exactly the fwd_bwd mode of the reference's
benchmark (Performance_Comparison.py:70-77).

Workloads (BASELINE.json `configs`):
  --config 3 (default)  B=4 PER GPU, H=32, N=4096, D=64 causal bf16 fwd+bwd (configs[2]; configs[1] is its forward half,
                        reported in the fwd_* fields) -- weak scaling by batch;
  --config 4            the same at D=128 (configs[3]);
  --config 5            GLOBAL B=64, H=32, N=8192, D=64 causal bf16, the batch sharded over the ranks with
                        _scaling.shard_range (configs[4]) -- strong scaling.  Every default run also measures this
                        workload after the headline one and reports it in the `config5` object of the same JSON line.
The path has no cross-(batch, head) dependency, so there is no data-path collective: torch.distributed (nccl = RCCL over
xGMI) only carries the barriers, the MAX-reduce of the elapsed time and a SUM-reduce of shard checksums.

Ranks: with --gpus N > 1 and no WORLD_SIZE in the environment this process starts N child processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) BEFORE anything touches the GPU, waits for them and
exits with their status; under torchrun (WORLD_SIZE set) it is one of the ranks.  A rank count that does not match --gpus,
or fewer visible GPUs than ranks, is an error (non-zero exit), never a silent single-process run.
Rank 0 prints ONE JSON line.  TFLOPS use the reference's counted-FLOP convention (Performance_Comparison.py:101-107):
F = 4*B*H*Sq*Sk*D/(2 if causal), fwd = F, fwd+bwd = 3.5 F.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd")
sys.path.insert(0, PKG)

import torch  # noqa: E402

import _scaling as sc  # noqa: E402

PEAK_TFLOPS = 2516.6  # gfx950 dense bf16/fp16 MFMA: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz

# name -> (batch, heads, seq, dim, scaling); batch is per GPU for "weak", global for "strong"
CONFIGS = {"3": (4, 32, 4096, 64, "weak"), "4": (4, 32, 4096, 128, "weak"), "5": (64, 32, 8192, 64, "strong")}


def flops_fwd(B, H, Sq, Sk, D, causal):
    return 4 * B * H * Sq * Sk * D // (2 if causal else 1)


def kernel_source_hash():
    """sha256 over the kernel sources: ties profiles/pmc_traffic.json to the code it was measured on (there is no .git on
    the GPU box).  tools/pmc_traffic.py writes the same value."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="3", choices=sorted(CONFIGS), help="BASELINE.json workload (see the docstring)")
    ap.add_argument("--batch", type=int, default=None, help="override: batch PER GPU (weak scaling)")
    ap.add_argument("--heads", type=int, default=None)
    ap.add_argument("--seq", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--non-causal", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll-s", type=float, default=0.5,
                    help="seconds of the SAME step run untimed before the --warmup/--steps window (clock settle: the first "
                         "~40 steps after an idle period run in a DVFS transient, first above and then below the steady "
                         "clock); reported as preroll_s; 0 disables")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra config-5 measurement of a default run")
    ap.add_argument("--layout", default="bhsd", choices=["bhsd", "bshd"],
                    help="bhsd: contiguous [B,H,S,D] inputs (the BASELINE config); bshd: Q/K/V are transposed views of "
                         "[B,S,H,D] buffers, read in place by the kernels (the reference would copy them)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="run the N-rank code path with every rank on GPU 0 and gloo as the process group (RCCL refuses two "
                         "ranks on one device): the real kernels, shards, barriers, max-over-ranks and the JSON line of a "
                         "multi-GPU run on a one-GPU box; the line carries \"rehearsal\": true and is not a measurement")
    ap.add_argument("--harness-selftest", action="store_true",
                    help="CPU / gloo: exercise the rank spawn, sharding, barrier and reduce plumbing with a trivial "
                         "torch op as the step; no GPU, no kernels, value = null (tests/test_dist.py)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# rank management
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script (never exec: nothing here has touched the GPU, and the children are
    ordinary child processes), wait for all, return the first non-zero exit status (0 if none)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:   # a dead rank leaves the others in a barrier: stop exactly the processes started here
                    q.terminate()
    return rc


def fail(msg, code=2):
    print("bench.py: " + msg, file=sys.stderr, flush=True)
    sys.exit(code)


# ---------------------------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------------------------
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
    # the backward launches as the autograd path issues them: bf16 passes the scaled-Q workspace (mi355fa_opts.q_scaled)
    import ctypes
    qs = torch.empty_like(Q) if Q.dtype == torch.bfloat16 else None
    opts = fa.Opts.make(q_scaled=qs.data_ptr()) if qs is not None else None
    op = ctypes.byref(opts) if opts is not None else None
    launches = {
        "fa_fwd": lambda: fa.lib.fa_fwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), LSE.data_ptr(),
                                        B, H, Sq, Sk, D, dt, c, scale, st),
        "fa_bwd_dq": lambda: fa.lib.fa_bwd_dq_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(),
                                                 LSE.data_ptr(), dQ.data_ptr(), delta.data_ptr(),
                                                 B, H, Sq, Sk, D, dt, c, scale, op, st),
        "fa_bwd_dkv": lambda: fa.lib.fa_bwd_dkv_ex(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dO.data_ptr(),
                                                   LSE.data_ptr(), delta.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                                                   B, H, Sq, Sk, D, dt, c, scale, op, st),
    }
    out = {}
    for name, fn in launches.items():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:   # clock ramp: a cold start reads up to 20 % slow on this part
            for _ in range(10):
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


def traffic_for(kernel, shape_key):
    """HBM bytes per launch of `kernel` from profiles/pmc_traffic.json, or None when that file was measured on other
    kernel sources or another shape (it is a PMC measurement, not something this run can reproduce live)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None
    if t.get("source_hash") != kernel_source_hash() or t.get("shape") != shape_key:
        return None
    return t.get(kernel, {}).get("bytes")


def measure(M, dev, rank, world, lo, hi, H, S, D, dtype, causal, steps, warmup, layout="bhsd", with_fwd=True, preroll_s=0.0):
    """Timed fwd+bwd steps (and forward-only steps) on this rank's batch shard [lo, hi)."""
    Q, K, V, dO = sc.make_shard(lo, hi, H, S, S, D, dtype, dev)
    if layout == "bshd":   # same values, stored [B,S,H,D]; the step sees [B,H,S,D] views
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

    ms_total = sc.timed_steps(step, steps, warmup, dev, preroll_s)
    res = {"ms_step": ms_total / steps, "tensors": (Q, K, V, dO)}
    if with_fwd:
        res["ms_fwd"] = sc.timed_steps(step_fwd, steps, min(warmup, 3), dev, preroll_s) / steps
    if layout == "bshd":   # what the reference's binding does with such views (M:138-140): copy, then run
        def step_fwd_copy():
            with torch.no_grad():
                M.flash_attention(Q.contiguous(), K.contiguous(), V.contiguous(), causal)
        res["ms_fwd_copy"] = sc.timed_steps(step_fwd_copy, steps, min(warmup, 3), dev) / steps
    # whole-job checksum of one step's outputs (outside the timed region): independent of the number of ranks
    O = M.flash_attention(Q, K, V, causal)
    O.backward(dO)
    res["checksum"] = sc.sum_over_ranks(sc.checksum([O, Q.grad, K.grad, V.grad]), dev)
    Q.grad = K.grad = V.grad = None
    return res


def run_rank(args):
    rank, local_rank, world = sc.env_world()
    if world != max(1, args.gpus):
        fail("--gpus %d but WORLD_SIZE=%d: launch exactly --gpus ranks (python bench.py --gpus N starts them itself)"
             % (args.gpus, world))
    if args.harness_selftest:
        return harness_selftest(args)
    ndev = torch.cuda.device_count()
    if ndev < world and not args.rehearse_one_gpu:
        fail("%d ranks requested but only %d GPU(s) visible" % (world, ndev))
    if not torch.cuda.is_available():
        fail("bench.py needs a GPU (the HIP path is the only path)")
    if args.rehearse_one_gpu:
        rank, local_rank, world = sc.init(backend="gloo")
        local_rank = 0
    else:
        rank, local_rank, world = sc.init()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import My_FlashAttention_optimized as M  # raises if libmi355fa.so is missing

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    causal = not args.non_causal
    cb, ch, cs, cd, scaling = CONFIGS[args.config]
    overridden = any(x is not None for x in (args.batch, args.heads, args.seq, args.dim))
    B = args.batch if args.batch is not None else cb
    H = args.heads if args.heads is not None else ch
    S = args.seq if args.seq is not None else cs
    D = args.dim if args.dim is not None else cd
    if overridden:
        scaling = "weak"
    if scaling == "weak":       # B batches per GPU, global batch = B * world
        lo, hi, GB = rank * B, (rank + 1) * B, B * world
    else:                       # global batch B sharded over the ranks
        (lo, hi), GB = sc.shard_range(B, rank, world), B
        if hi - lo == 0:
            fail("rank %d of %d has an empty shard of global batch %d" % (rank, world, B))

    m = measure(M, dev, rank, world, lo, hi, H, S, D, dtype, causal, args.steps, args.warmup, args.layout,
                preroll_s=args.preroll_s)
    ms_step, ms_fwd = m["ms_step"], m["ms_fwd"]
    F_job = flops_fwd(GB, H, S, S, D, causal)            # whole job
    tf_step = 3.5 * F_job / (ms_step * 1e-3) / 1e12
    tf_fwd = F_job / (ms_fwd * 1e-3) / 1e12

    Q, K, V, dO = m["tensors"]
    F = flops_fwd(hi - lo, H, S, S, D, causal)           # this rank's launch
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
    shape_key = "B%d,H%d,N%d,D%d,%s,%s" % (hi - lo, H, S, D, "causal" if causal else "full", args.dtype)
    roofline = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"], "peak": PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": kernels[dom]["frac"], "traffic": traffic_for(dom, shape_key)}
    del Q, K, V, dO, m["tensors"]

    c5 = None
    if args.config == "3" and not overridden and not args.no_config5 and args.layout == "bhsd":
        # BASELINE configs[4] beside the headline line: GLOBAL B=64, H=32, N=8192, D=64 causal bf16, batch-sharded
        b5, h5, s5, d5, _ = CONFIGS["5"]
        lo5, hi5 = sc.shard_range(b5, rank, world)
        torch.cuda.empty_cache()
        k5 = max(3, min(args.steps, 10))
        m5 = measure(M, dev, rank, world, lo5, hi5, h5, s5, d5, torch.bfloat16, True, k5, 2, with_fwd=False,
                     preroll_s=args.preroll_s)
        f5 = 3.5 * flops_fwd(b5, h5, s5, s5, d5, True)
        c5 = {"workload": "GLOBAL B=%d,H=%d,N=%d,D=%d causal bf16 fwd+bwd, batch-sharded (BASELINE configs[4])" % (b5, h5, s5, d5),
              "scaling": "strong", "n_gpus": world, "batch_per_gpu": hi5 - lo5, "steps": k5,
              "ms_per_step": round(m5["ms_step"], 3), "tflops": round(f5 / (m5["ms_step"] * 1e-3) / 1e12, 2),
              "checksum": m5["checksum"]}
        del m5

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb_B, cb_S = (hi - lo, S) if (hi - lo) * S <= 4 * 4096 else (4, 4096)   # bounded sample of the same workload
        cpu = cpu_baseline(cb_B, H, cb_S, D, causal, dtype)

    if rank == 0:
        mask = "causal" if causal else "non-causal"
        metric = "achieved TFLOPS fwd and fwd+bwd (B=%d,H=%d,N=%d,D=%d %s); %%MFMA peak" % (
            B, H, S, D, mask) if scaling == "weak" else \
            "achieved TFLOPS fwd and fwd+bwd (global B=%d,H=%d,N=%d,D=%d %s, batch-sharded); %%MFMA peak" % (B, H, S, D, mask)
        line = {
            "metric": metric,
            "value": round(tf_step, 2), "unit": "TFLOPS (fwd+bwd, counted FLOPs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_s": args.preroll_s,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "FlashAttention fwd+bwd, %s,H=%d,N=%d,D=%d %s %s (BASELINE configs[%s]%s)%s" % (
                           ("B=%d per GPU" % B) if scaling == "weak" else ("global B=%d sharded by batch" % B), H, S, D, mask,
                           args.dtype, {"3": "2", "4": "3", "5": "4"}[args.config] if not overridden else "-, overridden",
                           "; fwd_* fields = configs[1]" if args.config == "3" and not overridden else "",
                           "" if args.layout == "bhsd" else "; inputs are [B,S,H,D] views read in place"),
                       "global_batch": GB, "seq_len": S, "parallelism": "batch-sharded x%d, no collective" % world},
            "fwd_bwd_tflops": round(tf_step, 2), "fwd_tflops": round(tf_fwd, 2), "fwd_ms": round(ms_fwd, 4),
            "pct_mfma_peak_fwd_bwd": round(100 * tf_step / world / PEAK_TFLOPS, 2),
            "pct_mfma_peak_fwd": round(100 * tf_fwd / world / PEAK_TFLOPS, 2),
            "checksum": m["checksum"],
            "kernels": kernels, "roofline": roofline, "cpu_baseline": cpu, "config5": c5,
        }
        if args.rehearse_one_gpu:
            line["rehearsal"] = True   # every rank shared GPU 0: plumbing check, not a measurement
        if "ms_fwd_copy" in m:
            line["fwd_ms_if_views_were_copied_first"] = round(m["ms_fwd_copy"], 4)
        print(json.dumps(line), flush=True)
    sc.finalize()


def harness_selftest(args):
    """The multi-rank plumbing of this script with no GPU: gloo ranks, the config's sharding, the barrier-bracketed
    timed loop, MAX / SUM reduces and the single rank-0 JSON line.  The step is a trivial torch CPU op on the shard --
    the product kernels have no CPU path and are not involved."""
    rank, local_rank, world = sc.init(backend="gloo")
    dev = torch.device("cpu")
    cb, ch, cs, cd, scaling = CONFIGS[args.config]
    lo, hi = (rank * cb, (rank + 1) * cb) if scaling == "weak" else sc.shard_range(cb, rank, world)
    x = torch.cat([sc.make_batch(i, 1, 4, 4, 8, torch.float32, dev, with_dout=False)[0] for i in range(lo, hi)])
    calls = []

    def step():
        calls.append(1)
        return (x * x).sum()

    ms = sc.timed_steps(step, args.steps, args.warmup, dev)
    assert len(calls) == args.steps + args.warmup
    total = sc.sum_over_ranks(float(hi - lo), dev)
    cs_ = sc.sum_over_ranks(sc.checksum([x]), dev)
    if rank == 0:
        print(json.dumps({"metric": "harness-selftest", "value": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(ms / args.steps, 6), "scaling": scaling,
                          "data": "harness-selftest", "global_batch": int(total), "checksum": cs_}), flush=True)
    sc.finalize()


def main():
    args = parse_args()
    if args.gpus < 1:
        fail("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no device query here: the parent must not open the GPU before it starts its children (torch.cuda.device_count()
        # falls back to hipGetDeviceCount when amdsmi discovery fails); every rank checks the device count itself and
        # exits 2 with "only N GPU(s) visible", which spawn_ranks returns
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
