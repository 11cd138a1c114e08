"""One-process-per-GPU scaling harness for the attention hot path.

The path is embarrassingly parallel over (batch, head): every kernel program touches only
its own (b, h) slice (reference K:60-63, K:188-191, K:315-318), so it shards over the GPUs
of a node BY BATCH with no collective in the data path.  torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm, "gloo" on CPU for the tests) is used only OUTSIDE the timed region:
barrier, MAX-reduce of the elapsed time, SUM-reduce of shard checksums.

Inputs are generated per GLOBAL batch index (seed = base_seed + index), so the union of
all shards is the same tensor whatever the world size.
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Initialise the process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def finalize():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_range(global_batch, rank, world):
    """Contiguous batch slice [lo, hi) of this rank; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_batch(index, H, S_q, S_k, D, dtype, device, base_seed=1000, with_dout=True):
    """Q, K, V (and dO) of ONE global batch index: [1, H, S, D] each, N(0,1), seeded by the index."""
    g = torch.Generator(device=device)
    g.manual_seed(base_seed + index)
    shapes = [(1, H, S_q, D), (1, H, S_k, D), (1, H, S_k, D)] + ([(1, H, S_q, D)] if with_dout else [])
    return [torch.randn(s, generator=g, device=device, dtype=torch.float32).to(dtype) for s in shapes]


def make_shard(lo, hi, H, S_q, S_k, D, dtype, device, base_seed=1000, with_dout=True):
    parts = [make_batch(i, H, S_q, S_k, D, dtype, device, base_seed, with_dout) for i in range(lo, hi)]
    return [torch.cat([p[j] for p in parts], dim=0).contiguous() for j in range(len(parts[0]))]


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def barrier(device):
    _sync(device)
    if dist.is_initialized():
        dist.barrier()
    _sync(device)


def timed_steps(step, steps, warmup, device, preroll_s=0.0):
    """`preroll_s` seconds of untimed steps (clock settle), then `warmup` untimed steps, then EXACTLY `steps` steps between
    barrier+synchronize pairs.  Returns the elapsed milliseconds, MAX over ranks."""
    if preroll_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < preroll_s:
            for _ in range(8):
                step()
            _sync(device)
    for _ in range(warmup):
        step()
    barrier(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _sync(device)
    ms = (time.perf_counter() - t0) * 1e3
    barrier(device)
    return max_over_ranks(ms, device)


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def checksum(tensors):
    """Order-independent fp64 checksum of a list of tensors (sum of sums and of squares)."""
    s = 0.0
    for t in tensors:
        d = t.detach().to(torch.float64)
        s += float(d.sum().item()) + float((d * d).sum().item())
    return s
