"""CPU tests (gloo, world_size 2) of the batch-sharding harness used by bench.py for N > 1."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (PKG, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import _scaling as sc
    import fa_oracle as fo
    torch.set_num_threads(1)
    r, lr, w = sc.init(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    GB, H, S, D = 4, 2, 64, 64
    lo, hi = sc.shard_range(GB, rank, world)
    Q, K, V, dO = sc.make_shard(lo, hi, H, S, S, D, torch.float16, dev)
    calls = []

    def step():  # the CPU oracle stands in for the HIP kernels: the harness is what is under test
        calls.append(1)
        return fo.fwd_tiled(Q, K, V, True)

    ms = sc.timed_steps(step, steps=2, warmup=1, device=dev)
    assert len(calls) == 3 and ms > 0
    O, LSE = fo.fwd_tiled(Q, K, V, True)
    cs = sc.sum_over_ranks(sc.checksum([O, LSE]), dev)
    mx = sc.max_over_ranks(float(rank + 1), dev)
    ret[rank] = (lo, hi, cs, mx, ms)
    sc.finalize()


def test_shard_range_partitions():
    import _scaling as sc
    for gb in (1, 4, 7, 64):
        for w in (1, 2, 3, 8):
            spans = [sc.shard_range(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_inputs_are_independent_of_world_size():
    import _scaling as sc
    whole = sc.make_shard(0, 4, 2, 32, 48, 64, torch.float16, "cpu")
    parts = [sc.make_shard(lo, hi, 2, 32, 48, 64, torch.float16, "cpu") for lo, hi in ((0, 1), (1, 3), (3, 4))]
    for j in range(4):
        assert torch.equal(whole[j], torch.cat([p[j] for p in parts], dim=0))
    assert whole[0].shape == (4, 2, 32, 64) and whole[1].shape == (4, 2, 48, 64)


@pytest.mark.timeout(180)
def test_two_rank_gloo_harness_matches_single_process():
    import _scaling as sc
    import fa_oracle as fo
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    assert (ret[0][0], ret[0][1], ret[1][0], ret[1][1]) == (0, 2, 2, 4)
    # whole-job checksum equals the single-process one (no collective in the data path, shards independent)
    Q, K, V, dO = sc.make_shard(0, 4, 2, 64, 64, 64, torch.float16, torch.device("cpu"))
    O, LSE = fo.fwd_tiled(Q, K, V, True)
    ref = sc.checksum([O, LSE])
    assert abs(ret[0][2] - ref) < 1e-6 * abs(ref) and ret[0][2] == ret[1][2]
    assert ret[0][3] == ret[1][3] == 2.0          # MAX over ranks
    assert ret[0][4] == ret[1][4]                 # both ranks report the same (max) time
