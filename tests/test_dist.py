"""CPU tests (gloo, world_size 2) of the batch-sharding harness used by bench.py for N > 1, and one GPU rehearsal of that path."""
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


# ---- bench.py's own rank management: harness only, gloo, no GPU, no kernels ----
def _bench(args, env=None, timeout=150):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr


@pytest.mark.timeout(300)
def test_bench_spawns_its_own_ranks_and_shards_config5():
    rc1, one, err1 = _bench(["--gpus", "1", "--harness-selftest", "--config", "5", "--steps", "3", "--warmup", "1"])
    rc2, two, err2 = _bench(["--gpus", "2", "--harness-selftest", "--config", "5", "--steps", "3", "--warmup", "1"])
    assert rc1 == 0 and rc2 == 0, (err1, err2)
    assert len(one) == 1 and len(two) == 1            # exactly ONE JSON line, printed by rank 0
    a, b = one[0], two[0]
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and a["scaling"] == b["scaling"] == "strong"
    assert a["global_batch"] == b["global_batch"] == 64      # the shards cover the global batch of BASELINE configs[4]
    assert abs(a["checksum"] - b["checksum"]) <= 1e-9 * abs(a["checksum"])   # same whole-job data at N = 1 and N = 2
    assert b["steps"] == 3 and b["warmup"] == 1 and b["ms_per_step"] > 0


@pytest.mark.timeout(120)
def test_bench_rank_count_mismatch_is_an_error_not_a_single_process_run():
    rc, lines, err = _bench(["--gpus", "2", "--harness-selftest"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "WORLD_SIZE=1" in err
    # without a GPU the real benchmark refuses to start ranks at all (and never falls back to anything)
    if not torch.cuda.is_available():
        rc, lines, err = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert rc != 0 and not lines and "GPU" in err


def test_bench_parent_never_queries_the_gpu_before_it_spawns(monkeypatch):
    """the spawning parent may not touch anything that can open /dev/kfd (torch.cuda.device_count()
    falls back to hipGetDeviceCount when amdsmi discovery fails) -- the ranks check the device count themselves."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_parent", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def boom(*a, **k):
        raise AssertionError("the bench.py parent queried the GPU before spawning its ranks")

    for name in ("device_count", "is_available", "init", "_lazy_init", "set_device", "current_device"):
        monkeypatch.setattr(torch.cuda, name, boom)
    if hasattr(torch._C, "_cuda_getDeviceCount"):
        monkeypatch.setattr(torch._C, "_cuda_getDeviceCount", boom)
    seen = {}
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv: seen.update(n=n, argv=argv) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["n"] == 2 and not torch.cuda.is_initialized()


def test_bench_metric_string_follows_the_arguments():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.CONFIGS["5"] == (64, 32, 8192, 64, "strong") and bench.CONFIGS["3"][:4] == (4, 32, 4096, 64)
    a = bench.parse_args([])
    assert (a.gpus, a.config) == (1, "3") and a.steps * 1.3 < 60          # the default finishes within minutes
    assert bench.flops_fwd(64, 32, 8192, 8192, 64, True) * 3.5 == 61572651155456     # tests/golden/kat.json
    assert len(bench.kernel_source_hash()) == 16


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_code_path_on_one_gpu_matches_the_single_rank_union():
    """The N > 1 path of bench.py with the real kernels (no 2-GPU box on this pool): two ranks share GPU 0 over gloo
    (--rehearse-one-gpu).  The line must describe a 2-rank weak-scaling job, and the checksum of the ranks' outputs must
    equal that of ONE rank computing the union of their shards -- the per-batch seeding makes it independent of the
    world size."""
    import json
    import subprocess
    common = ["--heads", "2", "--seq", "512", "--steps", "2", "--warmup", "1", "--preroll-s", "0", "--no-cpu-baseline"]

    def run(extra):
        e = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            e.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=e, capture_output=True,
                           text=True, timeout=500)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    two = run(["--gpus", "2", "--batch", "2", "--rehearse-one-gpu"])
    one = run(["--gpus", "1", "--batch", "4"])
    assert two["rehearsal"] is True and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["global_batch"] == 4 == one["config"]["global_batch"]
    assert two["config"]["parallelism"] == "batch-sharded x2, no collective"
    assert abs(two["checksum"] - one["checksum"]) <= 1e-6 * abs(one["checksum"])
    assert two["value"] > 0 and two["ms_per_step"] > 0


def test_nccl_branch_builds_the_right_arguments_with_a_mocked_process_group(monkeypatch):
    """the nccl (= RCCL) branch of the harness has never run -- the pool gives one-GPU boxes and the
    driver's 8-GPU node is the first to execute it.  With torch.distributed and torch.cuda mocked, check what it WOULD pass:
    the device is selected from LOCAL_RANK before init_process_group, device_id names that device, rendezvous defaults to
    127.0.0.1, and the scalars of the max / sum reduces live on that device for nccl (on the CPU for gloo)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, PKG)
    import _scaling as S
    calls = {}
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.setdefault("set_device", d))
    state = {"init": False}
    monkeypatch.setattr(dist, "is_initialized", lambda: state["init"])

    def fake_init(**kw):
        assert "set_device" in calls, "the device must be selected before the process group is created"
        calls["init"] = kw
        state["init"] = True
    monkeypatch.setattr(dist, "init_process_group", fake_init)
    monkeypatch.setattr(dist, "get_backend", lambda: "nccl")
    seen = []
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None: seen.append((t.device, t.dtype, op)))
    rank, local_rank, world = S.init()
    assert (rank, local_rank, world) == (3, 3, 8) and calls["set_device"] == 3
    kw = calls["init"]
    assert kw["backend"] == "nccl" and kw["rank"] == 3 and kw["world_size"] == 8
    assert kw["device_id"] == torch.device("cuda", 3)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["MASTER_PORT"]
    # the reduces: a float64 scalar on the rank's device (tensor construction itself is mocked away: no GPU here)
    made = []
    real_tensor = torch.tensor
    monkeypatch.setattr(torch, "tensor", lambda data, dtype=None, device=None: (made.append(str(device)), real_tensor(data, dtype=dtype))[1])
    dev = torch.device("cuda", 3)
    assert S.max_over_ranks(1.5, dev) == 1.5 and S.sum_over_ranks(2.5, dev) == 2.5
    assert made == ["cuda:3", "cuda:3"] and [op for _, _, op in seen] == [dist.ReduceOp.MAX, dist.ReduceOp.SUM]
    monkeypatch.setattr(dist, "get_backend", lambda: "gloo")
    made.clear()
    S.max_over_ranks(1.0, dev)
    assert made == ["cpu"]
