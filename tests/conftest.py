import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- GPU repeatability probe (round 4) -----------------------------------------------------------------------------------
# One box of the pool returned other bits from launch to launch for EVERY schedule family of EVERY kernel (bf16, first
# batch only, a few scattered elements; profiles/r04_race_stress.txt section 4) while four other boxes reproduced every
# result over thousands of launches.  Many GPU tests assert bit-equality (determinism, family-vs-family, strided-vs-dense):
# on such a card they fail for a reason that is not in this repository.  The probe runs once per session when GPU tests are
# selected: the round-1 forward (family 1: barrier-synchronised, vmcnt(0) waits only) 60 times on fixed bf16 inputs at the
# headline shape.  It never skips or hides anything -- it adds a line to the header and a note to every failing GPU test.
_PROBE = {"done": False, "bad": None}


def _probe_gpu():
    if _PROBE["done"]:
        return _PROBE["bad"]
    _PROBE["done"] = True
    try:
        import ctypes
        import torch
        import _mi355fa as fa
        lib = fa.lib
        lib.fa_debug_force_impl.argtypes = [ctypes.c_int] * 3
        lib.fa_debug_force_impl.restype = None
        B, H, S, D = 4, 32, 4096, 64
        g = torch.Generator(device="cuda").manual_seed(1234)
        Q, K, V = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
        O0, O1 = torch.empty_like(Q), torch.empty_like(Q)
        L0, L1 = (torch.empty(B, H, S, device="cuda") for _ in range(2))
        st = torch.cuda.current_stream().cuda_stream
        P = lambda t: t.data_ptr()
        lib.fa_debug_force_impl(1, 0, 0)
        assert lib.fa_fwd(P(Q), P(K), P(V), P(O0), P(L0), B, H, S, S, D, 1, 1, D ** -0.5, st) == 0
        changed = 0
        for _ in range(60):
            assert lib.fa_fwd(P(Q), P(K), P(V), P(O1), P(L1), B, H, S, S, D, 1, 1, D ** -0.5, st) == 0
            if not (torch.equal(O0.view(torch.int16), O1.view(torch.int16)) and torch.equal(L0, L1)):
                changed += 1
        lib.fa_debug_force_impl(0, 0, 0)
        _PROBE["bad"] = changed
    except Exception as e:   # the probe must never break a session
        _PROBE["bad"] = None
        _PROBE["error"] = repr(e)[:200]
    return _PROBE["bad"]


def pytest_report_header(config):
    if not _has_gpu() or "not gpu" in (config.getoption("-m") or ""):
        return None
    bad = _probe_gpu()
    if bad is None:
        return "GPU repeatability probe: not run (%s)" % _PROBE.get("error", "?")
    if bad:
        return ("GPU repeatability probe: THIS CARD DOES NOT REPRODUCE ITS OWN RESULTS -- the round-1 forward gave other bits in %d of 60 "
                "launches on fixed inputs; bit-equality assertions below may fail for that reason (tests/conftest.py)" % bad)
    return "GPU repeatability probe: 60 of 60 launches of the control kernel bit-identical"


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when == "call" and rep.failed and "gpu" in item.keywords and _PROBE.get("bad"):
        rep.sections.append(("GPU repeatability probe", "this card changed the control kernel's result in %d of 60 launches on fixed inputs "
                             "(see the session header): a bit-equality failure here is not evidence against the kernel under test" % _PROBE["bad"]))
