"""CPU test: register spills in the gfx950 kernels of libmi355fa.so.

The code objects are read straight out of the shared library (tools/codeobj.py: ELF section .hip_fatbin -> clang
offload bundles -> AMDGPU metadata notes); no GPU and no HIP runtime are involved.  A kernel the schedule rule of
csrc/fa_kernels.h can select must not spill: scratch traffic showed up as 1.43x the algorithmic HBM bytes in round 1's
headline dQ kernel.  Variants that only fa_debug_force_impl() can reach are listed explicitly.
"""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj  # noqa: E402

def _table():
    """csrc/fa_table.h parsed: family[kernel][D128][bf16][causal][bh][s]."""
    import re
    src = open(os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd", "csrc", "fa_table.h")).read()
    body = re.sub(r"//[^\n]*", "", src[src.index("kFamily["):])
    body = body[body.index("=") + 1:]
    nums = [int(x) for x in re.findall(r"\b\d+\b", body)]
    assert len(nums) == 3 * 2 * 2 * 2 * 4 * 8, len(nums)
    it = iter(nums)
    return [[[[[[next(it) for _ in range(8)] for _ in range(4)] for _ in range(2)] for _ in range(2)] for _ in range(2)]
            for _ in range(3)]


def _forced_only():
    """Kernel variants the generated table never selects (only fa_debug_force_impl() reaches them): derived from
    fa_table.h, so that a re-tuned table that starts selecting one of them is held to the no-spill rule at once."""
    t = _table()
    stems = {(0, 2): "fa_fwd2_kernel", (0, 3): "fa_fwd3_kernel", (1, 2): "fa_bwd_dq2_kernel", (1, 3): "fa_bwd_dq3_kernel",
             (1, 4): "fa_bwd_dq4_kernel", (2, 3): "fa_bwd_dkv3_kernel", (2, 4): "fa_bwd_dkv4_kernel"}      # D = 64 only families, templated <T, CAUSAL>
    out = []
    for (kern, fam), stem in stems.items():
        for bf16, tname in ((0, "FP16"), (1, "BF16")):
            for causal in (0, 1):
                used = any(fam in row for row in t[kern][0][bf16][causal])
                if not used:
                    out.append("%sINS_4%sELb%dE" % (stem, tname, causal))
    return tuple(out)


FORCED_ONLY = _forced_only()


def _is_dropout_variant(name):
    # fa_fwd_kernel<D, T, CAUSAL, DROP>, fa_bwd_dq_kernel<D, T, CAUSAL, OCC, DROP>, fa_bwd_dkv_kernel<D, T, CAUSAL, DROP>
    short = codeobj.demangle_short(name)
    if "<" not in short:   # not a template: the debug poison kernel (fa_api.hip)
        return False
    stem, args = short.split("<")[0], short.rstrip(">").split("<")[1].split(",")
    return (stem in ("fa_fwd_kernel", "fa_bwd_dkv_kernel") and len(args) == 4 and args[3] == "true") or \
           (stem == "fa_bwd_dq_kernel" and len(args) == 5 and args[4] == "true")


def test_library_contains_the_expected_kernels():
    ks = codeobj.kernels()
    names = " ".join(k["name"] for k in ks)
    for stem in ("fa_fwd_kernel", "fa_fwd2_kernel", "fa_fwd3_kernel", "fa_bwd_dq_kernel", "fa_bwd_dq2_kernel", "fa_bwd_dq3_kernel",
                 "fa_bwd_dkv_kernel", "fa_bwd_dkv2_kernel", "fa_bwd_dkv3_kernel", "fa_fwd4_kernel", "fa_bwd_dq4_kernel", "fa_bwd_dkv4_kernel"):
        assert stem in names, stem
    assert len(ks) >= 40
    assert all(k["wg"] == 256 for k in ks)


def test_rule_selectable_kernels_do_not_spill():
    # (the dropout variants included: spill-free since round 3)
    bad = [(codeobj.demangle_short(k["name"]), k["spill"], k["scratch"]) for k in codeobj.kernels()
           if (k["spill"] or k["scratch"]) and not any(f in k["name"] for f in FORCED_ONLY)]
    # (SGPR spills go to VGPR lanes with v_writelane, not to memory: they cost no scratch traffic and are not counted)
    assert not bad, "kernels with register spills / scratch: %s" % bad


def test_register_budgets_match_the_intended_occupancy():
    """VGPR + AGPR per lane against the workgroups-per-CU each kernel is designed for (512 registers per SIMD lane,
    allocation granule 8): 3 workgroups per CU need <= 168, 2 need <= 256, 1 needs <= 512."""
    for k in codeobj.kernels():
        n = k["name"]
        total = k["vgpr"]      # .vgpr_count already includes the AGPRs on gfx950
        if _is_dropout_variant(n):
            assert total <= (512 if "dkv_kernelILi128E" in n else 256), (n, total)
        elif "fa_fwd_kernelILi64E" in n or "fa_bwd_dq_kernelILi64ENS_4BF16ELb1ELi3E" in n or "fa_bwd_dq_kernelILi64ENS_4BF16ELb0ELi3E" in n:
            assert total <= 168, (n, total)
        elif "dkv_kernelILi128E" in n or "dkv2_kernelILi128E" in n or "fa_bwd_dkv3_kernel" in n or "fa_fwd4_kernel" in n or "fa_bwd_dq4_kernel" in n or "fa_bwd_dkv4_kernel" in n or "fa_poison_kernel" in n:   # one workgroup per CU
            assert total <= 512, (n, total)
        else:
            assert total <= 256, (n, total)


# Findings of tools/mfma_lint.py that predate the lint (round 3 kernels, all GPU tests green with them); each is a kernel
# name -> number of findings that may not grow.  Round 4 keeps the list to shrink it, never to add to it.
KNOWN_LINT = {}


def test_inline_asm_mfmas_respect_the_wait_states_hipcc_does_not_pad():
    """the VGPR-form MFMAs are inline asm, invisible to hipcc's hazard recognizer, and both hazards they open have
    been met on the GPU as silent wrong results (a chain-start operand's registers reused inside the MFMA's read window; an
    operand copied into place right in front of the MFMA).  tools/mfma_lint.py checks every such MFMA in the built code
    objects on a wait-state model, plus that nothing but pin_write / MFMAs touches the pinned accumulator registers."""
    import mfma_lint
    res = mfma_lint.lint()
    assert any("fa_bwd_dq4_kernel" in k for k in res) and any("fa_bwd_dkv4_kernel" in k for k in res) and any("fa_bwd_dkv3_kernel" in k for k in res) and any("fa_fwd4_kernel" in k for k in res)
    bad = {k: v[1] for k, v in res.items() if len(v[1]) > KNOWN_LINT.get(k, 0)}
    assert not bad, "\n".join("%s: %s" % (k, "; ".join(v[:4])) for k, v in bad.items())
