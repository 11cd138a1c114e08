"""Parity metrics: counterpart of the reference's code/_verify_func.py:3-40.

Same five metrics, same pass criterion (allclose(rtol, atol) on the fp32 upcasts and
cosine > 0.999), same defaults and printout; additionally RETURNS the numbers so tests can
assert on them (the reference prints and returns None).
"""
import torch


def verify_results(bench, triton_output, name="Attention", rtol=1e-2, atol=1e-3, verbose=True):
    # metrics in fp32 so that computing them adds no second rounding (V:4-6)
    b = bench.to(torch.float32)
    t = triton_output.to(torch.float32)
    diff_abs = torch.abs(b - t)

    max_abs_err = torch.max(diff_abs).item()
    mean_abs_err = torch.mean(diff_abs).item()
    # relative error against the BENCH magnitude, eps avoids 0/0 (V:13-15)
    max_rel_err = torch.max(diff_abs / (torch.abs(b) + 1e-5)).item()
    # allclose-style normalised error against the OUTPUT magnitude; < 1 passes (V:17-20)
    max_norm = (diff_abs / (atol + rtol * t.abs())).max().item()
    cosine_sim = torch.nn.functional.cosine_similarity(b.flatten(), t.flatten(), dim=0).item()

    is_allclose = torch.allclose(b, t, rtol=rtol, atol=atol)
    passed = bool(is_allclose and cosine_sim > 0.999)
    if verbose:
        print(f"[{name} Verification]")
        print(f"Max Abs Error: {max_abs_err:.2e}")
        print(f"Mean Abs Error: {mean_abs_err:.2e}")
        print(f"Max Rel Error: {max_rel_err:.2e}")
        print(f"Max Normalized Error (allclose-style): {max_norm:.2e}")
        print(f"Cosine Similarity: {cosine_sim:.6f}")
        print("✅ Test Passed!" if passed else "❌ Test Failed!")
    return {
        "max_abs": max_abs_err,
        "mean_abs": mean_abs_err,
        "max_rel": max_rel_err,
        "max_norm": max_norm,
        "cos": cosine_sim,
        "passed": passed,
    }
