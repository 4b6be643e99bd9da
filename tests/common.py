"""Shared helpers for the parity tests: seeded models / inputs and an error reporter."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from effi_mvs_plus_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def model_args(ndepths="48,8,8", gru="3,3,3", cost_num=3):
    return argparse.Namespace(ndepths=ndepths, GRUiters=gru, CostNum=cost_num)


def build_model(ndepths="48,8,8", seed=1, device="cpu"):
    """Our model with seeded-random weights (BN statistics randomised) + the same state dict on CPU."""
    from effi_mvs_plus_amd.models import Effi_MVS_plus
    with contextlib.redirect_stdout(io.StringIO()):
        net = Effi_MVS_plus(model_args(ndepths))
    sd = synth.randomize_state_dict(net.state_dict(), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.eval()
    if device != "cpu":
        net = net.to(device)
    return net, sd


def stats(got, want):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    diff = (got - want).abs()
    scale = want.abs().max().item() + 1e-30
    return {"max_abs": diff.max().item(), "mean_abs": diff.mean().item(), "ref_max": scale,
            "max_rel_to_peak": diff.max().item() / scale,
            "p99": torch.quantile(diff.flatten()[:: max(1, diff.numel() // 1_000_000)], 0.99).item()}


def check_close(name, got, want, rtol=1e-4, atol=1e-5, frac_ok=1.0, outlier_atol=None):
    """allclose with a readable report; ``frac_ok`` < 1 tolerates a small fraction of outliers
    (discontinuities: floor / clamp / out-of-bounds flips).  The excluded set is not ignored: its size and its largest error
    are printed and the largest error is bounded by ``outlier_atol`` (default: the peak of ``want`` -- a flip can move a value
    across the data's own range, garbage cannot hide behind the fraction)."""
    assert tuple(got.shape) == tuple(want.shape), f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    s = stats(got, want)
    g, w = got.detach().double().cpu(), want.detach().double().cpu()
    err = (g - w).abs()
    ok = (err <= atol + rtol * w.abs())
    frac = ok.double().mean().item()
    n_out = int((~ok).sum())
    out_max = float(err[~ok].max()) if n_out else 0.0
    print(f"[parity] {name:42s} max_abs={s['max_abs']:.3e} mean_abs={s['mean_abs']:.3e} p99={s['p99']:.3e} "
          f"peak={s['ref_max']:.3e} within_tol={frac:.6f} excluded={n_out} excluded_max_abs={out_max:.3e}")
    assert torch.isfinite(g).all(), f"{name}: non-finite values"
    assert frac >= frac_ok, f"{name}: only {frac:.6f} of elements within rtol={rtol} atol={atol} (need {frac_ok})"
    bound = s["ref_max"] + atol if outlier_atol is None else outlier_atol
    assert out_max <= bound, f"{name}: an excluded element is off by {out_max:.3e} (bound {bound:.3e})"
    s["excluded"], s["excluded_max_abs"] = n_out, out_max
    return s


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def t(x, device):
    return x.to(device=device, dtype=torch.float32).contiguous()


def conv_tol(precision, want, rtol, atol, layers=1):
    """Per-operator tolerance of a convolution chain: the fp32 MFMA path keeps (rtol, atol); the split-bf16 path adds
    4e-5 of the output's peak per chained layer (each product carries ~2^-16 relative error; measured ~5e-6 of peak per layer)."""
    if precision == "split":
        peak = float(torch.as_tensor(want).abs().max())
        return dict(rtol=rtol, atol=atol + 4e-5 * layers * peak)
    return dict(rtol=rtol, atol=atol)
