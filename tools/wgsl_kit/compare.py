#!/usr/bin/env python3
"""compare.py ours.f32 reference.f32 --width W --height H --frames N [--first ours_first.f32 --second ours_second.f32]

Statistics that are meaningful between two Monte-Carlo renders whose elementary functions differ in the last bits
(tools/wgsl_kit/README.md): per-pixel RMSE against the RMSE two INDEPENDENT renders would show, bias of the image mean in
units of its standard error, fraction of bit-identical pixels, and north_star's per-pixel L2 for the record."""
import argparse
import json
import os

import numpy as np


def load(path, W, H):
    a = np.fromfile(path, np.float32)
    assert a.size == W * H * 4, "%s: %d floats, expected %d" % (path, a.size, W * H * 4)
    return a.reshape(H, W, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ours")
    ap.add_argument("reference")
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--frames", type=int, required=True)
    ap.add_argument("--first", default=None, help="this build's frames 1..N/2 (default: ours_first.f32 next to `ours`)")
    ap.add_argument("--second", default=None)
    a = ap.parse_args()
    W, H, N = a.width, a.height, a.frames
    ours, ref = load(a.ours, W, H), load(a.reference, W, H)
    o, r = ours[..., :3].astype(np.float64) / N, ref[..., :3].astype(np.float64) / N
    ok = np.isfinite(o).all(-1) & np.isfinite(r).all(-1)
    d = (o - r)[ok]
    out = {
        "pixels": int(ok.size), "non_finite_pixels": int((~ok).sum()),
        "bit_identical_pixel_frac": float((ours.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()),
        "per_pixel_l2_mean": float(np.sqrt((d ** 2).sum(-1)).mean()), "per_pixel_l2_max": float(np.sqrt((d ** 2).sum(-1)).max()),
        "pixels_over_1e-4": int((np.sqrt((d ** 2).sum(-1)) > 1e-4).sum()),
        "rmse": float(np.sqrt((d ** 2).mean())),
    }
    base = os.path.dirname(os.path.abspath(a.ours))
    f1, f2 = a.first or os.path.join(base, "ours_first.f32"), a.second or os.path.join(base, "ours_second.f32")
    if os.path.exists(f1) and os.path.exists(f2):
        h1 = load(f1, W, H)[..., :3].astype(np.float64) / (N // 2)
        h2 = load(f2, W, H)[..., :3].astype(np.float64) / (N - N // 2)
        hd = (h1 - h2)[ok]
        # Var(h1 - h2) = 2 sigma^2 / (N/2) = 4 sigma^2 / N; two independent N-frame renders differ with variance 2 sigma^2 / N
        indep = np.sqrt((hd ** 2).mean() / 2.0)
        out["expected_rmse_of_two_independent_renders"] = float(indep)
        out["rmse_over_expected_independent"] = float(out["rmse"] / indep) if indep > 0 else None
        per_px_var = (hd ** 2) / 4.0 * N          # sigma^2 per pixel and channel (noisy, unbiased)
        stderr_mean = np.sqrt((2.0 * per_px_var / N).sum(0)) / ok.sum()
        out["mean_diff_rgb"] = [float(x) for x in d.mean(0)]
        out["mean_diff_over_stderr_rgb"] = [float(x) for x in d.mean(0) / np.where(stderr_mean > 0, stderr_mean, 1)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
