"""tools/wgsl_kit: the off-box comparison procedure against the reference's WGSL path (the only route to pinned device parity).
compare.py's statistics are checked on synthetic renders; make_kit.py runs on the GPU box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

KIT = os.path.join(ROOT, "tools", "wgsl_kit")


def _compare(*argv):
    r = subprocess.run([sys.executable, os.path.join(KIT, "compare.py")] + [str(a) for a in argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def test_compare_statistics_on_synthetic_renders(tmp_path):
    rng = np.random.default_rng(0)
    W, H, N, sig = 64, 48, 512, 0.5
    truth = rng.uniform(0, 2, (H, W, 3))

    def rend(n, bias=0.0):
        a = np.ones((H, W, 4), np.float32)
        a[..., :3] = ((truth + bias) * n + rng.normal(0, sig * np.sqrt(n), (H, W, 3))).astype(np.float32)
        return a

    h1, h2 = rend(N // 2), rend(N // 2)
    ours = h1 + h2
    ours[..., 3] = 1
    for name, a in (("ours.f32", ours), ("ours_first.f32", h1), ("ours_second.f32", h2), ("indep.f32", rend(N)), ("biased.f32", rend(N, 0.01))):
        a.tofile(tmp_path / name)
    o = _compare(tmp_path / "ours.f32", tmp_path / "indep.f32", "--width", W, "--height", H, "--frames", N)
    assert 0.9 < o["rmse_over_expected_independent"] < 1.1 and max(abs(x) for x in o["mean_diff_over_stderr_rgb"]) < 4
    assert o["pixels_over_1e-4"] > 0.9 * W * H  # two independent 512-spp renders never meet a 1e-4 per-pixel bar
    o = _compare(tmp_path / "ours.f32", tmp_path / "biased.f32", "--width", W, "--height", H, "--frames", N)
    assert max(abs(x) for x in o["mean_diff_over_stderr_rgb"]) > 10  # a 0.5 % bias is unmistakable in the image mean
    o = _compare(tmp_path / "ours.f32", tmp_path / "ours.f32", "--width", W, "--height", H, "--frames", N)
    assert o["bit_identical_pixel_frac"] == 1.0 and o["rmse"] == 0.0


@pytest.mark.gpu
def test_make_kit_writes_the_agreed_inputs(tmp_path, pkg, oracle):
    r = subprocess.run([sys.executable, os.path.join(KIT, "make_kit.py"), "--frames", "6", "--bounces", "5", "--width", "90", "--height", "60", "--out", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    u = json.load(open(tmp_path / "uniforms.json"))["uniforms"]
    assert len(u) == 6 and u[2][:4] == [90.0, 60.0, 3.0, 0.0] and len(u[0]) == 20
    assert "MAX_BOUNCES = 5" in open(tmp_path / "constants.txt").read()
    b = pkg.scenes.golden_buffers("default")
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["default"])
    want, _ = oracle.render(b, 90, 60, view, 1, 6, max_bounces=5)
    got = np.fromfile(tmp_path / "ours.f32", np.float32).reshape(60, 90, 4)
    assert np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(want).view(np.uint32))
    halves = np.fromfile(tmp_path / "ours_first.f32", np.float32) + np.fromfile(tmp_path / "ours_second.f32", np.float32)
    assert np.allclose(halves.reshape(60, 90, 4)[..., :3], got[..., :3], rtol=1e-5, atol=1e-5)
