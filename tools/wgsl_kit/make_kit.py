#!/usr/bin/env python3
"""make_kit.py — this build's half of the off-box WGSL comparison (tools/wgsl_kit/README.md): renders the reference's default
scene at the reference's canvas size with the HIP path and writes, into --out:

  ours.f32         raw accumulation buffer after N frames (W*H*4 little-endian f32, running sum, [R,G,B,1])
  ours_first.f32   the same for frames 1..N/2, ours_second.f32 for frames N/2+1..N (the two halves give the noise estimate)
  ours_1.f32, ours_8.f32   after 1 and 8 frames (bit-identity decay, README item 3)
  uniforms.json    the 20 floats of every frame (renderer.js:70-77,265-278)
  constants.txt    the shaders/header.wgsl constants the browser side must use

Needs an MI355X (run through gpurun); the product path only, no oracle."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--bounces", type=int, default=100, help="MAX_BOUNCES (shaders/header.wgsl:10 ships 100)")
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--out", default="gpurun_out/wgsl_kit")
    a = ap.parse_args()
    pkg = entry._load_pkg()
    os.makedirs(a.out, exist_ok=True)
    b = pkg.scenes.golden_buffers("default")
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["default"])  # index.js:40
    N, W, H = a.frames, a.width, a.height
    with pkg.Context(0) as ctx:
        ctx.upload_scene(b)
        ctx.set_params(max_bounces=a.bounces)  # every other parameter = the shipped constants (ptmi_default_params)
        ctx.resize(W, H)

        def render(first, n, name):
            ctx.clear()
            ctx.render(view, first, n)
            fb = ctx.read_framebuffer()
            fb.tofile(os.path.join(a.out, name))
            return fb

        full = render(1, N, "ours.f32")
        h1 = render(1, N // 2, "ours_first.f32")
        render(N // 2 + 1, N - N // 2, "ours_second.f32")
        render(1, 1, "ours_1.f32")
        render(1, min(8, N), "ours_8.f32")
        p = ctx.get_params()
    uniforms = [[float(W), float(H), float(f), 0.0] + [float(v) for v in np.asarray(view, np.float32).reshape(-1)] for f in range(1, N + 1)]
    json.dump({"note": "uniforms[k] = the 20 f32 written for frame k+1: [W, H, frameNum, resetBuffer, viewMatrix col-major] (renderer.js:70-77)",
               "uniforms": uniforms}, open(os.path.join(a.out, "uniforms.json"), "w"))
    with open(os.path.join(a.out, "constants.txt"), "w") as f:
        f.write("shaders/header.wgsl:9-13 must read\n")
        f.write("const NUM_SAMPLES = %d;\nconst MAX_BOUNCES = %d;\nconst STRATIFY = %s;\nconst IMPORTANCE_SAMPLING = %s;\nconst STACK_SIZE = %d;\n" % (
            p.num_samples, p.max_bounces, "true" if p.stratify else "false", "true" if p.importance_sampling else "false", p.stack_size))
        f.write("background_color (shaders/traceRay.wgsl:8) = vec3f(%g, %g, %g); fov = %g degrees (shaders/main.wgsl:7)\n" % (p.background[0], p.background[1], p.background[2], p.fov_degrees))
        f.write("ray_tmin (shaders/header.wgsl:37) = %.9g; light / surface mixture (shaders/traceRay.wgsl:43,49) = %.9g / %.9g (only read when IMPORTANCE_SAMPLING)\n" % (
            p.tmin, p.light_mix, float(np.float32(1.0) - np.float32(p.light_mix))))
        f.write("canvas %dx%d, camera eye (0.5,0,2.5) center (0.5,0,0) up (0,1,0) (index.js:40), frames 1..%d, resetBuffer 0\n" % (W, H, N))
    m = full[..., :3] / N
    print(json.dumps({"out": a.out, "frames": N, "mean_rgb": [float(x) for x in m.reshape(-1, 3).mean(0)], "half_mean_rgb": [float(x) for x in (h1[..., :3] / (N // 2)).reshape(-1, 3).mean(0)]}))


if __name__ == "__main__":
    main()
