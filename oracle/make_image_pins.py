#!/usr/bin/env python3
"""Writes tests/golden/oracle_pins.json: SHA-256 of the oracle's framebuffers, work counters and first-bounce hit records on
the golden scenes.  These are REGRESSION pins of this build's own oracle — the reference has no image
or hit-record fixtures to pin against — so that an edit to oracle/ptm_oracle.cpp or include/ptmi_math.h cannot drift silently
(the GPU path is held bit-exact to the oracle, so it is pinned with it).  Canonical NaNs before hashing.
    python oracle/make_image_pins.py        (rewrites the file; review the diff)"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

CASES = {
    # name: (scene, camera, W, H, frames, params)
    "c1_256x256_4spp_4b": ("c1", "cornell", 256, 256, 4, dict(max_bounces=4)),  # BASELINE configs[0]
    "c1_256x256_4spp_4b_is": ("c1", "cornell", 256, 256, 4, dict(max_bounces=4, importance_sampling=1)),
    "c2_160x90_2spp_8b": ("c2", "cornell", 160, 90, 2, dict(max_bounces=8)),
    "c2m_128x96_2spp_6b_is": ("c2m", "oblique", 128, 96, 2, dict(max_bounces=6, importance_sampling=1)),
    "default_120x80_2spp_12b": ("default", "default", 120, 80, 2, dict(max_bounces=12)),
}


def canon(a):
    a = np.ascontiguousarray(a, np.float32).copy()
    a[np.isnan(a)] = np.float32(np.nan)
    return a.view(np.uint32)


def compute():
    pkg = entry._load_pkg()
    from oracle import ptm_oracle

    out = {}
    for name, (scene, cam, w, h, frames, params) in CASES.items():
        b = pkg.scenes.golden_buffers(scene)
        view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
        fb, st = ptm_oracle.render(b, w, h, view, 1, frames, **params)
        rec = {"framebuffer_sha256": hashlib.sha256(canon(fb).tobytes()).hexdigest(),
               "centre_pixel_bits": [int(x) for x in canon(fb[h // 2, w // 2])],
               "counters": {k: int(st[k]) for k in ("rays", "paths", "node_visits", "tri_tests", "sphere_tests", "quad_tests", "mat_fetches")}}
        out[name] = rec
    # G5: first-bounce hit records of a fixed ray fan (seeded) per scene
    rng = np.random.default_rng(2024)
    o = rng.uniform(-0.2, 0.2, (4096, 3)) + np.array([0, 0, 2.4])
    d = rng.uniform(-1, 1, (4096, 3)) * np.array([1.1, 1.0, 0.5]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    seeds = np.arange(4096, dtype=np.uint32) * np.uint32(2654435761)
    for scene in ("c1", "c2", "c2m", "default"):
        b = pkg.scenes.golden_buffers(scene)
        hit, g, st = ptm_oracle.hit_scene(b, rays, seeds, stack_size=20)
        m = hit["hit"] == 1
        blob = b"".join([hit["hit"].astype(np.int32).tobytes(), canon(hit["t"][m]).tobytes(), canon(hit["normal"][m]).tobytes(),
                         hit["front_face"][m].astype(np.int32).tobytes(), canon(hit["material"][m]).tobytes(), g.tobytes()])
        out["hits_" + scene] = {"sha256": hashlib.sha256(blob).hexdigest(), "n_hit": int(m.sum()), "node_visits": int(st["node_visits"])}
    return out


if __name__ == "__main__":
    pins = compute()
    path = os.path.join(ROOT, "tests", "golden", "oracle_pins.json")
    json.dump(pins, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
