"""Regression pins of the oracle (tests/golden/oracle_pins.json, written by oracle/make_image_pins.py): framebuffer hashes, work
counters and first-bounce hit records on the golden scenes.  The reference holds no image fixtures (parity unpinned by the
reference), so these pin THIS build's restatement against silent drift; the HIP path is checked against the
same hashes on the GPU."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pins_module():
    spec = importlib.util.spec_from_file_location("make_image_pins", os.path.join(ROOT, "oracle", "make_image_pins.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_pins.json")))


def test_oracle_reproduces_its_pins(oracle):
    got = _pins_module().compute()
    assert got.keys() == PINS.keys()
    for k in PINS:
        assert got[k] == PINS[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("name", [k for k in PINS if not k.startswith("hits_")])
def test_device_framebuffer_matches_pins(ctx, pkg, name):
    m = _pins_module()
    scene, cam, w, h, frames, params = m.CASES[name]
    ctx.upload_scene(pkg.scenes.golden_buffers(scene))
    ctx.set_params(**params)
    ctx.resize(w, h)
    ctx.reset_stats()
    ctx.set_counters(True)
    ctx.render(pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam]), 1, frames)
    fb = ctx.read_framebuffer()
    st = ctx.stats()
    ctx.set_counters(False)
    assert hashlib.sha256(m.canon(fb).tobytes()).hexdigest() == PINS[name]["framebuffer_sha256"]
    for k, v in PINS[name]["counters"].items():
        assert st[k] == v, k
