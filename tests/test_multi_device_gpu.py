"""ptmi_create_multi: one context over several GPUs — tiles of the caller's shard dealt round-robin to the local devices, ONE
collective inside ptmi_read_framebuffer: by default the tile gather (every device's own tiles read into place on the first device: 1/N of
the bytes, no arithmetic), PTMI_MULTI_REDUCE=rccl the ncclReduce of the full buffers over xGMI, =copy peer copies + an add kernel.
A one-GPU box can list device 0 several times: the shards then share the GPU; the RCCL library itself is exercised with a one-rank
communicator (PTMI_MULTI_REDUCE=rccl)."""
import numpy as np
import pytest

from conftest import assert_same_bits, cornell_view

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wavefront", "mixed"])
def pipeline(request, monkeypatch):
    """These images are small enough for k_tail to take every shard's queue whole at step 0 (the library's default): run every case through the
    per-bounce kernels as well (PTMI_TAIL_LIMIT=0), so that the tile arithmetic of k_generate / k_bvh / k_shade stays under test."""
    if request.param == "wavefront":
        monkeypatch.setenv("PTMI_TAIL_LIMIT", "0")
    else:
        monkeypatch.delenv("PTMI_TAIL_LIMIT", raising=False)
    return request.param


def _render(pkg, devices, b, view, w, h, frames, **params):
    with pkg.Context(devices) as ctx:
        ctx.upload_scene(b)
        ctx.set_params(**params)
        ctx.resize(w, h)
        ctx.set_counters(True)
        ctx.render(view, 1, frames)
        fb = ctx.read_framebuffer()
        st = ctx.stats()
        st["reduce_info"] = ctx.reduce_info()
        px = ctx.resolve_rgba8(frames)
    return fb, st, px


@pytest.mark.parametrize("n,mode", [(2, None), (3, None), (8, None), (3, "copy"), (8, "copy")])
def test_shards_in_one_context_equal_single_device(pkg, oracle, monkeypatch, n, mode):
    """... through the tile gather (the default: reduce_mode 4, the devices' own tiles put side by side) and through the peer-copy + add reduce."""
    b = pkg.scenes.golden_buffers("default")  # spheres, fog volumes, quads, a cube mesh
    view = cornell_view(pkg, "default") if "default" in pkg.scenes.CAMERAS else cornell_view(pkg)
    one, st1, px1 = _render(pkg, 0, b, view, 200, 120, 6, max_bounces=6)
    if mode:
        monkeypatch.setenv("PTMI_MULTI_REDUCE", mode)
    else:
        monkeypatch.delenv("PTMI_MULTI_REDUCE", raising=False)
    many, stn, pxn = _render(pkg, [0] * n, b, view, 200, 120, 6, max_bounces=6)
    assert stn["reduce_mode"] == (2 if mode else 4) and (("tile gather" in stn["reduce_info"]) == (mode is None)), stn["reduce_info"]
    if not mode:
        assert "0.0 MB between GPUs" in stn["reduce_info"]  # (the shards share the GPU: nothing crosses a link)
    assert_same_bits(many, one, "%d shards in one context" % n)
    assert np.array_equal(px1, pxn)
    for k in ("rays", "paths", "node_visits", "tri_tests", "sphere_tests", "quad_tests", "mat_fetches"):
        assert stn[k] == st1[k], k
    assert stn["devices"] == n and st1["devices"] == 1
    want, ost = oracle.render(b, 200, 120, view, 1, 6, max_bounces=6)
    assert_same_bits(many, want, "vs oracle")
    assert stn["rays"] == ost["rays"]


def test_multi_context_progressive_frames_checkpoint_and_nested_shard(pkg, oracle):
    """render_frame (incl. resetBuffer), reading back in the middle (the per-device buffers must stay partial sums),
    write_framebuffer (scattered to the owners) and a caller-side shard on top of the local one."""
    b = pkg.scenes.golden_buffers("c2m")
    view, v2 = cornell_view(pkg), cornell_view(pkg, "oblique")
    W, H = 96, 64

    def u(frame, reset, v):
        return np.concatenate([np.array([W, H, frame, reset], np.float32), np.asarray(v, np.float32).reshape(-1)])

    with pkg.Context([0, 0]) as ctx:
        ctx.upload_scene(b)
        ctx.set_params(max_bounces=5)
        ctx.resize(W, H)
        for f in range(1, 4):
            ctx.render_frame(u(f, 0, view))
        mid = ctx.read_framebuffer()
        want, _ = oracle.render(b, W, H, view, 1, 3, max_bounces=5)
        assert_same_bits(mid, want, "after 3 frames")
        for f in range(4, 6):
            ctx.render_frame(u(f, 0, view))
        want5, _ = oracle.render(b, W, H, view, 1, 5, max_bounces=5)
        assert_same_bits(ctx.read_framebuffer(), want5, "after 5 frames, read twice")
        ctx.render_frame(u(1, 1, v2))  # camera moved: resetBuffer = 1 replaces every pixel
        want_r, _ = oracle.render(b, W, H, v2, 1, 1, max_bounces=5)
        assert_same_bits(ctx.read_framebuffer(), want_r, "after reset")
        ctx.write_framebuffer(want5)  # resume from a checkpoint
        ctx.render(view, 6, 2)
        want7, _ = oracle.render(b, W, H, view, 1, 7, max_bounces=5)
        assert_same_bits(ctx.read_framebuffer(), want7, "checkpoint + 2 frames")
        # caller-side shard (2 processes) on top of the 2 local devices: this context owns ranks 2 and 3 of 4
        ctx.clear()
        ctx.set_shard(1, 2, 64)
        ctx.render(view, 1, 2)
        part = ctx.read_framebuffer().reshape(-1, 4)
        full, _ = oracle.render(b, W, H, view, 1, 2, max_bounces=5)
        tile = np.arange(W * H) // 64
        mine = (tile % 4) >= 2
        assert_same_bits(part[mine], full.reshape(-1, 4)[mine], "owned tiles")
        assert not part[~mine].any()
        with pytest.raises(pkg.PtmiError):
            ctx.framebuffer_device_ptr()


def test_rccl_library_one_rank_communicator(pkg, oracle, monkeypatch):
    """PTMI_MULTI_REDUCE=rccl on a single device: librccl is loaded, ncclCommInitAll builds a one-rank communicator and
    ptmi_read_framebuffer goes through ncclReduce (a copy for one rank).  What a one-GPU box can check of the RCCL path."""
    monkeypatch.setenv("PTMI_MULTI_REDUCE", "rccl")
    b = pkg.scenes.golden_buffers("c1")
    view = cornell_view(pkg)
    fb, st, _ = _render(pkg, [0], b, view, 128, 128, 3, max_bounces=4)
    want, ost = oracle.render(b, 128, 128, view, 1, 3, max_bounces=4)
    assert_same_bits(fb, want, "through ncclReduce")
    assert st["rays"] == ost["rays"]
    with pytest.raises(pkg.PtmiError):  # a communicator cannot hold one GPU twice
        pkg.Context([0, 0])


@pytest.mark.parametrize("where", ["init", "reduce", "mid"])
def test_rccl_failure_falls_back_to_the_peer_copy_reduce(pkg, hooks, oracle, monkeypatch, where):
    """The RCCL path must be able to fail without taking the render with it (ncclReduce with N > 1 has never run on hardware):
    when ncclCommInitAll fails the context is created all the same, when the reduce's group fails — at its start, or in the middle, after a
    ncclReduce has been enqueued: the communicators are then aborted before anybody waits for a stream — the read-back succeeds all the same, both
    through the peer-copy + add reduce, bit-identical, and the context says so (stats.reduce_mode 3, ptmi_reduce_info "FALLBACK: ...").
    PTMI_TEST_RCCL_FAIL (the tests' build of the library only) simulates the failures on a box where RCCL works."""
    monkeypatch.setenv("PTMI_MULTI_REDUCE", "rccl")
    monkeypatch.setenv("PTMI_TEST_RCCL_FAIL", where)
    b = pkg.scenes.golden_buffers("c1")
    view = cornell_view(pkg)
    with pkg.Context([0]) as ctx:  # the product build does not know the variable
        assert ctx.stats()["reduce_mode"] == 1 and "ncclReduce" in ctx.reduce_info()
    with pkg.Context([0], lib=hooks) as ctx:
        assert ctx.stats()["reduce_mode"] == (3 if where == "init" else 1)
        ctx.upload_scene(b)
        ctx.set_params(max_bounces=4)
        ctx.resize(128, 128)
        ctx.render(view, 1, 2)
        mid = ctx.read_framebuffer()
        assert ctx.stats()["reduce_mode"] == 3
        info = ctx.reduce_info()
        assert info.startswith("FALLBACK: hipMemcpyPeer + add") and "simulated failure" in info, info
        ctx.render(view, 3, 1)  # and the context goes on rendering and reading back
        fb = ctx.read_framebuffer()
    want2, _ = oracle.render(b, 128, 128, view, 1, 2, max_bounces=4)
    want3, _ = oracle.render(b, 128, 128, view, 1, 3, max_bounces=4)
    assert_same_bits(mid, want2, "first read-back, through the fall-back")
    assert_same_bits(fb, want3, "second read-back")
    monkeypatch.delenv("PTMI_TEST_RCCL_FAIL")
    with pkg.Context([0], lib=hooks) as ctx:  # (the simulated failure belongs to the context it was created with: RCCL itself is intact)
        assert ctx.stats()["reduce_mode"] == 1 and "ncclReduce" in ctx.reduce_info()


def test_two_real_devices_in_one_context(pkg, oracle, monkeypatch):
    """The in-library multi-GPU path on devices [0, 1]: the tile gather over peer access, then ncclCommInitAll over two GPUs and ncclReduce with two
    ranks, then the peer-copy reduce, against the oracle.  Needs two visible GPUs — the builder's box has one; the driver's multi-GPU node runs it."""
    if pkg.load_library().ptmi_device_count() < 2:
        pytest.skip("one GPU visible: the two-device communicator cannot be built here")
    b = pkg.scenes.golden_buffers("c2m")
    view = cornell_view(pkg)
    want, ost = oracle.render(b, 192, 108, view, 1, 4, max_bounces=6)
    for mode, expect in ((None, 4), ("rccl", 1), ("copy", 2)):
        if mode:
            monkeypatch.setenv("PTMI_MULTI_REDUCE", mode)
        else:
            monkeypatch.delenv("PTMI_MULTI_REDUCE", raising=False)
        with pkg.Context([0, 1]) as ctx:
            ctx.upload_scene(b)
            ctx.set_params(max_bounces=6)
            ctx.resize(192, 108)
            ctx.render(view, 1, 2)
            ctx.reduce_framebuffer()
            ctx.render(view, 3, 2)  # the per-device buffers stay partial sums across a reduce
            fb = ctx.read_framebuffer()
            st = ctx.stats()
            info = ctx.reduce_info()
        assert_same_bits(fb, want, "two GPUs, %s" % info)
        assert st["rays"] == ost["rays"] and st["devices"] == 2
        # an RCCL that fails on this node is reported, not fatal: mode 3 is a pass here too, and the line says why
        assert st["reduce_mode"] in (expect, 3), (st["reduce_mode"], info)
        assert st["peer_links"] in (0, 2)


@pytest.mark.parametrize("sah", [False, True], ids=["median", "sah"])
def test_scene_bvh_built_on_every_device_and_the_collective_on_its_own(pkg, oracle, sah):
    """ptmi_build_scene_bvh / ptmi_build_scene_bvh_sah on a multi-device context build the tree on every device (deterministic: the same bytes everywhere), and
    ptmi_reduce_framebuffer — the step's one collective as bench.py times it — leaves the image for ptmi_read_framebuffer: bit-identical to the
    host pipeline on one device."""
    sc = lambda: pkg.scenes.c3_scene(20011)
    host = sc().buffers(native=pkg.ptmi.NativeHost(), sah=sah)
    view = cornell_view(pkg)
    params = dict(max_bounces=5, stack_size=40 if sah else 24)
    one, st1, _ = _render(pkg, 0, host, view, 128, 72, 3, **params)
    with pkg.Context([0, 0, 0]) as ctx:
        ctx.upload_scene(sc().buffers_unbuilt())
        ctx.build_scene_bvh(sah=sah)
        ctx.set_params(**params)
        ctx.resize(128, 72)
        ctx.set_counters(True)
        ctx.render(view, 1, 3)
        ctx.reduce_framebuffer()
        got = ctx.read_framebuffer()
        st = ctx.stats()
        info = ctx.scene_bvh_info()
        assert info["nodes"] == host["bvh"].size // 12 and (sah or info["nodes"] == 2 * 20011 - 1)
        assert np.array_equal(ctx.read_scene_buffer("bvh", info["nodes"]).reshape(-1).view(np.uint32), np.asarray(host["bvh"], np.float32).view(np.uint32))
    assert_same_bits(got, one, "three shards, trees built on the device")
    for k in ("rays", "node_visits", "tri_tests", "quad_tests", "mat_fetches"):
        assert st[k] == st1[k], k
    want, ost = oracle.render(host, 128, 72, view, 1, 3, **params)
    assert_same_bits(got, want, "vs oracle")
