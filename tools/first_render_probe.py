import sys, time, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g._load_pkg()
b = pkg.scenes.golden_buffers("c2"); view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["cornell"])
for spp in (64, 256):
    with pkg.Context(0) as ctx:
        ctx.upload_scene(b); ctx.set_params(max_bounces=8); ctx.resize(1920, 1080)
        t0 = time.perf_counter(); ctx.render(view, 1, spp); ctx.synchronize(); t1 = time.perf_counter()
        ctx.clear(); ctx.render(view, 1, spp); ctx.synchronize(); t2 = time.perf_counter()
        print("tries", os.environ.get("PTMI_PLACEMENT_TRIES"), "spp", spp, "first render %.0f ms, second %.0f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
