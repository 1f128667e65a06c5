import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
b = pkg.scenes.golden_buffers("c2"); view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["cornell"])
ctx = pkg.Context(0); ctx.upload_scene(b); ctx.set_params(max_bounces=8); ctx.resize(1920, 1080)
for world in (1, 2, 4, 8):
    ctx.set_shard(0, world, 4096); spp = 64 * world
    ctx.clear(); ctx.render(view, 1, spp); ctx.synchronize(); ctx.reset_stats()
    t = time.perf_counter()
    for _ in range(3):
        ctx.clear(); ctx.render(view, 1, spp); ctx.synchronize()
    dt = time.perf_counter() - t; st = ctx.stats()
    print("rank 0 of %d, %d spp: %.0f Mrays/s per GPU, %.1f ms/step" % (world, spp, st["rays"] / dt / 1e6, dt / 3 * 1e3))
