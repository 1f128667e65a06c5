"""One context vs one context with two (or three) shards on the same GPU (ptmi_create_multi with a repeated device id: two streams, tiles dealt
round-robin, renders enqueued from two host threads): do the shards fill each other's k_bvh tails?  python tools/twin_probe.py c3 256"""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if wl == "c2":
    b = pkg.scenes.golden_buffers("c2"); cam = "cornell"
elif wl == "c3":
    b = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "cornell"
else:
    b = pkg.scenes.c4_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "interior"
view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
for devs in (0, [0, 0], [0, 0, 0], 0, [0, 0]):
    c = pkg.Context(devs); c.upload_scene(b); c.set_params(max_bounces=8, stack_size=24 if wl != "c2" else 20); c.resize(1920, 1080)
    c.clear(); c.render(view, 1, spp); c.synchronize()
    ts = []
    for rep in range(3):
        t = time.perf_counter(); c.clear(); c.render(view, 1, spp); c.synchronize(); ts.append(time.perf_counter() - t)
    print("%s %d spp, context %s: %.1f ms" % (wl, spp, devs, min(ts) * 1e3), flush=True)
    c.close()
