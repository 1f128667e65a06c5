"""Do two contexts (two HIP streams) rendering half the frames each finish sooner than one context rendering all of them?"""
import sys, time, threading
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
def mk():
    c = pkg.Context(0); c.upload_scene(b); c.set_params(max_bounces=8, stack_size=24 if wl != "c2" else 20); c.resize(1920, 1080); return c
A, B = mk(), mk()
def run(c, first, n):
    c.clear(); c.render(view, first, n); c.synchronize()
run(A, 1, spp); run(B, 1, spp // 2)
for rep in range(2):
    t = time.perf_counter(); run(A, 1, spp); one = time.perf_counter() - t
    t = time.perf_counter()
    ta = threading.Thread(target=run, args=(A, 1, spp // 2)); tb = threading.Thread(target=run, args=(B, 1 + spp // 2, spp // 2))
    ta.start(); tb.start(); ta.join(); tb.join()
    two = time.perf_counter() - t
    print("%s %d spp: one context %.1f ms, two contexts with half each %.1f ms" % (wl, spp, one * 1e3, two * 1e3), flush=True)
