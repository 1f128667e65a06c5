"""Per-frame cost of the reference's progressive loop (one ptmi_render_frame per frame, as renderer.js:173-188 drives it)
next to the batched ptmi_render."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
if wl == "c2":
    b = pkg.scenes.golden_buffers("c2"); cam = "cornell"
elif wl == "c3":
    b = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "cornell"
else:
    b = pkg.scenes.c4_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "interior"
view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
ctx = pkg.Context(0); ctx.upload_scene(b); ctx.resize(1920, 1080)
for bounces in (8, 100):
    ctx.set_params(max_bounces=bounces, stack_size=24)
    u = lambda f: np.concatenate([[1920, 1080, f, 0], view]).astype(np.float32)
    for f in range(1, 9): ctx.render_frame(u(f))
    ctx.synchronize(); ctx.clear(); ctx.synchronize()
    n = 64
    t = time.perf_counter()
    for f in range(1, n + 1): ctx.render_frame(u(f))
    ctx.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter(); ctx.clear(); ctx.render(view, 1, n); ctx.synchronize(); db = time.perf_counter() - t
    print(wl + " max_bounces %d: render_frame loop %.3f ms/frame (%.0f fps), batched %.3f ms/frame" % (bounces, dt / n * 1e3, n / dt, db / n * 1e3), flush=True)
