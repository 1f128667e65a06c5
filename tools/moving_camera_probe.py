"""Cost of one displayed frame while the camera MOVES (the reference's renderAnimation with resetBuffer = 1 every frame, renderer.js:173-188): no
render-ahead is possible, every frame is a lone wavefront pass.  python tools/moving_camera_probe.py c2|c3"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g._load_pkg()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
if wl == "c2":
    b = pkg.scenes.golden_buffers("c2"); cam = "cornell"
elif wl == "c3":
    b = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "cornell"
else:
    b = pkg.scenes.c4_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "interior"
eye, center = pkg.scenes.CAMERAS[cam][0], pkg.scenes.CAMERAS[cam][1]
ctx = pkg.Context(0); ctx.upload_scene(b); ctx.resize(1920, 1080)
for bounces in (8, 100):
    ctx.set_params(max_bounces=bounces, stack_size=24)
    views = [pkg.scenes.camera_view([eye[0] + 0.002 * k, eye[1], eye[2]], center) for k in range(72)]
    u = lambda f, v: np.concatenate([[1920, 1080, f, 1], v]).astype(np.float32)
    for f in range(8): ctx.render_frame(u(1, views[f]))
    ctx.synchronize()
    n = 64
    t = time.perf_counter()
    for f in range(n):
        ctx.render_frame(u(1, views[8 + f]))
    ctx.synchronize(); dt = time.perf_counter() - t
    # latency of ONE frame incl. the synchronisation a display needs
    t = time.perf_counter()
    for f in range(16):
        ctx.render_frame(u(1, views[f])); ctx.synchronize()
    dl = time.perf_counter() - t
    print("%s max_bounces %d, camera moving: %.3f ms/frame pipelined (%.0f fps), %.3f ms/frame with a sync after every frame" % (wl, bounces, dt / n * 1e3, n / dt, dl / 16 * 1e3), flush=True)
