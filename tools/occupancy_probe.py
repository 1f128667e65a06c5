"""k_bvh time vs waves per CU on a mid-size mesh whose stacks leave room in LDS (tuning aid; PTMI_WAVES_PER_CU is read by ptmi_reload_tuning)."""
import os, sys, time
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
cam = sys.argv[2] if len(sys.argv) > 2 else "cornell"
sc = pkg.scenes.c4_scene(n) if cam == "interior" else pkg.scenes.c3_scene(n)
b = sc.buffers(native=pkg.ptmi.NativeHost())
view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
ctx = pkg.Context(0); ctx.upload_scene(b)
stack = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ctx.set_params(max_bounces=8, stack_size=stack); ctx.resize(1920, 1080)
for w in (8, 12, 16, 20, 24):
    os.environ["PTMI_WAVES_PER_CU"] = str(w)
    ctx.reload_tuning()
    ctx.clear(); ctx.render(view, 1, 16); ctx.synchronize(); ctx.reset_stats(); ctx.set_timing(1)
    ctx.clear(); ctx.render(view, 1, 32); ctx.synchronize(); st = ctx.stats(); ctx.set_timing(0)
    print("tris %d waves/CU %d: bvh %.1f ms, shade %.1f ms, %.0f Mrays/s" % (n, w, st["bvh_ms"], st["shade_ms"], st["rays"] / (st["render_ms"] / 1e3) / 1e6), flush=True)
