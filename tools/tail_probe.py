"""What k_tail buys where the wavefront pipeline runs out of parallelism: lone frames (ptmi_render of 1 frame, wall clock per frame) and the
reference's default MAX_BOUNCES = 100, for several PTMI_TAIL_LIMIT values (0 = k_tail never launched).  python tools/tail_probe.py [limit ...]   (-1 = the library's own limits)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g._load_pkg()
out = {}
for wl, cam, bounces, W, H in (("c2", "cornell", 8, 1920, 1080), ("c2", "cornell", 100, 1920, 1080), ("default", "default", 100, 900, 600), ("c3", "cornell", 8, 1920, 1080), ("c3sah", "cornell", 8, 1920, 1080)):
    if wl == "c3":
        b = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost())
    elif wl == "c3sah":  # the same scene from the opt-in SAH tree, built on the GPU (ptmi_build_scene_bvh_sah)
        b = pkg.scenes.c3_scene().buffers_unbuilt()
    else:
        b = pkg.scenes.golden_buffers(wl)
    view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
    with pkg.Context(0) as ctx:
        ctx.upload_scene(b)
        if wl == "c3sah":
            ctx.build_scene_bvh(sah=True)
        ctx.set_params(max_bounces=bounces, stack_size=40 if wl == "c3sah" else 24); ctx.resize(W, H)
        for frames in (1, 8):
            ctx.clear(); ctx.render(view, 1, frames); ctx.synchronize()
            t = []
            for rep in range(5):
                ctx.clear(); ctx.synchronize()
                t0 = time.perf_counter(); ctx.render(view, 1 + rep, frames); ctx.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
            out["%%s_%%db_%%dx%%d_%%df_ms" %% (wl, bounces, W, H, frames)] = round(sorted(t)[2], 3)
print(json.dumps(out))
''' % ROOT
res = {}
for lim in sys.argv[1:] or ["0", "262144", "1048576", "4194304"]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PTMI_TAIL_LIMIT=lim), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode:
        print(r.stderr[-1500:]); sys.exit(1)
    res["limit_" + lim] = json.loads(r.stdout.strip().splitlines()[-1])
    print("limit", lim, res["limit_" + lim], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tail_probe.json"), "w"), indent=1)
