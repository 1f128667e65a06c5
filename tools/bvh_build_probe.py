"""Build time of the builders on the configs[2] mesh (871,414 triangles) and on the 298 k-triangle count of the reference's
benchmarks.txt (its own JS: 4.5 s), incl. the single-threaded JavaScript restatement (js/bvh_time.mjs) under Node."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
sc = pkg.scenes.c3_scene(); sc.init_mesh_data(); sc.create_meshes()
bmin = np.concatenate([m.bmin for m in sc.meshes]); bmax = np.concatenate([m.bmax for m in sc.meshes])
nh = pkg.ptmi.NativeHost(); ctx = pkg.Context(0)
ctx.build_bvh(bmin[:1000], bmax[:1000])
for name, fn in (("host, all threads", lambda: nh.build_bvh(bmin, bmax)), ("device", lambda: ctx.build_bvh(bmin, bmax)), ("host SAH (1 thread)", lambda: nh.build_bvh_sah(bmin, bmax))):
    t = time.perf_counter(); fn(); print("%-22s %.3f s" % (name, time.perf_counter() - t), flush=True)
rng = np.random.default_rng(1)
c = rng.uniform(-1, 1, (871414, 3)); e = rng.uniform(0, 0.01, (871414, 3))
rmin, rmax = np.ascontiguousarray(c - e), np.ascontiguousarray(c + e)
for name, fn in (("host, random boxes", lambda: nh.build_bvh(rmin, rmax)), ("device, random boxes", lambda: ctx.build_bvh(rmin, rmax))):
    t = time.perf_counter(); fn(); print("%-22s %.3f s" % (name, time.perf_counter() - t), flush=True)

import json, os, shutil, subprocess, tempfile
node = shutil.which("node")
if node:
    for n in (298000, 871414):
        sc2 = pkg.scenes.c3_scene(n); sc2.init_mesh_data(); sc2.create_meshes()
        a = np.concatenate([m.bmin for m in sc2.meshes]); b = np.concatenate([m.bmax for m in sc2.meshes])
        with tempfile.NamedTemporaryFile(suffix=".f64", delete=False) as f:
            f.write(np.ascontiguousarray(a, np.float64).tobytes()); f.write(np.ascontiguousarray(b, np.float64).tobytes())
        try:
            out = subprocess.run([node, "--max-old-space-size=8192", os.path.join("webgpu-path-tracer_amd", "js", "bvh_time.mjs"), f.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            js = json.loads(out.stdout) if out.returncode == 0 else {"error": out.stderr[-300:]}
        finally:
            os.remove(f.name)
        t = time.perf_counter(); nh.build_bvh(a, b); th = time.perf_counter() - t
        t = time.perf_counter(); ctx.build_bvh(a, b); td = time.perf_counter() - t
        print("n=%d: JavaScript (1 thread, %s) %s ms | host native %.0f ms | device %.0f ms" % (n, js.get("node"), js.get("ms", js), th * 1e3, td * 1e3), flush=True)
