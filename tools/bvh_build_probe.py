"""Build time of the three builders on the configs[2] mesh (871,414 triangles)."""
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
