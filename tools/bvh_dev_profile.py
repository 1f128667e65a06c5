"""Kernel-level profile of ptmi_build_bvh_device on the configs[2] mesh: run under `rocprofv3 --kernel-trace --stats`."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g._load_pkg()
sc = pkg.scenes.c3_scene(); sc.init_mesh_data(); sc.create_meshes()
bmin = np.concatenate([m.bmin for m in sc.meshes]); bmax = np.concatenate([m.bmax for m in sc.meshes])
ctx = pkg.Context(0)
ctx.build_bvh(bmin[:1000], bmax[:1000])
for _ in range(2):
    t = time.perf_counter(); ctx.build_bvh(bmin, bmax); print("device build %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
