"""Does k_shade's time on configs[1] depend on where the path buffers happen to be allocated?  Several contexts alive in one process, timed round-robin:
a context that is consistently slower than its neighbours points at placement, one that is slow only sometimes at the clock."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g._load_pkg()
b = pkg.scenes.golden_buffers("c2")
view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS["cornell"])
n_ctx, rounds = (int(sys.argv[1]) if len(sys.argv) > 1 else 5), (int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctxs = []
for i in range(n_ctx):
    ctx = pkg.Context(0)
    ctx.upload_scene(b); ctx.set_params(max_bounces=8); ctx.resize(1920, 1080)
    ctx.render(view, 1, 64); ctx.synchronize()
    ctxs.append(ctx)
tab = [[0.0] * rounds for _ in ctxs]
for r in range(rounds):
    for i, ctx in enumerate(ctxs):
        ctx.reset_stats(); ctx.set_timing(1)
        for _ in range(2):
            ctx.clear(); ctx.render(view, 1, 64)
        st = ctx.stats(); ctx.set_timing(0)
        tab[i][r] = (st["shade_ms"] / 2, st["generate_ms"] / 2)
for i in range(n_ctx):
    print("context %d  shade: %s   generate: %s" % (i, " ".join("%5.2f" % t[0] for t in tab[i]), " ".join("%4.2f" % t[1] for t in tab[i])), flush=True)
