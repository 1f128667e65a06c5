"""A few single-frame renders of a workload, for kernel traces: python tools/one_frame.py c3 [frames] [lone]
(default: ptmi_render_frame with a resting camera — renders ahead; "lone": ptmi_render of one frame at a time, what a moving camera pays)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g._load_pkg()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if wl == "c2":
    b = pkg.scenes.golden_buffers("c2"); cam = "cornell"
elif wl == "c3":
    b = pkg.scenes.c3_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "cornell"
else:
    b = pkg.scenes.c4_scene().buffers(native=pkg.ptmi.NativeHost()); cam = "interior"
view = pkg.scenes.camera_view(*pkg.scenes.CAMERAS[cam])
ctx = pkg.Context(0); ctx.upload_scene(b); ctx.set_params(max_bounces=8, stack_size=24); ctx.resize(1920, 1080)
lone = len(sys.argv) > 3 and sys.argv[3] == "lone"
for f in range(1, n + 1):
    if lone:
        ctx.render(view, f, 1)
    else:
        ctx.render_frame(np.concatenate([[1920, 1080, f, 0], view]).astype(np.float32))
ctx.synchronize()
