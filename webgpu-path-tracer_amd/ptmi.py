"""ctypes binding of include/ptmi.h (libptmi.so).  No fallbacks: if the library is missing or a call
fails, this raises — there is no CPU path in the product."""
import ctypes
import os

import numpy as np

from . import _build

_HERE = os.path.dirname(os.path.abspath(__file__))

BUF = {"spheres": 1, "quads": 2, "triangles": 5, "meshes": 6, "transforms": 7, "materials": 8, "bvh": 9}

# every symbol include/ptmi.h declares
SYMBOLS = [
    "ptmi_version", "ptmi_status_string", "ptmi_last_error", "ptmi_create", "ptmi_create_multi", "ptmi_destroy", "ptmi_default_params",
    "ptmi_set_params", "ptmi_get_params", "ptmi_upload", "ptmi_resize", "ptmi_clear_framebuffer", "ptmi_set_shard",
    "ptmi_render_frame", "ptmi_render", "ptmi_synchronize", "ptmi_prepare", "ptmi_read_framebuffer", "ptmi_write_framebuffer", "ptmi_reduce_framebuffer",
    "ptmi_framebuffer_device_ptr", "ptmi_bind_framebuffer", "ptmi_stream", "ptmi_resolve_rgba8", "ptmi_set_counters",
    "ptmi_set_timing", "ptmi_get_stats", "ptmi_reset_stats", "ptmi_trace", "ptmi_math_eval", "ptmi_selftest", "ptmi_build_bvh",
    "ptmi_build_bvh_sah", "ptmi_build_bvh_device", "ptmi_build_scene_bvh", "ptmi_read_scene_buffer", "ptmi_obj_parse", "ptmi_free",
    "ptmi_device_count", "ptmi_reduce_info", "ptmi_reload_tuning", "ptmi_build_scene_bvh_sah", "ptmi_scene_bvh_info", "ptmi_build_bvh_sah_device",
]


class Params(ctypes.Structure):
    _fields_ = [
        ("num_samples", ctypes.c_int32), ("max_bounces", ctypes.c_int32), ("stratify", ctypes.c_int32),
        ("importance_sampling", ctypes.c_int32), ("stack_size", ctypes.c_int32), ("background", ctypes.c_float * 3),
        ("fov_degrees", ctypes.c_float), ("frames_in_flight", ctypes.c_int32), ("tmin", ctypes.c_float), ("light_mix", ctypes.c_float), ("reserved", ctypes.c_int32 * 3),
    ]


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "rays", "paths", "node_visits", "tri_tests", "sphere_tests", "quad_tests", "mat_fetches", "frames",
        "intersect_launches", "shade_launches", "bvh_node_visits", "bvh_mat_fetches")] + [
        (n, ctypes.c_double) for n in ("render_ms", "intersect_ms", "shade_ms", "other_ms", "prims_ms", "bvh_ms", "generate_ms", "accumulate_ms")] + [
        (n, ctypes.c_uint64) for n in ("generate_launches", "accumulate_launches", "devices")] + [("tail_ms", ctypes.c_double), ("tail_launches", ctypes.c_uint64)] + [
        (n, ctypes.c_uint64) for n in ("reduce_mode", "peer_links", "placement_sets")] + [("placement_ms", ctypes.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("normal", "<f4", 3), ("front_face", "<i4"), ("material", "<f4", 16)])


class PtmiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("ptmi status %d: %s" % (status, message))
        self.status = status


_lib = None
_libs = {}


def lib_path():
    return _build.LIB


def load_library(build=False, path=None):
    """dlopen libptmi.so (optionally building it first).  Raises if it is not there.  `path`: another build of the library next to the default
    one (the tests' fault-injection build, _build.build_testhooks) — pass the result to Context(..., lib=...)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and path in _libs:
        return _libs[path]
    if build:
        _build.build_lib()
    explicit = path is not None
    path = path or os.environ.get("PTMI_LIB") or _build.LIB  # PTMI_LIB: an A/B build (_build.build_variant), as for the N-API addon
    if not os.path.exists(path):
        raise OSError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)" % os.path.basename(path))
    L = ctypes.CDLL(path)
    vp, i32, u32, sz, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p
    L.ptmi_version.restype = i32
    L.ptmi_status_string.restype = ctypes.c_char_p
    L.ptmi_status_string.argtypes = [i32]
    L.ptmi_last_error.restype = ctypes.c_char_p
    L.ptmi_last_error.argtypes = [vp]
    L.ptmi_create.argtypes = [ctypes.POINTER(vp), i32]
    L.ptmi_create_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int), i32]
    L.ptmi_prepare.argtypes = [vp]
    L.ptmi_destroy.argtypes = [vp]
    L.ptmi_destroy.restype = None
    L.ptmi_default_params.argtypes = [ctypes.POINTER(Params)]
    L.ptmi_default_params.restype = None
    L.ptmi_set_params.argtypes = [vp, ctypes.POINTER(Params)]
    L.ptmi_get_params.argtypes = [vp, ctypes.POINTER(Params)]
    L.ptmi_upload.argtypes = [vp, i32, fp, sz]
    L.ptmi_resize.argtypes = [vp, i32, i32]
    L.ptmi_clear_framebuffer.argtypes = [vp]
    L.ptmi_set_shard.argtypes = [vp, i32, i32, i32]
    L.ptmi_render_frame.argtypes = [vp, fp]
    L.ptmi_render.argtypes = [vp, fp, u32, u32]
    L.ptmi_synchronize.argtypes = [vp]
    L.ptmi_read_framebuffer.argtypes = [vp, fp, sz]
    L.ptmi_write_framebuffer.argtypes = [vp, fp, sz]
    L.ptmi_reduce_framebuffer.argtypes = [vp]
    L.ptmi_framebuffer_device_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.ptmi_bind_framebuffer.argtypes = [vp, vp, sz]
    L.ptmi_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.ptmi_resolve_rgba8.argtypes = [vp, ctypes.c_float, fp, sz]
    L.ptmi_set_counters.argtypes = [vp, i32]
    L.ptmi_set_timing.argtypes = [vp, i32]
    L.ptmi_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.ptmi_reset_stats.argtypes = [vp]
    L.ptmi_trace.argtypes = [vp, sz, fp, fp, fp]
    L.ptmi_math_eval.argtypes = [vp, i32, sz, fp, fp, fp]
    if hasattr(L, "ptmi_selftest"):  # (an older A/B build loaded through PTMI_LIB may lack the newest test hooks)
        L.ptmi_selftest.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    L.ptmi_build_bvh.argtypes = [sz, fp, fp, i32, fp, fp]
    L.ptmi_build_bvh_sah.argtypes = [sz, fp, fp, i32, fp, fp, ctypes.POINTER(sz)]
    L.ptmi_build_bvh_device.argtypes = [vp, sz, fp, fp, i32, fp, fp]
    L.ptmi_build_scene_bvh.argtypes = [vp]
    L.ptmi_build_scene_bvh_sah.argtypes = [vp]
    L.ptmi_scene_bvh_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.ptmi_build_bvh_sah_device.argtypes = [vp, sz, fp, fp, i32, fp, fp, ctypes.POINTER(sz)]
    L.ptmi_read_scene_buffer.argtypes = [vp, i32, fp, sz]
    L.ptmi_obj_parse.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.ptmi_free.argtypes = [vp]
    L.ptmi_free.restype = None
    L.ptmi_device_count.restype = i32
    L.ptmi_reduce_info.restype = ctypes.c_char_p
    L.ptmi_reduce_info.argtypes = [vp]
    L.ptmi_reload_tuning.argtypes = [vp]
    if explicit:
        _libs[path] = L
    else:
        _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def default_params(**kw):
    p = Params()
    load_library().ptmi_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if k == "background":
            p.background[:] = list(v)
        else:
            setattr(p, k, v)
    return p


class NativeHost:
    """Host-side natives (no GPU): plug into host.scene.build_bvh(native=...)."""

    def __init__(self):
        self.lib = load_library()

    def build_bvh(self, bmin, bmax, prim_type=2):
        bmin = np.ascontiguousarray(bmin, np.float64)
        bmax = np.ascontiguousarray(bmax, np.float64)
        n = bmin.shape[0]
        nodes = np.zeros((max(2 * n - 1, 0), 12), np.float32)
        order = np.zeros(n, np.int64)
        st = self.lib.ptmi_build_bvh(n, _ptr(bmin), _ptr(bmax), prim_type, _ptr(nodes), _ptr(order))
        if st != 0:
            raise PtmiError(st, "ptmi_build_bvh failed")
        return nodes, order


    def build_bvh_sah(self, bmin, bmax, prim_type=2):
        """The reference's binned-SAH builder (lib/BVH/bvhNode.js:108-283, dead code there), opt-in."""
        bmin = np.ascontiguousarray(bmin, np.float64)
        bmax = np.ascontiguousarray(bmax, np.float64)
        n = bmin.shape[0]
        nodes = np.zeros((max(2 * n - 1, 0), 12), np.float32)
        order = np.zeros(n, np.int64)
        count = ctypes.c_size_t()
        st = self.lib.ptmi_build_bvh_sah(n, _ptr(bmin), _ptr(bmax), prim_type, _ptr(nodes), _ptr(order), ctypes.byref(count))
        if st != 0:
            raise PtmiError(st, "ptmi_build_bvh_sah failed")
        return nodes[: count.value].copy(), order

    def parse_obj(self, text):
        """ObjReader.parse in native code: {vertices, normals} float32 arrays (lib/primitives/objReader.js grammar)."""
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        pv, pn, nv, nn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        st = self.lib.ptmi_obj_parse(data, len(data), ctypes.byref(pv), ctypes.byref(nv), ctypes.byref(pn), ctypes.byref(nn))
        if st != 0:
            raise PtmiError(st, "ptmi_obj_parse failed")
        try:
            v = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (nv.value,)).copy() if nv.value else np.zeros(0, np.float32)
            n = np.ctypeslib.as_array(ctypes.cast(pn, ctypes.POINTER(ctypes.c_float)), (nn.value,)).copy() if nn.value else np.zeros(0, np.float32)
        finally:
            self.lib.ptmi_free(pv)
            self.lib.ptmi_free(pn)
        return {"vertices": v, "normals": n}

    def load_obj(self, path):
        with open(path, "rb") as f:
            return self.parse_obj(f.read())


class Context:
    """One integrator context (mirrors the reference's Renderer+WebGPU pair for the hot path).  `device` is a GPU index, or a
    list of them for a multi-device context (ptmi_create_multi: tiles sharded across the GPUs, one RCCL reduce on read-back)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        h = ctypes.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (ctypes.c_int * len(device))(*[int(d) for d in device])
            st = self.lib.ptmi_create_multi(ctypes.byref(h), ids, len(device))
        else:
            st = self.lib.ptmi_create(ctypes.byref(h), device)
        if st != 0:
            raise PtmiError(st, self.lib.ptmi_last_error(None).decode())
        self.h = h
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.ptmi_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st):
        if st != 0:
            raise PtmiError(st, self.lib.ptmi_last_error(self.h).decode())

    def set_params(self, params=None, **kw):
        p = params if params is not None else default_params(**kw)
        self._ck(self.lib.ptmi_set_params(self.h, ctypes.byref(p)))
        return p

    def get_params(self):
        p = Params()
        self._ck(self.lib.ptmi_get_params(self.h, ctypes.byref(p)))
        return p

    def upload(self, which, array):
        a = np.ascontiguousarray(array)
        if a.dtype not in (np.float32, np.int32):
            raise TypeError("buffers are float32 (int32 for meshes)")
        self._ck(self.lib.ptmi_upload(self.h, BUF[which] if isinstance(which, str) else which, _ptr(a), a.nbytes))

    def upload_scene(self, buffers):
        for k in ("spheres", "quads", "triangles", "meshes", "transforms", "materials", "bvh"):
            a = buffers[k]
            self.upload(k, np.asarray(a, np.int32 if k == "meshes" else np.float32))

    def resize(self, w, h):
        self._ck(self.lib.ptmi_resize(self.h, w, h))
        self.width, self.height = w, h

    def clear(self):
        self._ck(self.lib.ptmi_clear_framebuffer(self.h))

    def set_shard(self, rank, world, tile=64):
        self._ck(self.lib.ptmi_set_shard(self.h, rank, world, tile))

    def render_frame(self, uniforms20):
        u = np.ascontiguousarray(uniforms20, np.float32)
        assert u.size == 20
        self._ck(self.lib.ptmi_render_frame(self.h, _ptr(u)))

    def render(self, view16, first_frame, n_frames):
        v = np.ascontiguousarray(view16, np.float32)
        assert v.size == 16
        self._ck(self.lib.ptmi_render(self.h, _ptr(v), first_frame, n_frames))

    def synchronize(self):
        self._ck(self.lib.ptmi_synchronize(self.h))

    def prepare(self):
        """Validate the uploaded scene and build the device-side digests now (otherwise the first render does it)."""
        self._ck(self.lib.ptmi_prepare(self.h))

    def read_framebuffer(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._ck(self.lib.ptmi_read_framebuffer(self.h, _ptr(out), out.nbytes))
        return out

    def reduce_info(self):
        """One line about how this context sums its devices' buffers (RCCL, add kernel, or the FALLBACK after an RCCL failure)."""
        return self.lib.ptmi_reduce_info(self.h).decode()

    def reload_tuning(self):
        """Re-read the PTMI_* tuning variables (they are read once, at creation)."""
        self._ck(self.lib.ptmi_reload_tuning(self.h))

    def reduce_framebuffer(self):
        """The one collective of a multi-device render (sum of the per-device buffers on the first device); a sync on one device."""
        self._ck(self.lib.ptmi_reduce_framebuffer(self.h))

    def write_framebuffer(self, fb):
        a = np.ascontiguousarray(fb, np.float32)
        self._ck(self.lib.ptmi_write_framebuffer(self.h, _ptr(a), a.nbytes))

    def framebuffer_device_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._ck(self.lib.ptmi_framebuffer_device_ptr(self.h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def bind_framebuffer(self, dev_ptr, nbytes):
        self._ck(self.lib.ptmi_bind_framebuffer(self.h, ctypes.c_void_p(dev_ptr), nbytes))

    def stream(self):
        p = ctypes.c_void_p()
        self._ck(self.lib.ptmi_stream(self.h, ctypes.byref(p)))
        return p.value

    def resolve_rgba8(self, frame_num):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._ck(self.lib.ptmi_resolve_rgba8(self.h, float(frame_num), _ptr(out), out.nbytes))
        return out

    def set_counters(self, on):
        self._ck(self.lib.ptmi_set_counters(self.h, int(on)))

    def set_timing(self, mode):
        """0/False off, 1/True every kernel, 2 / 3 / 4 / 5 only k_bvh / k_shade / k_generate / k_accumulate."""
        self._ck(self.lib.ptmi_set_timing(self.h, int(mode)))

    def stats(self):
        s = Stats()
        self._ck(self.lib.ptmi_get_stats(self.h, ctypes.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._ck(self.lib.ptmi_reset_stats(self.h))

    def build_bvh(self, bmin, bmax, prim_type=2):
        """ptmi_build_bvh_device: the median-split build on this context's GPU; same result as NativeHost.build_bvh."""
        bmin = np.ascontiguousarray(bmin, np.float64)
        bmax = np.ascontiguousarray(bmax, np.float64)
        n = bmin.shape[0]
        nodes = np.zeros((max(2 * n - 1, 0), 12), np.float32)
        order = np.zeros(n, np.int64)
        self._ck(self.lib.ptmi_build_bvh_device(self.h, n, _ptr(bmin), _ptr(bmax), prim_type, _ptr(nodes), _ptr(order)))
        return nodes, order

    def build_scene_bvh(self, sah=False):
        """ptmi_build_scene_bvh: Scene.create_bvh() on the GPU over the uploaded (unordered) triangles, meshes and transforms; nothing comes back.
        sah=True: the reference's other builder (lib/BVH/bvhNode.js:108-283), the opt-in."""
        self._ck((self.lib.ptmi_build_scene_bvh_sah if sah else self.lib.ptmi_build_scene_bvh)(self.h))

    def scene_bvh_info(self):
        """{nodes, depth, on_device} of the scene's BVH (ptmi_scene_bvh_info)."""
        n, d, o = ctypes.c_uint64(), ctypes.c_int32(), ctypes.c_int32()
        self._ck(self.lib.ptmi_scene_bvh_info(self.h, ctypes.byref(n), ctypes.byref(d), ctypes.byref(o)))
        return {"nodes": n.value, "depth": d.value, "on_device": bool(o.value)}

    def build_bvh_sah(self, bmin, bmax, prim_type=2):
        """ptmi_build_bvh_sah_device: the binned-SAH build on this context's GPU; same result as NativeHost.build_bvh_sah."""
        bmin = np.ascontiguousarray(bmin, np.float64)
        bmax = np.ascontiguousarray(bmax, np.float64)
        n = bmin.shape[0]
        nodes = np.zeros((max(2 * n - 1, 0), 12), np.float32)
        order = np.zeros(n, np.int64)
        count = ctypes.c_size_t()
        self._ck(self.lib.ptmi_build_bvh_sah_device(self.h, n, _ptr(bmin), _ptr(bmax), prim_type, _ptr(nodes), _ptr(order), ctypes.byref(count)))
        return nodes[: count.value].copy(), order

    def read_scene_buffer(self, which, count):
        """Test hook: the context's triangles ('triangles', count = number of triangles) or BVH rows ('bvh', count = number of nodes) as an array."""
        out = np.empty((count, 24 if which == "triangles" else 12), np.float32)
        self._ck(self.lib.ptmi_read_scene_buffer(self.h, BUF[which], _ptr(out), out.nbytes))
        return out

    def trace(self, rays6, rng=None):
        r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
        n = r.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        g = None if rng is None else np.ascontiguousarray(rng, np.uint32).copy()
        self._ck(self.lib.ptmi_trace(self.h, n, _ptr(r), None if g is None else _ptr(g), _ptr(out)))
        return out, g

    def selftest(self, which):
        """Exhaustive device-side check of a unary shortcut against the IEEE operation: (mismatches, first bad argument bits)."""
        n, first = ctypes.c_uint64(), ctypes.c_uint32()
        self._ck(self.lib.ptmi_selftest(self.h, which, ctypes.byref(n), ctypes.byref(first)))
        return n.value, first.value

    def math_eval(self, fn, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        yy = None if y is None else np.ascontiguousarray(y, np.float32)
        self._ck(self.lib.ptmi_math_eval(self.h, fn, x.size, _ptr(x), None if yy is None else _ptr(yy), _ptr(out)))
        return out
