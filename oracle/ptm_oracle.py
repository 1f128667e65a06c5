"""ctypes binding of oracle/libptm_oracle.so — TEST INFRASTRUCTURE (see ptm_oracle.cpp's header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports this module.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libptm_oracle.so")


class _Scene(ctypes.Structure):
    _fields_ = []
    for _n in ("spheres", "quads", "triangles", "meshes", "transforms", "materials", "bvh"):
        _fields_ += [(_n, ctypes.c_void_p), ("n_" + _n, ctypes.c_int32)]


class _Params(ctypes.Structure):
    _fields_ = [("num_samples", ctypes.c_int32), ("max_bounces", ctypes.c_int32), ("stratify", ctypes.c_int32),
                ("importance_sampling", ctypes.c_int32), ("stack_size", ctypes.c_int32), ("background", ctypes.c_float * 3),
                ("fov_factor", ctypes.c_float), ("ray_tmin", ctypes.c_float), ("light_mix", ctypes.c_float)]


class _Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("rays", "node_visits", "tri_tests", "sphere_tests", "quad_tests", "mat_fetches", "paths")]


HIT_DTYPE = np.dtype([("hit", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("normal", "<f4", 3), ("front_face", "<i4"), ("material", "<f4", 16)])
_STRIDES = {"spheres": 8, "quads": 20, "triangles": 24, "meshes": 4, "transforms": 32, "materials": 16, "bvh": 12}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ptm_oracle.cpp")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "ptmi_math.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fov_factor(fov_degrees=60.0):
    """main.wgsl:7 `1 / tan(60 * (PI / 180) / 2)`: const-expression folded in f64, rounded once to f32."""
    return np.float32(1.0 / math.tan(float(fov_degrees) * (math.pi / 180.0) / 2.0))


def _scene(buffers, keep):
    s = _Scene()
    for k, stride in _STRIDES.items():
        a = np.ascontiguousarray(buffers[k], np.int32 if k == "meshes" else np.float32).reshape(-1)
        keep.append(a)
        setattr(s, k, a.ctypes.data if a.size else None)
        setattr(s, "n_" + k, a.size // stride)
    return s


def _params(num_samples=1, max_bounces=100, stratify=0, importance_sampling=0, stack_size=20, background=(0.0, 1.0, 1.0), fov_degrees=60.0,
            tmin=0.000001, light_mix=0.2):
    p = _Params()
    p.num_samples, p.max_bounces, p.stratify = num_samples, max_bounces, int(stratify)
    p.importance_sampling, p.stack_size = int(importance_sampling), stack_size
    p.background[:] = list(background)
    p.fov_factor = float(fov_factor(fov_degrees))
    p.ray_tmin, p.light_mix = float(np.float32(tmin)), float(np.float32(light_mix))
    return p


def render(buffers, width, height, view16, first_frame=1, n_frames=1, reset_first=0, framebuffer=None, threads=0,
           shard=(0, 1, 64), pixel_range=(0, -1), **params):
    """Frames first_frame..first_frame+n_frames-1 accumulated into `framebuffer` (H,W,4 f32; zeros if None)."""
    keep = []
    s, p = _scene(buffers, keep), _params(**params)
    fb = np.zeros((height, width, 4), np.float32) if framebuffer is None else np.ascontiguousarray(framebuffer, np.float32).copy()
    un = np.zeros(20, np.float32)
    un[0], un[1], un[2], un[3] = width, height, first_frame, reset_first
    un[4:] = np.asarray(view16, np.float32)
    st = _Stats()
    L = lib()
    L.ptmo_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    rc = L.ptmo_render(ctypes.byref(s), ctypes.byref(p), _ptr(un), first_frame, n_frames, _ptr(fb), ctypes.byref(st), threads,
                       shard[0], shard[1], shard[2], pixel_range[0], pixel_range[1])
    if rc != 0:
        raise RuntimeError("ptmo_render failed (%d)" % rc)
    return fb, {n: getattr(st, n) for n, _ in _Stats._fields_}


def hit_scene(buffers, rays6, rng=None, **params):
    keep = []
    s, p = _scene(buffers, keep), _params(**params)
    r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
    out = np.zeros(r.shape[0], HIT_DTYPE)
    g = None if rng is None else np.ascontiguousarray(rng, np.uint32).copy()
    st = _Stats()
    L = lib()
    L.ptmo_hit_scene.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.ptmo_hit_scene(ctypes.byref(s), ctypes.byref(p), r.shape[0], _ptr(r), None if g is None else _ptr(g), _ptr(out), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError("ptmo_hit_scene failed (%d)" % rc)
    return out, g, {n: getattr(st, n) for n, _ in _Stats._fields_}


def hit_bruteforce(buffers, rays6, **params):
    keep = []
    s, p = _scene(buffers, keep), _params(**params)
    r = np.ascontiguousarray(rays6, np.float32).reshape(-1, 6)
    out = np.zeros(r.shape[0], HIT_DTYPE)
    L = lib()
    L.ptmo_hit_bruteforce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.ptmo_hit_bruteforce(ctypes.byref(s), ctypes.byref(p), r.shape[0], _ptr(r), _ptr(out))
    if rc != 0:
        raise RuntimeError("ptmo_hit_bruteforce failed (%d)" % rc)
    return out


def rand(seed, n):
    f = np.zeros(n, np.float32)
    st = np.zeros(n, np.uint32)
    L = lib()
    L.ptmo_rand.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.ptmo_rand(seed, n, _ptr(f), _ptr(st))
    return f, st


def math_eval(fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    L = lib()
    L.ptmo_math.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.ptmo_math(fn, x.size, _ptr(x), None if yy is None else _ptr(yy), _ptr(out))
    if rc != 0:
        raise ValueError("unknown math function id %d" % fn)
    return out


def resolve_rgba8(fb, frame_num):
    """The display pass on the CPU: (H,W,4) f32 framebuffer sum -> (H,W,4) uint8."""
    a = np.ascontiguousarray(fb, np.float32)
    out = np.empty(a.shape[:-1] + (4,), np.uint8)
    L = lib()
    L.ptmo_resolve_rgba8.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    rc = L.ptmo_resolve_rgba8(_ptr(a), a.size // 4, float(frame_num), _ptr(out))
    if rc != 0:
        raise RuntimeError("ptmo_resolve_rgba8 failed (%d)" % rc)
    return out


def max_threads():
    return lib().ptmo_max_threads()
