"""Canonical scenes built through the Scene API, plus golden-buffer loading.

``default``  the reference scene exactly as lib/scene.js ships it (19 spheres, 8 quads, rotated cube)
``c1``       Cornell box + mirror sphere + glass sphere, no triangles           (BASELINE configs[0])
``c2``       Cornell box + monkey_968.obj, scale 0.6, translate (0,-0.4,0)      (BASELINE configs[1])
``c2m``      Cornell box + 2 meshes + fog/glass sphere pair (multi-mesh / volume coverage)
"""
import json
import math
import os

import numpy as np

from .host.scene import Camera, ObjReader, Scene

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BUFFER_NAMES = ("spheres", "quads", "triangles", "meshes", "transforms", "materials", "bvh")


def _cornell_materials(sc):  # lib/scene.js:107-113
    sc.add_material("red", 0, [0.75, 0.1, 0.1], [0.75, 0.1, 0.1], [0, 0, 0], 0.05, 0.95, 0)
    sc.add_material("green", 0, [0.05, 0.55, 0.05], [0.05, 0.55, 0.05], [0, 0, 0], 0.05, 0.95, 0)
    sc.add_material("blue", 0, [0.05, 0.05, 0.55], [0.05, 0.05, 0.55], [0, 0, 0], 0.05, 0.95, 0)
    sc.add_material("white", 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.05, 0.95, 0)
    sc.add_material("glossywhite", 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.3, 0.1, 0)
    sc.add_material("black", 0, [0.2, 0.2, 0.2], [0.2, 0.2, 0.2], [0, 0, 0], 0.05, 0.95, 0)
    sc.add_material("glass", 1, [0.95, 0.95, 0.95], [0, 0, 0], [0, 0, 0], 0, 0, 0)


def _cornell_quads(sc):  # walls of lib/scene.js:128-132 + the light of SURVEY.md §8d-C1
    _cornell_materials(sc)
    d = sc.material_dict
    sc.add_quad([-0.35, 0.9999, -0.3], [0.7, 0, 0], [0, 0, 0.6], sc.add_material("light", 0, [0, 0, 0], [0, 0, 0], [10, 10, 10], 0, 0, 0))
    sc.add_quad([-1, -1, -1], [2, 0, 0], [0, 2, 0], d["black"])
    sc.add_quad([-1, -1, 1], [0, 0, -2], [0, 2, 0], d["red"])
    sc.add_quad([1, -1, -1], [0, 0, 2], [0, 2, 0], d["green"])
    sc.add_quad([-1, 1, -1], [2, 0, 0], [0, 0, 2], d["white"])
    sc.add_quad([1, -1, -1], [-2, 0, 0], [0, 0, 2], d["glossywhite"])
    sc.lights.append(sc.quads[0])
    sc.objs.extend(sc.quads)


class CornellScene(Scene):
    """Cornell walls + light; spheres and meshes supplied by callables."""

    def __init__(self, spheres=None, meshes=None):
        self._spheres_fn, self._meshes_fn = spheres, meshes
        super().__init__()

    def create_spheres(self):
        self.add_material("default", 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0)
        if self._spheres_fn:
            self._spheres_fn(self)
        self.objs.extend(self.spheres)

    def create_quads(self):
        _cornell_quads(self)

    def create_meshes(self):
        if self._meshes_fn:
            self._meshes_fn(self)
        self._finish_meshes()


def c1_scene():
    def spheres(sc):
        sc.add_sphere([-0.5, -0.7, -0.5], 0.3, sc.add_material("mirror_ball", 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0, 0))
        sc.add_sphere([0.6, -0.75, 0.5], 0.25, sc.add_material("glass_ball", 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5))

    return CornellScene(spheres=spheres)


DRAGON_MAT = ("dragonMat", 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5)  # lib/scene.js:166


def mesh_scene(mesh_data, scale=(1, 1, 1), rotate=None, translate=(0, 0, 0), material=DRAGON_MAT):
    """Cornell box + one mesh with scale -> rotate -> translate (the order of lib/scene.js:216-220)."""

    def meshes(sc):
        m = sc.add_mesh(mesh_data, sc.add_material(*material))
        ops = [m.transform.scale(*scale)]
        if rotate is not None:
            ops.append(m.transform.rotate(rotate[0], rotate[1]))
        ops.append(m.transform.translate(*translate))
        m.transform.update(*ops)

    return CornellScene(meshes=meshes)


def c2_scene(monkey_data):
    def meshes(sc):
        m = sc.add_mesh(monkey_data, sc.add_material(*DRAGON_MAT))
        m.transform.update(m.transform.scale(0.6, 0.6, 0.6), m.transform.translate(0, -0.4, 0))

    return CornellScene(meshes=meshes)


def c2m_scene(ico_data, cube_data):
    def spheres(sc):
        sc.add_sphere([-0.45, -0.6, 0.45], 0.3, sc.add_material("fog", 3, [0.56, 0.93, 0.56], [0, 0, 0], [0, 0, 0], 0.00001, -1 / 4, 0))
        sc.add_sphere([-0.45, -0.6, 0.45], 0.3, sc.add_material("gg4t", 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5))

    def meshes(sc):
        a = sc.add_mesh(ico_data, sc.add_material(*DRAGON_MAT))
        b = sc.add_mesh(cube_data, sc.add_material("box2", 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0.05, 1.5))
        a.transform.update(a.transform.scale(0.35, 0.35, 0.35), a.transform.rotate(math.pi / 4, [0, 1, 0]), a.transform.translate(0.45, -0.64, 0))
        b.transform.update(b.transform.scale(0.2, 0.3, 0.2), b.transform.rotate(-math.pi / 4, [1, 1, 0]), b.transform.translate(-0.1, -0.55, -0.4))

    return CornellScene(spheres=spheres, meshes=meshes)


class DefaultScene(Scene):
    """lib/scene.js:36-251 as shipped."""

    def __init__(self, cube_data):
        self._cube = cube_data
        super().__init__()

    def create_spheres(self):  # lib/scene.js:36-103
        self.add_material("default", 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0)
        temp = [0.94, 0.70, 0.75]
        green = [0.56, 0.93, 0.56]

        def pair(c, r, fog_col, density, eta):
            self.add_sphere(c, r, self.add_material("fog", 3, fog_col, [0, 0, 0], [0, 0, 0], 0.00001, density, 0))
            self.add_sphere(c, r, self.add_material("gg4t", 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, eta))

        for c, r in (([-0.3, -0.65, 0.3], 0.35), ([-0.3, -0.05, 0.3], 0.25), ([-0.3, 0.3, 0.3], 0.1), ([-0.3, 0.45, 0.3], 0.05)):
            pair(c, r, green, -1 / 4, 1.5)
        pair([0.5, -0.65, -0.2], 0.35, [0.52, 0.8, 0.92], -1 / 7, 1)
        self.add_sphere([0.5, 0.1, 0.2], 0.2, self.add_material("gg4t", 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5))
        for c, r in (([1.3, -0.65, 0.3], 0.35), ([1.3, -0.05, 0.3], 0.25), ([1.3, 0.3, 0.3], 0.1), ([1.3, 0.45, 0.3], 0.05)):
            pair(c, r, temp, -1 / 10, 1)
        self.objs.extend(self.spheres)

    def create_quads(self):  # lib/scene.js:105-162
        _cornell_materials(self)
        d = self.material_dict
        self.add_quad([-1, 1, -1], [3, 0, 0], [0, 0, 2], self.add_material("tWall", 0, [0, 0, 0], [0, 0, 0], [2, 2, 2], 0, 0))
        self.add_quad([-1, -1, -1], [3, 0, 0], [0, 2, 0], d["black"])
        self.add_quad([-1, -1, 1], [0, 0, -2], [0, 2, 0], d["red"])
        self.add_quad([2, -1, -1], [0, 0, 2], [0, 2, 0], d["green"])
        self.add_quad([-1, 1, -1], [3, 0, 0], [0, 0, 2], d["white"])
        self.add_quad([2, -1, -1], [-3, 0, 0], [0, 0, 2], d["glossywhite"])
        self.add_quad([100, -1, -100], [-200, 0, 0], [0, 0, 200], d["white"])
        self.add_quad([2, -1, 1], [-3, 0, 0], [0, 2, 0], self.add_material("fWall", 0, [0.15, 0.15, 0.15], [0, 0, 0], [0, 0, 0], 0, 0, 0))
        self.lights.append(self.quads[0])
        self.objs.extend(self.quads)

    def create_meshes(self):  # lib/scene.js:164-251
        self.add_material(*DRAGON_MAT)
        m = self.add_mesh(self._cube, self.add_material("glassBox", 0, [0.95, 0.95, 0.95], [0, 0, 0], [0, 0, 0], 0, 0, 2.5))
        m.transform.update(m.transform.rotate(math.pi / 10, [0, 1, 0]))
        self._finish_meshes()


def camera_view(eye, center, up=(0, 1, 0)):
    cam = Camera()
    cam.set_camera(list(eye), list(center), list(up))
    return cam.viewMatrix.copy()


CAMERAS = {"default": ([0.5, 0, 2.5], [0.5, 0, 0]), "cornell": ([0, 0, 2.5], [0, 0, 0]), "oblique": ([1.2, 0.4, 2.1], [0.1, -0.2, 0])}


def golden_manifest():
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        return json.load(f)


def golden_buffers(name):
    """Host buffers captured from the reference's own JS (oracle/capture/capture.mjs)."""
    man = golden_manifest()[name]
    out = {}
    for k in BUFFER_NAMES:
        if k in man and man[k]["stored"]:
            dt = np.int32 if man[k]["dtype"] == "i32" else np.float32
            out[k] = np.fromfile(os.path.join(GOLDEN_DIR, f"{name}_{k}.bin"), dtype=dt)
        else:
            out[k] = np.zeros(0, np.int32 if k == "meshes" else np.float32)
    return out


# ----------------------------------------------------------------------------- procedural stand-ins
def _pcg_floats(seed, n):
    """n draws of the reference's PCG stream (shaders/common.wgsl:7-12) in numpy — the generator's only randomness."""
    out = np.empty(n, np.float64)
    s = np.uint64(seed)
    m32 = np.uint64(0xFFFFFFFF)
    for i in range(n):
        s = (s * np.uint64(747796405) + np.uint64(2891336453)) & m32
        w = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & m32
        out[i] = float((w >> np.uint64(22)) ^ w) / 4294967296.0
    return out


def dragon_class_mesh(n_tris=871414, seed=1):
    """Deterministic stand-in for stanfordDragon.obj (absent from the container, .MISSING_LARGE_BLOBS): a displaced
    (2,3) torus-knot tube tessellated to EXACTLY n_tris triangles, smooth per-vertex normals, longest extent 1.
    Returns the ObjReader layout {vertices, normals} (9 floats per triangle each)."""
    quads = (n_tris + 1) // 2
    nv = max(8, int(round(math.sqrt(quads / 2.0))))  # around the tube
    nu = (quads + nv - 1) // nv                      # along the knot
    r = _pcg_floats(seed, 24)
    u = np.linspace(0, 2 * math.pi, nu, endpoint=False)[:, None]
    v = np.linspace(0, 2 * math.pi, nv, endpoint=False)[None, :]
    p, q = 2.0, 3.0
    cu = np.concatenate([(2 + np.cos(q * u)) * np.cos(p * u), (2 + np.cos(q * u)) * np.sin(p * u), np.sin(q * u)], axis=1)  # (nu,3)
    du = 2 * math.pi / nu
    t = np.roll(cu, -1, 0) - np.roll(cu, 1, 0)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    ref = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(t, ref)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(t, n1)
    rad = 0.45 * (1.0 + sum(0.06 * (0.5 + r[3 * k]) * np.sin((k + 2) * (3 * u + 2 * v) * (1 if k % 2 else -1) + 6.283 * r[3 * k + 1]) * np.cos((k + 1) * v + 6.283 * r[3 * k + 2]) for k in range(6)))
    P = cu[:, None, :] + rad[..., None] * (np.cos(v)[..., None] * n1[:, None, :] + np.sin(v)[..., None] * n2[:, None, :])  # (nu,nv,3)
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    P = (P - (lo + hi) / 2) / (hi - lo).max()
    du_ = np.roll(P, -1, 0) - np.roll(P, 1, 0)
    dv_ = np.roll(P, -1, 1) - np.roll(P, 1, 1)
    N = np.cross(du_, dv_)
    N /= np.linalg.norm(N, axis=2, keepdims=True)
    i0 = np.arange(nu)[:, None]
    j0 = np.arange(nv)[None, :]
    i1, j1 = (i0 + 1) % nu, (j0 + 1) % nv

    def tri(a, b, c):
        return np.stack([a, b, c], axis=2)  # (nu,nv,3 verts,3)

    def grid(A):
        return tri(A[i0, j0], A[i1, j0], A[i1, j1]), tri(A[i0, j0], A[i1, j1], A[i0, j1])

    (Pa, Pb), (Na, Nb) = grid(P), grid(N)
    V = np.stack([Pa, Pb], axis=2).reshape(-1, 9)[:n_tris]
    Nn = np.stack([Na, Nb], axis=2).reshape(-1, 9)[:n_tris]
    return {"vertices": V.astype(np.float32).reshape(-1), "normals": Nn.astype(np.float32).reshape(-1)}


def c3_scene(n_tris=871414, seed=1):
    """BASELINE configs[2]: Cornell walls/light + dragon-class mesh with the reference's dragon transform
    (lib/scene.js:216-220: scale 1.1, rotate pi/4 about y, translate (0.65,-0.64,0)) and dragonMat."""
    return mesh_scene(dragon_class_mesh(n_tris, seed), scale=(1.1, 1.1, 1.1), rotate=(math.pi / 4, [0, 1, 0]), translate=(0.65, -0.64, 0))


def _grid_surface(P, wrap_u, wrap_v, flip=False):
    """Triangulate a (nu, nv, 3) grid of points (2 triangles per cell) with smooth per-vertex normals from central
    differences.  Returns (T,3,3) vertices and normals."""
    nu, nv = P.shape[:2]
    du = (np.roll(P, -1, 0) - np.roll(P, 1, 0)) if wrap_u else np.gradient(P, axis=0)
    dv = (np.roll(P, -1, 1) - np.roll(P, 1, 1)) if wrap_v else np.gradient(P, axis=1)
    N = np.cross(du, dv)
    ln = np.linalg.norm(N, axis=2, keepdims=True)
    N = N / np.where(ln == 0, 1.0, ln)
    if flip:
        N = -N
    iu = np.arange(nu if wrap_u else nu - 1)[:, None]
    iv = np.arange(nv if wrap_v else nv - 1)[None, :]
    i1, j1 = (iu + 1) % nu, (iv + 1) % nv
    order = (lambda a, b, c: (a, c, b)) if flip else (lambda a, b, c: (a, b, c))

    def tris(A):
        t1 = np.stack(order(A[iu, iv], A[i1, iv], A[i1, j1]), axis=2)
        t2 = np.stack(order(A[iu, iv], A[i1, j1], A[iu, j1]), axis=2)
        return np.stack([t1, t2], axis=2).reshape(-1, 3, 3)

    return tris(P), tris(N)


def _as_obj(parts, n_tris):
    V = np.concatenate([p[0] for p in parts])[:n_tris]
    N = np.concatenate([p[1] for p in parts])[:n_tris]
    assert V.shape[0] == n_tris, (V.shape[0], n_tris)
    return {"vertices": V.astype(np.float32).reshape(-1), "normals": N.astype(np.float32).reshape(-1)}


def sponza_class_mesh(n_tris=262267, seed=2):
    """Stand-in for sponzaAtrium.obj (absent): an interior — a room shell with inward normals, two rows of fluted
    columns and a wavy drape — tessellated to EXACTLY n_tris triangles.  Fits [-1,1]^3; meant to be viewed from inside."""
    r = _pcg_floats(seed, 16)
    parts = []
    per = n_tris / 10.0
    # room shell: 6 finely tessellated faces, normals pointing inwards
    k = max(2, int(math.sqrt(per * 3 / 6 / 2)))
    a = np.linspace(-1, 1, k)
    A, B = np.meshgrid(a, a, indexing="ij")
    for axis, sign in ((0, -1), (0, 1), (1, -1), (1, 1), (2, -1), (2, 1)):
        P = np.zeros((k, k, 3))
        P[..., axis] = sign
        P[..., (axis + 1) % 3] = A
        P[..., (axis + 2) % 3] = B
        parts.append(_grid_surface(P, False, False, flip=(sign > 0)))
    # columns: fluted cylinders
    ncol = 12
    nu = max(8, int(math.sqrt(per * 6 / ncol / 2 * 2)))
    nv = max(4, nu // 2)
    u = np.linspace(0, 2 * math.pi, nu, endpoint=False)[:, None]
    h = np.linspace(-1, 0.55, nv)[None, :]
    for c in range(ncol):
        cx = -0.62 if c % 2 == 0 else 0.62
        cz = -0.85 + (c // 2) * 0.34
        rad = 0.07 * (1 + 0.08 * np.cos(12 * u)) * (1 + 0.15 * np.exp(-((h - 0.5) / 0.05) ** 2) + 0.2 * np.exp(-((h + 0.95) / 0.05) ** 2))
        P = np.stack([cx + rad * np.cos(u), np.broadcast_to(h, rad.shape), cz + rad * np.sin(u)], axis=-1)
        parts.append(_grid_surface(P, True, False, flip=True))
    # drape: fills the remaining triangle budget exactly
    have = sum(p[0].shape[0] for p in parts)
    rest = max(2, n_tris - have)
    m = int(math.sqrt(rest / 2)) + 2
    s = np.linspace(-0.55, 0.6, m)[:, None]   # height
    t = np.linspace(-0.9, 0.9, m)[None, :]    # along the left wall
    P = np.stack([-0.92 + 0.04 * np.sin(9 * s + 6.283 * r[0]) * np.cos(5 * t + 6.283 * r[1]) + 0 * t, s + 0 * t, t + 0 * s], axis=-1)
    parts.append(_grid_surface(P, False, False))
    return _as_obj(parts, n_tris)


def buddha_class_mesh(n_tris=1087716, seed=3):
    """Stand-in for buddha.obj (absent): a displaced, stacked-lobe solid of revolution tessellated to EXACTLY n_tris
    triangles with smooth normals; longest extent 1."""
    r = _pcg_floats(seed, 24)
    quads = (n_tris + 1) // 2
    nv = max(8, int(round(math.sqrt(quads / 2.0))))
    nu = (quads + (nv - 1) - 1) // (nv - 1) + 1
    u = np.linspace(0, 2 * math.pi, nu, endpoint=False)[:, None]
    v = np.linspace(0.02, math.pi - 0.02, nv)[None, :]
    prof = 0.30 + 0.12 * np.cos(3 * v) + 0.05 * np.cos(7 * v + 1.0)
    bump = 1.0 + sum(0.03 * (0.5 + r[3 * k]) * np.sin((k + 3) * u + 6.283 * r[3 * k + 1]) * np.sin((2 * k + 2) * v + 6.283 * r[3 * k + 2]) for k in range(6))
    rad = prof * bump * np.sin(v) ** 0.8
    P = np.stack([rad * np.cos(u), -0.5 * np.cos(v) + 0 * u, rad * np.sin(u)], axis=-1)
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    P = (P - (lo + hi) / 2) / (hi - lo).max()
    return _as_obj([_grid_surface(P, True, False, flip=True)], n_tris)


class InteriorScene(Scene):
    """configs[3]: the mesh IS the room; one emissive quad under the ceiling (the light for importance sampling)."""

    def __init__(self, mesh_data, material):
        self._mesh, self._mat = mesh_data, material
        super().__init__()

    def create_spheres(self):
        self.add_material("default", 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0)
        self.objs.extend(self.spheres)

    def create_quads(self):
        self.add_quad([-0.3, 0.98, -0.3], [0.6, 0, 0], [0, 0, 0.6], self.add_material("light", 0, [0, 0, 0], [0, 0, 0], [12, 12, 12], 0, 0, 0))
        self.lights.append(self.quads[0])
        self.objs.extend(self.quads)

    def create_meshes(self):
        self.add_mesh(self._mesh, self.add_material(*self._mat))
        self._finish_meshes()


def c4_scene(n_tris=262267, seed=2):
    """BASELINE configs[3]: sponza-class interior, camera inside (CAMERAS['interior'])."""
    return InteriorScene(sponza_class_mesh(n_tris, seed), ("stone", 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.05, 0.95, 0))


def c5_scene(n_tris=1087716, seed=3):
    """BASELINE configs[4]: Cornell box + buddha-class mesh with a GLASS material (eta 1.5); run with
    importance_sampling=1, 16 bounces, stack_size 24."""
    return mesh_scene(buddha_class_mesh(n_tris, seed), scale=(1.3, 1.3, 1.3), translate=(0.0, -0.35, 0.0),
                      material=("glassBuddha", 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5))


CAMERAS["interior"] = ([0.0, -0.35, 0.92], [0.0, -0.2, -0.5])
