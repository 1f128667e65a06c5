"""Host-side scene builder: the Python mirror of the reference's ``lib/`` classes.

Same class and method names as the reference (``Scene``, ``Sphere``, ``Quad``, ``Mesh``, ``Transform``,
``Camera``, ``ObjReader``, ``build_bvh``) and byte-identical output arrays — the eight typed arrays of
SURVEY.md §8a-0 that cross the device boundary.  The arithmetic follows the reference's JS (doubles,
rounded to f32 where the reference stores into a Float32Array); tests/test_host_buffers.py compares the
results with goldens captured from the reference's own code.

Triangles are kept as arrays (one row per triangle) instead of one object per triangle, and the BVH
build can be delegated to the native builder in libptmi_host.so (``native=True``) for large meshes.
"""
import math

import numpy as np

from .glmatrix import mat4, vec3


# ----------------------------------------------------------------------------- lib/transform.js
class Transform:
    def __init__(self):
        self.translateM = mat4.create()
        self.scaleM = mat4.create()
        self.rotationM = mat4.create()
        self.M = mat4.create()
        self.modelMatrix = mat4.create()
        self.invModelMatrix = mat4.create()

    def getTransform(self):  # lib/transform.js:38-40
        return np.concatenate([self.modelMatrix, self.invModelMatrix]).astype(np.float32)

    def update(self, *transforms):  # lib/transform.js:42-58
        if transforms:
            mat4.identity(self.M)
            for t in transforms:
                mat4.mul(self.M, t, self.M)
            mat4.identity(self.modelMatrix)
            mat4.mul(self.modelMatrix, self.M, self.modelMatrix)
            mat4.invert(self.invModelMatrix, self.modelMatrix)

    def translate(self, x, y, z):  # lib/transform.js:60-68
        return mat4.fromTranslation(self.translateM, np.array([x, y, z], np.float32))

    def scale(self, sx, sy, sz):  # lib/transform.js:70-79
        return mat4.fromScaling(self.scaleM, np.array([sx, sy, sz], np.float32))

    def rotate(self, theta, axis):  # lib/transform.js:81-87
        return mat4.fromRotation(self.rotationM, theta, np.array(axis, np.float32))


# ----------------------------------------------------------------------------- lib/BVH/AABB.js
def _pad(bmin, bmax):
    """AABB.pad() (lib/BVH/AABB.js:35-51) on arrays of boxes, in float64."""
    delta = 0.0001 / 2
    thin = (bmax - bmin) < delta
    bmax = np.where(thin, bmax + delta, bmax)
    bmin = np.where(thin, bmin - delta, bmin)
    return bmin, bmax


# ----------------------------------------------------------------------------- lib/primitives/*.js
class Sphere:
    type = 0

    def __init__(self, center, r, global_id, local_id, material_id):  # lib/primitives/sphere.js:6-30
        self.global_id, self.local_id = global_id, local_id
        self.data = [center[0], center[1], center[2], r, global_id, local_id, material_id, -1]
        self.transform = Transform()


class Quad:
    type = 1

    def __init__(self, Q, u, v, global_id, local_id, material_id):  # lib/primitives/quad.js:5-36
        self.global_id, self.local_id = global_id, local_id
        n, normal, w = vec3.create(), vec3.create(), vec3.create()
        vec3.cross(n, u, v)
        vec3.normalize(normal, n)
        D = vec3.dot(normal, Q)
        temp = vec3.dot(n, n)
        vec3.set(w, float(n[0]) / temp, float(n[1]) / temp, float(n[2]) / temp)
        self.data = [
            Q[0], Q[1], Q[2], -1,
            u[0], u[1], u[2], local_id,
            v[0], v[1], v[2], global_id,
            float(normal[0]), float(normal[1]), float(normal[2]), D,
            float(w[0]), float(w[1]), float(w[2]), material_id,
        ]
        self.transform = Transform()


class Mesh:
    """lib/primitives/mesh.js + lib/primitives/triangle.js, one row per triangle."""

    type = 2

    def __init__(self, data, offset, id, mesh_id, local_id, material_id):
        v = np.asarray(data["vertices"], np.float32)
        nrm = np.asarray(data["normals"], np.float32)
        T = v.size // 9  # mesh.js:24 loops i < vertices.length / 9
        self.numTriangle = T
        self.verts = v[: T * 9].reshape(T, 3, 3)  # object space A,B,C
        nr = nrm[: T * 9].reshape(T, 3, 3)
        # triangle.js:42-52 row layout
        d = np.full((T, 24), -1.0, np.float32)
        d[:, 0:3], d[:, 4:7], d[:, 8:11] = self.verts[:, 0], self.verts[:, 1], self.verts[:, 2]
        d[:, 12:15], d[:, 16:19], d[:, 20:23] = nr[:, 0], nr[:, 1], nr[:, 2]
        d[:, 19] = local_id + np.arange(T)
        d[:, 23] = mesh_id
        self.tri_data = d
        self.mesh = [T, offset, id, material_id]  # mesh.js:58-63
        self.global_id = id
        self.transform = Transform()
        self.bmin = self.bmax = None

    def calc_bbox(self, transform):  # triangle.js:27-39 for every triangle
        m = [float(x) for x in transform.modelMatrix]
        P = self.verts.astype(np.float64)  # (T,3,3)
        x, y, z = P[..., 0], P[..., 1], P[..., 2]
        w = m[3] * x + m[7] * y + m[11] * z + m[15]
        w = np.where((w == 0) | np.isnan(w), 1.0, w)  # `w = w || 1.0`
        wx = ((m[0] * x + m[4] * y + m[8] * z + m[12]) / w).astype(np.float32)
        wy = ((m[1] * x + m[5] * y + m[9] * z + m[13]) / w).astype(np.float32)
        wz = ((m[2] * x + m[6] * y + m[10] * z + m[14]) / w).astype(np.float32)
        W = np.stack([wx, wy, wz], axis=-1).astype(np.float64)  # world-space, f32 values
        self.bmin, self.bmax = _pad(W.min(axis=1), W.max(axis=1))


# ----------------------------------------------------------------------------- lib/primitives/objReader.js
_DEC = None


def _js_number(tok):
    """JavaScript Number(token): '' -> 0, decimal literal, [+-]Infinity, 0x/0o/0b integer; anything else NaN."""
    global _DEC
    import re

    if _DEC is None:
        _DEC = re.compile(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?\Z")
    tok = tok.strip(" \t\r\n\v\f")
    if tok == "":
        return 0.0
    if len(tok) > 2 and tok[0] == "0" and tok[1] in "xXoObB":
        try:
            return float(int(tok[2:], {"x": 16, "o": 8, "b": 2}[tok[1].lower()]))
        except ValueError:
            return float("nan")
    if tok in ("Infinity", "+Infinity"):
        return float("inf")
    if tok == "-Infinity":
        return float("-inf")
    return float(tok) if _DEC.match(tok) else float("nan")


class ObjReader:
    @staticmethod
    def parse(text):  # lib/primitives/objReader.js:15-68
        import re

        verts, norms, vidx, nidx = [], [], [], []
        for raw in text.split("\n"):
            line = raw.strip()
            if line.startswith("#"):
                continue
            elif line.startswith("v "):
                verts.append([_js_number(t) for t in line.split(" ")[1:]])
            elif line.startswith("f "):
                toks = re.split(r"[\s/]+", line)[1:]
                vidx.extend(_js_number(t) - 1 for i, t in enumerate(toks) if i % 3 == 0)
                nidx.extend(_js_number(t) - 1 for i, t in enumerate(toks) if i % 3 == 2)
            elif line.startswith("vn "):
                norms.append([_js_number(t) for t in line.split(" ")[1:]])

        def flat(rows, idx):  # indexArray.map(v => rows[v]).flat(1): an invalid index is `undefined` -> NaN
            out = []
            for d in idx:
                if d != d or d < 0 or d != int(d) or d >= len(rows):
                    out.append(float("nan"))
                else:
                    out.extend(rows[int(d)])
            return np.asarray(out, np.float64).astype(np.float32)

        return {"vertices": flat(verts, vidx), "normals": flat(norms, nidx)}

    @staticmethod
    def load_model(path):
        with open(path, "r") as f:
            return ObjReader.parse(f.read())


# ----------------------------------------------------------------------------- lib/BVH/*
def build_bvh(bmin, bmax, prim_type=2, native=None):
    """Median-split BVH, pre-order flattened (lib/BVH/bvhNode.js:21-101, lib/BVH/bvhBuilder.js:6-54).

    bmin/bmax: (N,3) float64 primitive boxes (already padded).  Returns (nodes f32 (2N-1,12), order)
    where ``order[k]`` is the input index of the primitive that ends up at position k (the reference
    reorders its ``objs`` array in place, lib/BVH/bvhNode.js:57-61, and uploads triangles in that order).
    """
    bmin = np.ascontiguousarray(bmin, np.float64)
    bmax = np.ascontiguousarray(bmax, np.float64)
    N = bmin.shape[0]
    if native is not None:
        return native.build_bvh(bmin, bmax, prim_type)
    if N == 0:
        return np.zeros((0, 12), np.float32), np.zeros(0, np.int64)
    order = np.arange(N)
    nn = 2 * N - 1
    nodes = np.zeros((nn, 12), np.float64)
    left = np.full(nn, -1, np.int64)
    right = np.full(nn, -1, np.int64)
    counter = [0]

    def gen(start, end):
        nid = counter[0]
        counter[0] += 1
        idx = order[start : end + 1]
        lo = np.minimum(bmin[idx].min(axis=0), 1e30)
        hi = np.maximum(bmax[idx].max(axis=0), -1e30)
        ext = hi - lo
        axis = 0
        if ext[1] > ext[0]:
            axis = 1
        if ext[2] > ext[axis]:
            axis = 2
        span = end - start
        row = nodes[nid]
        row[0:3], row[4:7] = lo, hi
        if span <= 0:
            row[3], row[7], row[8], row[9], row[11] = -1, prim_type, start, end - start + 1, 0
        else:
            perm = np.argsort(bmin[idx, axis], kind="stable")
            order[start : end + 1] = idx[perm]
            mid = start + span // 2
            left[nid] = gen(start, mid)
            right[nid] = gen(mid + 1, end)
            row[3], row[7], row[8], row[9], row[11] = right[nid], -1, -1, -1, axis
        return nid

    import sys

    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    gen(0, N - 1)
    # populate_links (bvhNode.js:76-93): skip link = next node in pre-order when the box is missed
    stack = [(0, -1)]
    while stack:
        n, nxt = stack.pop()
        nodes[n, 10] = nxt
        if left[n] >= 0:
            stack.append((right[n], nxt))
            stack.append((left[n], right[n]))
    return nodes.astype(np.float32), order


# ----------------------------------------------------------------------------- lib/scene.js
class Scene:
    """Same surface as lib/scene.js; subclasses fill create_spheres/create_quads/create_meshes."""

    def __init__(self):
        self.mats = []
        self.material_id = 0
        self.material_dict = {}
        self.global_id = 0
        self.sphere_id = 0
        self.quad_id = 0
        self.triangle_id = 0
        self.mesh_id = 0
        self.triangle_offset = 0
        self.spheres, self.quads, self.meshes, self.lights = [], [], [], []
        self.objs = []
        self.mesh_data = {}
        self.bvh_array = np.zeros((0, 12), np.float32)
        self.tri_data = np.zeros((0, 24), np.float32)
        self.create_spheres()
        self.create_quads()

    # hooks (lib/scene.js:36-251 hard-codes these)
    def create_spheres(self):
        self.objs.extend(self.spheres)

    def create_quads(self):
        self.objs.extend(self.quads)

    def init_mesh_data(self):
        pass

    def create_meshes(self):
        self._finish_meshes()

    # helpers used by the hooks
    def add_sphere(self, center, r, material_id):
        s = Sphere(center, r, self.global_id, self.sphere_id, material_id)
        self.global_id += 1
        self.sphere_id += 1
        self.spheres.append(s)
        return s

    def add_quad(self, Q, u, v, material_id):
        q = Quad(Q, u, v, self.global_id, self.quad_id, material_id)
        self.global_id += 1
        self.quad_id += 1
        self.quads.append(q)
        return q

    def add_mesh(self, data, material_id):  # lib/scene.js:168-174
        m = Mesh(data, self.triangle_offset, self.global_id, self.mesh_id, self.triangle_id, material_id)
        self.global_id += 1
        self.mesh_id += 1
        self.triangle_id += m.numTriangle
        self.triangle_offset += m.numTriangle
        self.meshes.append(m)
        return m

    def _finish_meshes(self):  # lib/scene.js:245-248
        for m in self.meshes:
            m.calc_bbox(m.transform)
        self.tri_data = np.concatenate([m.tri_data for m in self.meshes]) if self.meshes else np.zeros((0, 24), np.float32)
        self.objs.extend(self.meshes)

    def add_material(self, name, material_type, color, specularColor, emissionColor, percentSpecular, roughness, eta=float("nan")):
        # lib/scene.js:261-273; an omitted `eta` is `undefined` -> NaN in the Float32Array
        self.material_dict[name] = self.material_id
        self.mats.append([
            color[0], color[1], color[2], -1,
            specularColor[0], specularColor[1], specularColor[2], -1,
            emissionColor[0], emissionColor[1], emissionColor[2], percentSpecular,
            roughness, eta, material_type, -1,
        ])
        self.material_id += 1
        return self.material_id - 1

    def create_bvh(self, native=None, sah=False):  # lib/scene.js:253-259
        """sah=True swaps in the reference's other builder (BVH.generate_bvh_heirarchy_SAH, bvhNode.js:108-283 — dead
        code there; native only): an opt-in, the reference's renderer always uses the median split."""
        if not self.meshes:
            return
        bmin = np.concatenate([m.bmin for m in self.meshes])
        bmax = np.concatenate([m.bmax for m in self.meshes])
        if sah:
            if native is None:
                raise ValueError("the SAH builder exists in native code only: pass native=NativeHost()")
            self.bvh_array, order = native.build_bvh_sah(bmin, bmax, 2)
        else:
            self.bvh_array, order = build_bvh(bmin, bmax, 2, native=native)
        self.tri_data = self.tri_data[order]

    def get_bvh(self):
        return np.ascontiguousarray(self.bvh_array, np.float32).reshape(-1)

    def get_triangles(self):
        return np.ascontiguousarray(self.tri_data, np.float32).reshape(-1)

    def get_meshes(self):
        return np.asarray([m.mesh for m in self.meshes], np.int32).reshape(-1)

    def get_materials(self):
        return np.asarray(self.mats, np.float64).astype(np.float32).reshape(-1)

    def get_spheres(self):
        return np.asarray([s.data for s in self.spheres], np.float64).astype(np.float32).reshape(-1)

    def get_quads(self):
        return np.asarray([q.data for q in self.quads], np.float64).astype(np.float32).reshape(-1)

    def get_lights(self):
        return np.asarray([q.data for q in self.lights], np.float64).astype(np.float32).reshape(-1)

    def get_transforms(self):  # lib/scene.js:275-282
        if not self.objs:
            return np.zeros(0, np.float32)
        return np.concatenate([o.transform.getTransform() for o in self.objs]).astype(np.float32)

    def buffers(self, native=None, sah=False):
        """The renderer.js:78-87 call order, returning the seven uploadable arrays."""
        self.init_mesh_data()
        self.create_meshes()
        out = {
            "meshes": self.get_meshes(),
            "spheres": self.get_spheres(),
            "quads": self.get_quads(),
            "materials": self.get_materials(),
            "transforms": self.get_transforms(),
        }
        self.create_bvh(native=native, sah=sah)
        out["bvh"] = self.get_bvh()
        out["triangles"] = self.get_triangles()
        return out

    def buffers_unbuilt(self):
        """The same arrays BEFORE create_bvh: the triangles in mesh order and no BVH — what ptmi_build_scene_bvh (the build on the GPU, over the
        uploaded triangles) starts from.  Upload them (an empty `bvh`), then Context.build_scene_bvh()."""
        self.init_mesh_data()
        self.create_meshes()
        return {
            "meshes": self.get_meshes(), "spheres": self.get_spheres(), "quads": self.get_quads(), "materials": self.get_materials(),
            "transforms": self.get_transforms(), "triangles": self.get_triangles(), "bvh": np.zeros(0, np.float32),
        }


# ----------------------------------------------------------------------------- lib/camera.js
class Camera:
    def __init__(self, canvas=None):
        self.viewMatrix = mat4.create()
        self.eye, self.center, self.up, self.direction = vec3.create(), vec3.create(), vec3.create(), vec3.create()
        self.zoomSpeed, self.moveSpeed, self.keypressMoveSpeed = 0.1, 0.01, 0.1
        self.MOVING = 0
        self.keyPress = 0
        self.rotateAngle = 0

    def set_camera(self, eye=None, center=None, up=None):  # lib/camera.js:25-33
        eye = self.eye if eye is None else eye
        center = self.center if center is None else center
        up = self.up if up is None else up
        vec3.set(self.eye, eye[0], eye[1], eye[2])
        vec3.set(self.center, center[0], center[1], center[2])
        vec3.set(self.up, up[0], up[1], up[2])
        vec3.subtract(self.direction, self.eye, self.center)
        mat4.targetTo(self.viewMatrix, eye, center, up)  # raw arguments, as the reference does

    def zoom(self, delta):  # lib/camera.js:35-42
        s = math.copysign(1.0, delta) if delta != 0 else 0.0
        e = [float(v) for v in self.eye]
        d = [float(v) for v in self.direction]
        for i in range(3):
            self.eye[i] = e[i] + d[i] * self.zoomSpeed * s
        self.set_camera()

    def move(self, oldCoord, newCoord):  # lib/camera.js:44-53
        dX = (newCoord[0] - oldCoord[0]) * math.pi / 180 * self.moveSpeed
        self.rotateAngle = dX
        vec3.rotateY(self.eye, self.eye, [0, 0, 0], dX)
        self.set_camera()

    def _shift(self, axis, d):  # lib/camera.js:55-74: eye and center move together along x (left/right) or y (up/down)
        e = [float(v) for v in self.eye]
        c = [float(v) for v in self.center]
        e[axis] += d
        c[axis] += d
        vec3.set(self.eye, e[0], e[1], e[2])
        vec3.set(self.center, c[0], c[1], c[2])
        self.set_camera()

    def moveLeft(self):
        self._shift(0, +self.keypressMoveSpeed)

    def moveRight(self):
        self._shift(0, -self.keypressMoveSpeed)

    def moveUp(self):
        self._shift(1, -self.keypressMoveSpeed)

    def moveDown(self):
        self._shift(1, +self.keypressMoveSpeed)

    def dispatch(self, kind, **ev):
        """The listeners Camera.MoveCamera installs on the canvas / document (lib/camera.js:77-131), as one entry point:
        kind in {"wheel", "mousedown", "mousemove", "mouseup", "keydown"}; ev carries the DOM event's fields."""
        if kind == "wheel":
            self.zoom(ev.get("deltaY") or ev.get("detail") or ev.get("wheelDelta") or 0)
            self.keyPress = 1
        elif kind == "mousedown":
            if ev.get("button", 0) == 0:
                self._drag_from = [ev["x"], ev["y"]]  # stays the anchor for the whole drag: the reference never updates oldCoord
                self._dragging = True
        elif kind == "mousemove":
            if getattr(self, "_dragging", False):
                self.move(self._drag_from, [ev["x"], ev["y"]])
                self.MOVING = 1
        elif kind == "mouseup":
            self._dragging = False
            self.MOVING = 0
        elif kind == "keydown":
            fn = {"ArrowLeft": self.moveLeft, "ArrowRight": self.moveRight, "ArrowUp": self.moveUp, "ArrowDown": self.moveDown}.get(ev.get("key"))
            if fn:
                fn()
                self.keyPress = 1


def uniforms_array(width, height, frame_num, reset_buffer, view_matrix):
    """renderer.js:70-77 + flattenToFloat32Array (renderer.js:265-278): the 20-float uniform block."""
    return np.concatenate([[width, height, frame_num, reset_buffer], np.asarray(view_matrix, np.float32)]).astype(np.float32)
