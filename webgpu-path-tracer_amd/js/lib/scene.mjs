// lib/scene.mjs — the host-side scene builder for Node: same classes, method names and output bytes as the
// reference's lib/ (Scene, Sphere, Quad, Mesh, Transform, Camera, ObjReader, build_bvh).  Triangles are kept in
// typed arrays (one row per triangle) and the BVH build can use the native builder of the addon.
import { vec3, mat4 } from '../glmatrix.mjs';

export class Transform {   // lib/transform.js
  constructor() {
    this.translateV = vec3.create(); this.scaleV = vec3.create(); this.rotationAxis = vec3.create();
    this.translateM = mat4.create(); this.scaleM = mat4.create(); this.rotationM = mat4.create(); this.M = mat4.create();
    this.modelMatrix = mat4.create(); this.invModelMatrix = mat4.create();
  }
  getTransform() { return [...this.modelMatrix, ...this.invModelMatrix]; }
  update(...transforms) {
    if (transforms.length > 0) {
      mat4.identity(this.M);
      for (let i = 0; i < transforms.length; i++) mat4.mul(this.M, transforms[i], this.M);
      mat4.identity(this.modelMatrix);
      mat4.mul(this.modelMatrix, this.M, this.modelMatrix);
      mat4.invert(this.invModelMatrix, this.modelMatrix);
    }
  }
  translate(x, y, z) { vec3.set(this.translateV, x, y, z); mat4.fromTranslation(this.translateM, this.translateV); return this.translateM; }
  scale(sx, sy, sz) { vec3.set(this.scaleV, sx, sy, sz); mat4.fromScaling(this.scaleM, this.scaleV); return this.scaleM; }
  rotate(theta, axis) { vec3.set(this.rotationAxis, axis[0], axis[1], axis[2]); mat4.fromRotation(this.rotationM, theta, this.rotationAxis); return this.rotationM; }
}

export class Sphere {   // lib/primitives/sphere.js
  constructor(center, r, global_id, local_id, material_id) {
    this.type = 0; this.global_id = global_id; this.local_id = local_id;
    this.data = [center[0], center[1], center[2], r, global_id, local_id, material_id, -1];
    this.transform = new Transform();
  }
}

export class Quad {   // lib/primitives/quad.js
  constructor(Q, u, v, global_id, local_id, material_id) {
    this.type = 1; this.global_id = global_id; this.local_id = local_id;
    const n = vec3.create(), normal = vec3.create(), w = vec3.create();
    vec3.cross(n, u, v);
    vec3.normalize(normal, n);
    const D = vec3.dot(normal, Q);
    const temp = vec3.dot(n, n);
    vec3.set(w, n[0] / temp, n[1] / temp, n[2] / temp);
    this.data = [Q[0], Q[1], Q[2], -1, u[0], u[1], u[2], local_id, v[0], v[1], v[2], global_id, normal[0], normal[1], normal[2], D, w[0], w[1], w[2], material_id];
    this.transform = new Transform();
  }
}

const PAD = 0.0001 / 2;   // lib/BVH/AABB.js:35-51

export class Mesh {   // lib/primitives/mesh.js + triangle.js, one row per triangle
  constructor(data, offset, id, mesh_id, local_id, material_id) {
    this.type = 2;
    const T = Math.floor(data.vertices.length / 9);
    this.numTriangle = T;
    this.vertices = data.vertices;
    this.tri_data = new Float32Array(T * 24).fill(-1);
    for (let i = 0; i < T; i++) {
      const d = this.tri_data, o = i * 24, s = i * 9;
      for (let k = 0; k < 3; k++) {
        d[o + k] = data.vertices[s + k]; d[o + 4 + k] = data.vertices[s + 3 + k]; d[o + 8 + k] = data.vertices[s + 6 + k];
        d[o + 12 + k] = data.normals[s + k]; d[o + 16 + k] = data.normals[s + 3 + k]; d[o + 20 + k] = data.normals[s + 6 + k];
      }
      d[o + 19] = local_id + i;
      d[o + 23] = mesh_id;
    }
    this.mesh = [T, offset, id, material_id];
    this.global_id = id;
    this.transform = new Transform();
    this.bmin = null; this.bmax = null;
  }
  calc_bbox(transform) {   // triangle.js:27-39 per triangle
    const T = this.numTriangle, m = transform.modelMatrix, v = this.vertices;
    this.bmin = new Float64Array(3 * T); this.bmax = new Float64Array(3 * T);
    const p = vec3.create(), q = [0, 0, 0];
    for (let i = 0; i < T; i++) {
      const lo = [Infinity, Infinity, Infinity], hi = [-Infinity, -Infinity, -Infinity];
      for (let c = 0; c < 3; c++) {
        q[0] = v[i * 9 + 3 * c]; q[1] = v[i * 9 + 3 * c + 1]; q[2] = v[i * 9 + 3 * c + 2];
        vec3.transformMat4(p, q, m);
        for (let k = 0; k < 3; k++) { lo[k] = Math.min(lo[k], p[k]); hi[k] = Math.max(hi[k], p[k]); }
      }
      for (let k = 0; k < 3; k++) {
        if (hi[k] - lo[k] < PAD) { hi[k] += PAD; lo[k] -= PAD; }
        this.bmin[3 * i + k] = lo[k]; this.bmax[3 * i + k] = hi[k];
      }
    }
  }
}

export class ObjReader {   // lib/primitives/objReader.js:10-68
  static parse(text) {
    const lines = text.split('\n');
    let vertexArray = [], indexArray = [], normalArray = [], normalIndex = [];
    for (let i = 0; i < lines.length; i++) {
      const line = lines[i].trim();
      if (line.startsWith('#')) continue;
      else if (line.startsWith('v ')) vertexArray.push(line.split(' ').slice(1).map(Number));
      else if (line.startsWith('f ')) {
        const temp = line.split(/[\s/]+/).slice(1);
        indexArray.push(...temp.filter((v, k) => k % 3 == 0).map(Number).map((v) => v - 1));
        normalIndex.push(...temp.filter((v, k) => k % 3 == 2).map(Number).map((v) => v - 1));
      } else if (line.startsWith('vn ')) normalArray.push(line.split(' ').slice(1).map(Number));
    }
    const N = normalIndex.map((k) => normalArray[k]).flat(1);   // an invalid index stays `undefined` -> NaN
    const V = indexArray.map((k) => vertexArray[k]).flat(1);
    return { vertices: new Float32Array(V), normals: new Float32Array(N) };
  }
  // `native` = the addon (ptmi.node): same result from the C++ parser of libptmi.so, much faster on large files
  static async load_model(path, native = null) {
    const file = await fetch(path);
    const text = await file.text();
    return native ? native.parseObj(text) : ObjReader.parse(text);
  }
}

// lib/BVH/bvhNode.js:21-101 + lib/BVH/bvhBuilder.js:6-54 on box arrays.  Returns {nodes: Float32Array((2n-1)*12),
// order: Int32Array(n)}; `native` (the addon) is used when given, else the same algorithm runs in JS.
export function build_bvh(bmin, bmax, prim_type = 2, native = null) {
  const n = bmin.length / 3;
  if (native) return native.buildBVH(bmin, bmax, prim_type);
  if (n === 0) return { nodes: new Float32Array(0), order: new Int32Array(0) };
  const order = Array.from({ length: n }, (_, i) => i);
  const nn = 2 * n - 1, nodes = new Float64Array(nn * 12), left = new Int32Array(nn).fill(-1), right = new Int32Array(nn).fill(-1);
  let counter = 0;
  const gen = (start, end) => {
    const id = counter++;
    const lo = [1e30, 1e30, 1e30], hi = [-1e30, -1e30, -1e30];
    for (let i = start; i <= end; i++) for (let k = 0; k < 3; k++) { lo[k] = Math.min(bmin[3 * order[i] + k], lo[k]); hi[k] = Math.max(bmax[3 * order[i] + k], hi[k]); }
    const ext = [hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]];
    let axis = 0;
    if (ext[1] > ext[0]) axis = 1;
    if (ext[2] > ext[axis]) axis = 2;
    const o = id * 12, span = end - start;
    nodes.set([lo[0], lo[1], lo[2]], o); nodes.set([hi[0], hi[1], hi[2]], o + 4);
    if (span <= 0) {
      nodes[o + 3] = -1; nodes[o + 7] = prim_type; nodes[o + 8] = start; nodes[o + 9] = end - start + 1; nodes[o + 11] = 0;
    } else {
      const sub = order.slice(start, end + 1);
      sub.sort((a, b) => bmin[3 * a + axis] - bmin[3 * b + axis]);   // stable (V8 >= 7.0), like the reference's
      for (let i = start, j = 0; i <= end; i++, j++) order[i] = sub[j];
      const mid = start + Math.floor(span / 2);
      left[id] = gen(start, mid); right[id] = gen(mid + 1, end);
      nodes[o + 3] = right[id]; nodes[o + 7] = -1; nodes[o + 8] = -1; nodes[o + 9] = -1; nodes[o + 11] = axis;
    }
    return id;
  };
  gen(0, n - 1);
  const st = [[0, -1]];
  while (st.length) {
    const [k, nxt] = st.pop();
    nodes[k * 12 + 10] = nxt;
    if (left[k] >= 0) { st.push([right[k], nxt]); st.push([left[k], right[k]]); }
  }
  return { nodes: new Float32Array(nodes), order: Int32Array.from(order) };
}

export class Scene {   // lib/scene.js surface; subclasses or callers fill create_spheres / create_quads / create_meshes
  constructor() {
    this.mats = []; this.material_id = 0; this.material_dict = {};
    this.global_id = 0; this.sphere_id = 0; this.quad_id = 0; this.triangle_id = 0; this.mesh_id = 0; this.triangle_offset = 0;
    this.spheres = []; this.quads = []; this.meshes = []; this.objs = []; this.lights = [];
    this.mesh_data = {}; this.bvh_array = new Float32Array(0); this.tri_data = new Float32Array(0);
    this.native = null;
    this.create_spheres();
    this.create_quads();
  }
  create_spheres() { this.objs.push(...this.spheres); }
  create_quads() { this.objs.push(...this.quads); }
  async init_mesh_data() {}
  create_meshes() { this.finish_meshes(); }

  add_sphere(center, r, material_id) { const s = new Sphere(center, r, this.global_id++, this.sphere_id++, material_id); this.spheres.push(s); return s; }
  add_quad(Q, u, v, material_id) { const q = new Quad(Q, u, v, this.global_id++, this.quad_id++, material_id); this.quads.push(q); return q; }
  add_mesh(data, material_id) {   // lib/scene.js:168-174
    const m = new Mesh(data, this.triangle_offset, this.global_id++, this.mesh_id++, this.triangle_id, material_id);
    this.triangle_id += m.numTriangle; this.triangle_offset += m.numTriangle;
    this.meshes.push(m);
    return m;
  }
  finish_meshes() {   // lib/scene.js:245-248
    this.meshes.forEach((m) => m.calc_bbox(m.transform));
    const total = this.meshes.reduce((a, m) => a + m.numTriangle, 0);
    this.tri_data = new Float32Array(total * 24);
    let o = 0;
    for (const m of this.meshes) { this.tri_data.set(m.tri_data, o); o += m.tri_data.length; }
    this.objs.push(...this.meshes);
  }
  add_material(name, material_type, color, specularColor, emissionColor, percentSpecular, roughness, eta) {   // lib/scene.js:261-273
    this.material_dict[name] = this.material_id;
    this.mats.push([color[0], color[1], color[2], -1, specularColor[0], specularColor[1], specularColor[2], -1,
      emissionColor[0], emissionColor[1], emissionColor[2], percentSpecular, roughness, eta, material_type, -1]);
    return this.material_id++;
  }
  create_bvh() {   // lib/scene.js:253-259
    const total = this.tri_data.length / 24;
    if (total === 0) return;
    const bmin = new Float64Array(3 * total), bmax = new Float64Array(3 * total);
    let o = 0;
    for (const m of this.meshes) { bmin.set(m.bmin, o); bmax.set(m.bmax, o); o += m.bmin.length; }
    // this.useSAH (opt-in, native only): the reference's other builder, BVH.generate_bvh_heirarchy_SAH (bvhNode.js:108-283)
    if (this.useSAH && !this.native) throw new Error('useSAH needs the native host (scene.native = loadNative())');
    const r = this.useSAH ? this.native.buildBVHSAH(bmin, bmax, 2) : build_bvh(bmin, bmax, 2, this.native);
    this.bvh_array = r.nodes;
    const sorted = new Float32Array(this.tri_data.length);
    for (let k = 0; k < total; k++) sorted.set(this.tri_data.subarray(r.order[k] * 24, r.order[k] * 24 + 24), k * 24);
    this.tri_data = sorted;
  }
  get_transforms() { const t = []; this.objs.forEach((i) => t.push(...i.transform.getTransform())); return new Float32Array(t); }
  get_bvh() { return new Float32Array(this.bvh_array); }
  get_triangles() { return new Float32Array(this.tri_data); }
  get_meshes() { return new Int32Array(this.meshes.map((m) => m.mesh).flat()); }
  get_materials() { return new Float32Array(this.mats.flat()); }
  get_spheres() { return new Float32Array(this.spheres.map((s) => s.data).flat()); }
  get_quads() { return new Float32Array(this.quads.map((q) => q.data).flat()); }
  get_lights() { return new Float32Array(this.lights.map((l) => l.data).flat()); }
}

export class Camera {   // lib/camera.js (interaction handlers are attached only when a canvas is given)
  constructor(canvas = null, doc = (typeof document !== 'undefined' ? document : null)) {
    this.viewMatrix = mat4.create();
    this.eye = vec3.create(); this.center = vec3.create(); this.up = vec3.create(); this.direction = vec3.create();
    this.rotateAngle = 0; this.zoomSpeed = 0.1; this.moveSpeed = 0.01; this.keypressMoveSpeed = 0.1;
    this.MOVING = 0; this.keyPress = 0;
    if (canvas) this.attach(canvas, doc);
  }
  // lib/camera.js:77-131: drag with the left button orbits about the y axis (the anchor stays where the button went down),
  // the wheel zooms, the arrow keys shift eye and center together
  attach(canvas, doc) {
    let anchor = [];
    const onMove = (ev) => { this.move(anchor, [ev.clientX, ev.clientY]); this.MOVING = 1; };
    canvas.addEventListener('mousedown', (ev) => { if (ev.button == 0) { anchor = [ev.clientX, ev.clientY]; canvas.addEventListener('mousemove', onMove); } });
    canvas.addEventListener('mouseup', () => { canvas.removeEventListener('mousemove', onMove); this.MOVING = 0; });
    canvas.addEventListener('wheel', (ev) => { this.zoom(ev.deltaY || ev.detail || ev.wheelDelta); this.keyPress = 1; });
    if (doc) doc.addEventListener('keydown', (ev) => {
      const fn = { ArrowLeft: 'moveLeft', ArrowRight: 'moveRight', ArrowUp: 'moveUp', ArrowDown: 'moveDown' }[ev.key];
      if (fn) { this[fn](); this.keyPress = 1; }
    });
  }
  set_camera(eye = this.eye, center = this.center, up = this.up) {   // lib/camera.js:25-33
    vec3.set(this.eye, eye[0], eye[1], eye[2]);
    vec3.set(this.center, center[0], center[1], center[2]);
    vec3.set(this.up, up[0], up[1], up[2]);
    vec3.subtract(this.direction, this.eye, this.center);
    mat4.targetTo(this.viewMatrix, eye, center, up);
  }
  zoom(delta) {
    for (let k = 0; k < 3; k++) this.eye[k] += this.direction[k] * this.zoomSpeed * Math.sign(delta);
    this.set_camera();
  }
  move(oldCoord, newCoord) {
    const dX = (newCoord[0] - oldCoord[0]) * Math.PI / 180 * this.moveSpeed;
    this.rotateAngle = dX;
    vec3.rotateY(this.eye, this.eye, [0, 0, 0], dX);
    this.set_camera();
  }
  moveLeft() { this._shift(0, +this.keypressMoveSpeed); }
  moveRight() { this._shift(0, -this.keypressMoveSpeed); }
  moveUp() { this._shift(1, -this.keypressMoveSpeed); }
  moveDown() { this._shift(1, +this.keypressMoveSpeed); }
  _shift(axis, d) {
    const e = [this.eye[0], this.eye[1], this.eye[2]], c = [this.center[0], this.center[1], this.center[2]];
    e[axis] += d; c[axis] += d;
    vec3.set(this.eye, e[0], e[1], e[2]); vec3.set(this.center, c[0], c[1], c[2]);
    this.set_camera();
  }
}
