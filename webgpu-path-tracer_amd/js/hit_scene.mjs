// hit_scene.mjs — hitScene (shaders/hitRay.wgsl:1-113) with hit_sphere (common.wgsl:29-73), hit_quad (:148-187), hit_triangle (:191-242) and
// hit_aabb (:245-256) in plain single-threaded JavaScript over the reference's own buffer layouts (SURVEY.md §8a-0).
//
// What it is for: north_star words its CPU baseline as "the reference's single-threaded JS BVH traversal".  The reference has no such thing —
// its only traversal is the WGSL one — so this is this build's restatement of that WGSL in the reference's host language, timed by
// js/traverse_time.mjs and reported in bench.py's `cpu_baseline.js_traversal`.  Test infrastructure and baseline only: nothing in the product
// calls it.  Second use: an independent check of the C++ oracle in another language — tests/test_js_host.py compares its hit records with the
// oracle's bit for bit (same rays, goldens of configs[1]).
//
// f32 arithmetic: every + - * / and sqrt is evaluated in f64 on f32 operands and rounded once with Math.fround — for these five operations that IS
// the correctly rounded f32 result (53 >= 2 x 24 + 2 bits), i.e. what WGSL's f32 and the oracle compute.  Fog volumes (hit_volume, common.wgsl:102-146)
// need log() and the path's RNG stream and are not restated: a scene that holds one is refused.
const f = Math.fround;
const MAX_FLOAT = f(999999999.999);  // header.wgsl: 1.0e9f after rounding
const ISOTROPIC = 3;                 // header.wgsl:5-8 material types: a sphere whose material is >= ISOTROPIC is a volume

// WGSL min/max: a NaN operand yields the other one; -0 < +0 (include/ptmi_math.h)
function fmin(a, b) {
  if (a !== a) return b;
  if (b !== b) return a;
  if (a === b) return Object.is(a, -0) ? a : b;
  return a < b ? a : b;
}
function fmax(a, b) {
  if (a !== a) return b;
  if (b !== b) return a;
  if (a === b) return Object.is(a, -0) ? b : a;
  return a < b ? b : a;
}
// dot = (x x' + y y') + z z' (SURVEY.md §8a-W "summation order")
const dot3 = (ax, ay, az, bx, by, bz) => f(f(f(ax * bx) + f(ay * by)) + f(az * bz));

export class HitScene {
  // buffers: { spheres, quads, triangles, transforms, materials, bvh: Float32Array; meshes: Int32Array }
  constructor(b, { stackSize = 20, tmin = 1e-6 } = {}) {
    this.b = b;
    this.nSpheres = b.spheres.length / 8;
    this.nQuads = b.quads.length / 20;
    this.nNodes = b.bvh.length / 12;
    this.stackSize = stackSize;
    this.tmin = f(tmin);
    this.stack = new Int32Array(64);
    for (let i = 0; i < this.nSpheres; i++)
      if (b.materials[16 * Math.trunc(b.spheres[8 * i + 6]) + 14] >= ISOTROPIC) throw new Error('hit_scene.mjs: fog volumes are not restated');
    // the record of the closest hit (header.wgsl:119-125), plus counters
    this.t = 0; this.p = new Float32Array(3); this.n = new Float32Array(3); this.front = false; this.material = -1;
    this.nodeVisits = 0; this.triTests = 0;
  }

  setNormal(dx, dy, dz, nx, ny, nz) {  // normalize, front_face, flip (common.wgsl:58-66 and alike)
    const len = f(Math.sqrt(dot3(nx, ny, nz, nx, ny, nz)));
    nx = f(nx / len); ny = f(ny / len); nz = f(nz / len);
    this.front = dot3(dx, dy, dz, nx, ny, nz) < 0;
    if (!this.front) { nx = -nx; ny = -ny; nz = -nz; }
    this.n[0] = nx; this.n[1] = ny; this.n[2] = nz;
  }

  // common.wgsl:29-73
  hitSphere(i, tmax, ox, oy, oz, dx, dy, dz) {
    const s = this.b.spheres, k = 8 * i, tmin = this.tmin;
    const cx = s[k], cy = s[k + 1], cz = s[k + 2], r = s[k + 3];
    const ocx = f(ox - cx), ocy = f(oy - cy), ocz = f(oz - cz);
    const a = dot3(dx, dy, dz, dx, dy, dz);
    const halfB = dot3(dx, dy, dz, ocx, ocy, ocz);
    const c = f(dot3(ocx, ocy, ocz, ocx, ocy, ocz) - f(r * r));
    const disc = f(f(halfB * halfB) - f(a * c));
    if (disc < 0) return false;
    const sq = f(Math.sqrt(disc));
    let root = f(f(-halfB - sq) / a);
    if (root <= tmin || root >= tmax) {
      root = f(f(-halfB + sq) / a);
      if (root <= tmin || root >= tmax) return false;
    }
    this.t = root;
    const px = f(ox + f(root * dx)), py = f(oy + f(root * dy)), pz = f(oz + f(root * dz));
    this.p[0] = px; this.p[1] = py; this.p[2] = pz;
    this.setNormal(dx, dy, dz, f(f(px - cx) / r), f(f(py - cy) / r), f(f(pz - cz) / r));
    this.material = Math.trunc(s[k + 6]);
    return true;
  }

  // common.wgsl:148-187
  hitQuad(i, tmax, ox, oy, oz, dx, dy, dz) {
    const q = this.b.quads, k = 20 * i;
    const nx = q[k + 12], ny = q[k + 13], nz = q[k + 14];
    if (dot3(dx, dy, dz, nx, ny, nz) > 0) return false;
    const denom = dot3(nx, ny, nz, dx, dy, dz);
    if (Math.abs(denom) < f(1e-8)) return false;
    const t = f(f(q[k + 15] - dot3(nx, ny, nz, ox, oy, oz)) / denom);
    if (t <= this.tmin || t >= tmax) return false;
    const ix = f(ox + f(t * dx)), iy = f(oy + f(t * dy)), iz = f(oz + f(t * dz));
    const hx = f(ix - q[k]), hy = f(iy - q[k + 1]), hz = f(iz - q[k + 2]);
    const ux = q[k + 4], uy = q[k + 5], uz = q[k + 6], vx = q[k + 8], vy = q[k + 9], vz = q[k + 10];
    const wx = q[k + 16], wy = q[k + 17], wz = q[k + 18];
    // cross(a, b) = (ay bz - az by, az bx - ax bz, ax by - ay bx)
    const alpha = dot3(wx, wy, wz, f(f(hy * vz) - f(hz * vy)), f(f(hz * vx) - f(hx * vz)), f(f(hx * vy) - f(hy * vx)));
    const beta = dot3(wx, wy, wz, f(f(uy * hz) - f(uz * hy)), f(f(uz * hx) - f(ux * hz)), f(f(ux * hy) - f(uy * hx)));
    if (alpha < 0 || 1 < alpha || beta < 0 || 1 < beta) return false;
    this.t = t;
    this.p[0] = ix; this.p[1] = iy; this.p[2] = iz;
    this.setNormal(dx, dy, dz, nx, ny, nz);
    this.material = Math.trunc(q[k + 19]);
    return true;
  }

  // common.wgsl:191-242 — column-major mat4 * vec4: ((c0 x + c1 y) + c2 z) + c3 w
  hitTriangle(i, tmax, ox, oy, oz, dx, dy, dz) {
    this.triTests++;
    const tr = this.b.triangles, k = 24 * i, tmin = this.tmin;
    const mesh = 4 * Math.trunc(tr[k + 23]);
    const m = this.b.transforms, mo = 32 * this.b.meshes[mesh + 2] + 16;  // invModelMatrix
    const mul = (r, x, y, z, w) => f(f(f(f(m[mo + r] * x) + f(m[mo + 4 + r] * y)) + f(m[mo + 8 + r] * z)) + f(m[mo + 12 + r] * w));
    const rox = mul(0, ox, oy, oz, 1), roy = mul(1, ox, oy, oz, 1), roz = mul(2, ox, oy, oz, 1);
    const rdx = mul(0, dx, dy, dz, 0), rdy = mul(1, dx, dy, dz, 0), rdz = mul(2, dx, dy, dz, 0);
    const ax = tr[k], ay = tr[k + 1], az = tr[k + 2];
    const abx = f(tr[k + 4] - ax), aby = f(tr[k + 5] - ay), abz = f(tr[k + 6] - az);
    const acx = f(tr[k + 8] - ax), acy = f(tr[k + 9] - ay), acz = f(tr[k + 10] - az);
    const nx = f(f(aby * acz) - f(abz * acy)), ny = f(f(abz * acx) - f(abx * acz)), nz = f(f(abx * acy) - f(aby * acx));
    const det = -dot3(rdx, rdy, rdz, nx, ny, nz);
    if (Math.abs(det) < tmin) return false;
    const aox = f(rox - ax), aoy = f(roy - ay), aoz = f(roz - az);
    const dax = f(f(aoy * rdz) - f(aoz * rdy)), day = f(f(aoz * rdx) - f(aox * rdz)), daz = f(f(aox * rdy) - f(aoy * rdx));
    const invDet = f(1 / det);
    const dst = f(dot3(aox, aoy, aoz, nx, ny, nz) * invDet);
    const uu = f(dot3(acx, acy, acz, dax, day, daz) * invDet);
    const vv = f(-dot3(abx, aby, abz, dax, day, daz) * invDet);
    const ww = f(f(1 - uu) - vv);
    if (dst < tmin || dst > tmax || uu < tmin || vv < tmin || ww < tmin) return false;
    this.t = dst;
    this.p[0] = f(ox + f(dst * dx)); this.p[1] = f(oy + f(dst * dy)); this.p[2] = f(oz + f(dst * dz));
    // normal = nA ww + nB uu + nC vv, then transpose(invModelMatrix) * vec4(normal, 0)
    const c = (o) => f(f(f(tr[k + 12 + o] * ww) + f(tr[k + 16 + o] * uu)) + f(tr[k + 20 + o] * vv));
    const hx = c(0), hy = c(1), hz = c(2);
    // column r of the transpose is row r of m: ((row0 x + row1 y) + row2 z) + row3 w, componentwise
    const tmul = (r) => f(f(f(f(m[mo + 4 * r] * hx) + f(m[mo + 4 * r + 1] * hy)) + f(m[mo + 4 * r + 2] * hz)) + f(m[mo + 4 * r + 3] * 0));
    this.setNormal(dx, dy, dz, tmul(0), tmul(1), tmul(2));
    this.material = this.b.meshes[mesh + 3];
    return true;
  }

  // common.wgsl:245-256
  hitAabb(node, tmax, ox, oy, oz, ix, iy, iz) {
    this.nodeVisits++;
    const v = this.b.bvh, k = 12 * node;
    const t0x = f(f(v[k] - ox) * ix), t0y = f(f(v[k + 1] - oy) * iy), t0z = f(f(v[k + 2] - oz) * iz);
    const t1x = f(f(v[k + 4] - ox) * ix), t1y = f(f(v[k + 5] - oy) * iy), t1z = f(f(v[k + 6] - oz) * iz);
    const tMin = fmax(this.tmin, fmax(fmin(t0x, t1x), fmax(fmin(t0y, t1y), fmin(t0z, t1z))));
    const tMax = fmin(tmax, fmin(fmax(t0x, t1x), fmin(fmax(t0y, t1y), fmax(t0z, t1z))));
    return tMax > tMin;
  }

  // hitRay.wgsl:1-113 — returns hit_anything; the record is in this.t / p / n / front / material
  hit(ox, oy, oz, dx, dy, dz) {
    let closest = MAX_FLOAT, any = false;
    for (let i = 0; i < this.nSpheres; i++) if (this.hitSphere(i, closest, ox, oy, oz, dx, dy, dz)) { any = true; closest = this.t; }
    for (let i = 0; i < this.nQuads; i++) if (this.hitQuad(i, closest, ox, oy, oz, dx, dy, dz)) { any = true; closest = this.t; }
    if (this.nNodes <= 0) return any;
    const v = this.b.bvh, stack = this.stack;
    const ix = f(1 / dx), iy = f(1 / dy), iz = f(1 / dz);
    let top = 0, cur = 0;
    for (;;) {
      const k = 12 * cur;
      if (this.hitAabb(cur, closest, ox, oy, oz, ix, iy, iz)) {
        if (Math.trunc(v[k + 7]) === 2) {  // leaf
          const start = Math.trunc(v[k + 8]), count = Math.trunc(v[k + 9]);
          for (let j = 0; j < count; j++) if (this.hitTriangle(start + j, closest, ox, oy, oz, dx, dy, dz)) { any = true; closest = this.t; }
          if (top === 0) break;
          cur = stack[--top];
        } else {
          const axis = Math.trunc(v[k + 11]);
          if ((axis === 0 ? dx : axis === 1 ? dy : dz) < 0) { stack[top++] = cur + 1; cur = Math.trunc(v[k + 3]); }
          else { stack[top++] = Math.trunc(v[k + 3]); cur++; }
        }
      } else {
        if (top === 0) break;
        cur = stack[--top];
      }
      if (top >= this.stackSize) break;  // hitRay.wgsl:106-109 (Q7)
    }
    return any;
  }
}
