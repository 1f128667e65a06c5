// node traverse_time.mjs <dir> <tag> <rays.f32> [out.bin] [stack_size] — hitScene (js/hit_scene.mjs: the WGSL traversal restated in single-threaded JavaScript) over
// the rays of <rays.f32> (6 f32 each: origin, direction) against the scene stored as <dir>/<tag>_{spheres,quads,triangles,meshes,transforms,materials,bvh}.bin
// (the reference's layouts; a missing file is an empty array).  Prints JSON {rays, ms, mrays_s, node_visits, tri_tests, hits}; with out.bin it also writes one
// record of 9 f32 per ray — hit (0/1), t, p.xyz, normal.xyz, front_face — and an i32 material index per ray behind them, for the parity test.
import fs from 'fs';
import path from 'path';
import { performance } from 'perf_hooks';
import { HitScene } from './hit_scene.mjs';

const [dir, tag, raysFile, outFile, stackArg] = process.argv.slice(2);
const rd = (name, T) => {
  const p = path.join(dir, `${tag}_${name}.bin`);
  if (!fs.existsSync(p)) return new T(0);
  const b = fs.readFileSync(p);
  return new T(b.buffer.slice(b.byteOffset, b.byteOffset + b.length));
};
const buffers = { meshes: rd('meshes', Int32Array) };
for (const n of ['spheres', 'quads', 'triangles', 'transforms', 'materials', 'bvh']) buffers[n] = rd(n, Float32Array);
const rb = fs.readFileSync(raysFile);
const rays = new Float32Array(rb.buffer.slice(rb.byteOffset, rb.byteOffset + rb.length));
const n = rays.length / 6;
const hs = new HitScene(buffers, { stackSize: stackArg ? parseInt(stackArg, 10) : 20 });
const rec = outFile ? new Float32Array(9 * n) : null, mat = outFile ? new Int32Array(n) : null;
let hits = 0;
const t0 = performance.now();
for (let i = 0; i < n; i++) {
  const k = 6 * i;
  const h = hs.hit(rays[k], rays[k + 1], rays[k + 2], rays[k + 3], rays[k + 4], rays[k + 5]);
  if (h) hits++;
  if (rec) {
    const o = 9 * i;
    rec[o] = h ? 1 : 0;
    if (h) {
      rec[o + 1] = hs.t;
      rec[o + 2] = hs.p[0]; rec[o + 3] = hs.p[1]; rec[o + 4] = hs.p[2];
      rec[o + 5] = hs.n[0]; rec[o + 6] = hs.n[1]; rec[o + 7] = hs.n[2];
      rec[o + 8] = hs.front ? 1 : 0;
      mat[i] = hs.material;
    } else mat[i] = -1;
  }
}
const ms = performance.now() - t0;
if (outFile) fs.writeFileSync(outFile, Buffer.concat([Buffer.from(rec.buffer), Buffer.from(mat.buffer)]));
console.log(JSON.stringify({ rays: n, ms, mrays_s: n / ms / 1e3, node_visits: hs.nodeVisits, tri_tests: hs.triTests, hits, node: process.version }));
