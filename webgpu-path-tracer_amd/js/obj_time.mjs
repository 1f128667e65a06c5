// Times the shipped restatement of the reference's OBJ reader (lib/primitives/objReader.js:10-68 = js/lib/scene.mjs ObjReader.parse:
// line split, trim, split(' ') / split(/[\s/]+/), Number(), index arrays, .flat(1), Float32Array) on one file, single-threaded, and prints
// {ms, vertices, normals, sum} — sum = position-weighted sum of the f32 bit patterns of both arrays mod 2^32, to compare with the native parser's output.
//   node --max-old-space-size=16384 obj_time.mjs file.obj
import { readFileSync } from 'fs';
import { performance } from 'perf_hooks';
import { ObjReader } from './lib/scene.mjs';
const text = readFileSync(process.argv[2], 'utf8');
const t0 = performance.now();
const r = ObjReader.parse(text);
const ms = performance.now() - t0;
let h = 0;
for (const a of [r.vertices, r.normals]) {
  const u = new Uint32Array(a.buffer, a.byteOffset, a.length);
  for (let i = 0; i < u.length; i++) h = (h + Math.imul(u[i], (i & 0xffff) + 1)) >>> 0;
}
console.log(JSON.stringify({ ms, vertices: r.vertices.length, normals: r.normals.length, sum: h, node: process.version }));
