// node bvh_time.mjs <boxes.f64> — times the median-split build of lib/scene.mjs (the reference's algorithm, lib/BVH/bvhNode.js:21-101,
// single-threaded JavaScript as in the reference) on n boxes stored as [bmin (3n f64) | bmax (3n f64)].  Prints JSON.
import fs from 'fs';
import { performance } from 'perf_hooks';
import { build_bvh } from './lib/scene.mjs';
const raw = fs.readFileSync(process.argv[2]);
const all = new Float64Array(raw.buffer, raw.byteOffset, raw.length / 8);
const n = all.length / 6;
const bmin = all.subarray(0, 3 * n), bmax = all.subarray(3 * n);
const t = performance.now();
const r = build_bvh(bmin, bmax, 2, null);
const ms = performance.now() - t;
console.log(JSON.stringify({ n, ms, nodes: r.nodes.length / 12, node: process.version }));
