// node check_camera.mjs <tests/golden/camera_sequence.json> — replays the event sequence captured from the reference's
// lib/camera.js on this build's Camera (lib/scene.mjs) through the same kind of stub canvas / document and compares
// eye / center / direction / viewMatrix / flags after every event, bit for bit.  Prints {ok, steps, firstBad}.
import fs from 'fs';
import { Camera } from './lib/scene.mjs';

function makeTarget() {
  const ls = {};
  return { addEventListener(t, f) { (ls[t] = ls[t] || []).push(f); }, removeEventListener(t, f) { ls[t] = (ls[t] || []).filter((g) => g !== f); },
    fire(t, ev) { for (const f of (ls[t] || []).slice()) f(ev); } };
}
const gold = JSON.parse(fs.readFileSync(process.argv[2], 'utf8')).steps;
const canvas = makeTarget(), doc = makeTarget();
const cam = new Camera(canvas, doc);
const bits = (f32) => Array.from(new Uint32Array(new Float32Array(f32).buffer));
const same = (f32, want) => { const a = bits(f32); return a.length === want.length && a.every((v, i) => v === want[i]); };
let firstBad = null;
gold.forEach((g, i) => {
  const o = g.op;
  if (o.kind === 'set_camera') cam.set_camera(o.eye, o.center, o.up);
  else if (o.kind === 'wheel') canvas.fire('wheel', { deltaY: o.deltaY });
  else if (o.kind === 'keydown') doc.fire('keydown', { key: o.key });
  else if (o.kind === 'mousedown') canvas.fire('mousedown', { button: o.button, clientX: o.x, clientY: o.y });
  else if (o.kind === 'mousemove') canvas.fire('mousemove', { clientX: o.x, clientY: o.y });
  else if (o.kind === 'mouseup') canvas.fire('mouseup', {});
  const ok = same(cam.eye, g.eye) && same(cam.center, g.center) && same(cam.direction, g.direction) &&
    same(cam.viewMatrix, g.viewMatrix) && cam.MOVING === g.MOVING && cam.keyPress === g.keyPress && cam.rotateAngle === g.rotateAngle;
  if (!ok && firstBad === null) firstBad = { step: i, op: o, got: { eye: Array.from(cam.eye), view: Array.from(cam.viewMatrix), MOVING: cam.MOVING, keyPress: cam.keyPress }, want: g };
});
console.log(JSON.stringify({ ok: firstBad === null, steps: gold.length, firstBad }));
