// Generates tests/golden/camera_sequence.json by driving the reference's OWN lib/camera.js (imported unchanged from
// /root/reference) through its event handlers under Node:
//     cd oracle/capture && node --experimental-loader ./loader.mjs capture_camera.mjs
// A stub canvas / document records the listeners the Camera constructor installs (lib/camera.js:77-131); the script then
// fires wheel, mousedown/mousemove/mouseup and keydown events and dumps eye / center / direction / viewMatrix (f32 bit
// patterns) and the MOVING / keyPress flags after each one.  Output is DATA only.
// (gl-matrix itself is not in the container: the loader maps the CDN import to this build's gl-matrix-compatible module,
// so what the file pins is camera.js's own logic — SURVEY.md §8c "parity at the gl-matrix boundary is unpinned".)
import fs from 'fs';
import path from 'path';

const REF = '/root/reference/';
const OUT = path.resolve(path.dirname(new URL(import.meta.url).pathname), '../../tests/golden/camera_sequence.json');

function makeTarget() {
  const ls = {};
  return {
    ls,
    addEventListener(t, f) { (ls[t] = ls[t] || []).push(f); },
    removeEventListener(t, f) { ls[t] = (ls[t] || []).filter((g) => g !== f); },
    fire(t, ev) { for (const f of (ls[t] || []).slice()) f(ev); },
  };
}

async function main() {
  const canvas = makeTarget();
  const doc = makeTarget();
  globalThis.document = doc;
  const { Camera } = await import('file://' + REF + 'lib/camera.js');
  const cam = new Camera(canvas);
  const steps = [];
  const bits = (f32) => Array.from(new Uint32Array(new Float32Array(f32).buffer));   // JSON cannot carry the sign of -0: keep the bit patterns
  const snap = (op) => steps.push({ op, eye: bits(cam.eye), center: bits(cam.center), direction: bits(cam.direction), viewMatrix: bits(cam.viewMatrix),
    viewMatrix_f32: Array.from(cam.viewMatrix), MOVING: cam.MOVING, keyPress: cam.keyPress, rotateAngle: cam.rotateAngle });
  // renderer/index.js start-up (index.js:25): camera.set_camera(eye, center, up)
  cam.set_camera([0.5, 0, 2.5], [0.5, 0, 0], [0, 1, 0]);
  snap({ kind: 'set_camera', eye: [0.5, 0, 2.5], center: [0.5, 0, 0], up: [0, 1, 0] });
  const wheel = (deltaY) => { canvas.fire('wheel', { deltaY }); snap({ kind: 'wheel', deltaY }); };
  const key = (k) => { doc.fire('keydown', { key: k }); snap({ kind: 'keydown', key: k }); };
  const down = (x, y, button = 0) => { canvas.fire('mousedown', { button, clientX: x, clientY: y }); snap({ kind: 'mousedown', button, x, y }); };
  const move = (x, y) => { canvas.fire('mousemove', { clientX: x, clientY: y }); snap({ kind: 'mousemove', x, y }); };
  const up = () => { canvas.fire('mouseup', {}); snap({ kind: 'mouseup' }); };
  wheel(120); wheel(120); wheel(-53); wheel(3.5);
  key('ArrowLeft'); key('ArrowLeft'); key('ArrowUp'); key('ArrowRight'); key('ArrowDown'); key('ArrowDown'); key('a');
  move(10, 10);              // no button held: no listener yet, nothing moves
  down(400, 300); move(460, 310); move(523, 290); move(380, 300); up();
  move(100, 100);            // released: ignored again
  down(10, 10, 2);           // right button: ignored (event.button == 0 only)
  move(300, 10);
  wheel(-120); key('ArrowRight');
  down(100, 200); move(1000, 200); up();
  cam.set_camera([1.2, 0.4, 2.1], [0.1, -0.2, 0], [0, 1, 0]);
  snap({ kind: 'set_camera', eye: [1.2, 0.4, 2.1], center: [0.1, -0.2, 0], up: [0, 1, 0] });
  wheel(1); key('ArrowUp'); down(0, 0); move(-250, 40); up();
  fs.writeFileSync(OUT, JSON.stringify({ note: 'reference lib/camera.js driven through its own event listeners; eye/center/direction/viewMatrix are f32 bit patterns (u32), viewMatrix_f32 the same values for reading', steps }, null, 0));
  console.log('wrote', OUT, steps.length, 'steps');
}
main().catch((e) => { console.error(e); process.exit(1); });
