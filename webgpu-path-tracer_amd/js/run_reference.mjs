// Runs the reference application's OWN index.js (and through it renderer.js, webgpu-utils.js, lib/*.js)
// unchanged under Node, with the HIP integrator behind the WebGPU calls:
//
//   PTMI_REFERENCE_ROOT=/path/to/WebGPU-Path-Tracer \
//   node --experimental-loader ./ref_loader.mjs run_reference.mjs --frames 16 --out frame.ppm
//
//   --width/--height  canvas size (index.html: 900x600)     --frames N   animation frames to run
//   --device N        GPU to use                            --devices a,b,..  ONE context over several GPUs (tiles sharded, RCCL reduce)
//   --mock            record the backend calls instead of using a GPU (writes --dump <file.json>)
//   --out file.ppm    tone-mapped image of the last frame   --raw file.f32  raw RGBA f32 framebuffer sum
import fs from 'fs';
import path from 'path';
import { pathToFileURL } from 'url';
import { performance } from 'perf_hooks';
import { installBrowserShims } from './webgpu_node.mjs';

const args = process.argv.slice(2);
const opt = (k, d) => { const i = args.indexOf('--' + k); return i >= 0 ? args[i + 1] : d; };
const flag = (k) => args.includes('--' + k);
const root = path.resolve(process.env.PTMI_REFERENCE_ROOT || '/root/reference');
const width = Number(opt('width', 900)), height = Number(opt('height', 600)), frames = Number(opt('frames', 4));
globalThis.performance = performance;

async function main() {
  let backend;
  if (flag('mock')) backend = new (await import('./mock_backend.mjs')).MockBackend();
  else backend = new (await import('./ptmi.mjs')).Ptmi(opt('devices', null) ? opt('devices').split(',').map(Number) : Number(opt('device', 0)));
  const quiet = console.log;
  if (!flag('verbose')) console.log = () => {};
  const done = new Promise((resolve) => {
    installBrowserShims({ backend, width, height, root, maxFrames: frames, onDone: (device, canvas) => resolve({ device, canvas }) });
  });
  process.chdir(root);   // the reference fetches './assets/...' and './shaders/...' relative to the page
  await import(pathToFileURL(path.join(root, 'index.js')).href);
  const { device } = await done;
  console.log = quiet;
  if (flag('mock')) {
    const dump = opt('dump', null);
    const out = { calls: backend.calls, frames: backend.frames, params: backend.params, uploads: {} };
    for (const [k, v] of Object.entries(backend.uploads)) out.uploads[k] = { length: v.length, bytes: Buffer.from(v.buffer, v.byteOffset, v.byteLength).toString('base64') };
    if (dump) fs.writeFileSync(dump, JSON.stringify(out)); else console.log(JSON.stringify(out.calls));
    return;
  }
  backend.synchronize();
  const raw = opt('raw', null), ppm = opt('out', null);
  if (raw) { const fb = backend.readFramebuffer(); fs.writeFileSync(raw, Buffer.from(fb.buffer)); }
  if (ppm) {
    const px = backend.resolveRGBA8(device.frames);
    const rgb = Buffer.alloc(width * height * 3);
    for (let i = 0; i < width * height; i++) { rgb[3 * i] = px[4 * i]; rgb[3 * i + 1] = px[4 * i + 1]; rgb[3 * i + 2] = px[4 * i + 2]; }
    fs.writeFileSync(ppm, Buffer.concat([Buffer.from(`P6\n${width} ${height}\n255\n`), rgb]));
  }
  console.log(JSON.stringify({ frames: device.frames, stats: backend.stats() }));
  backend.destroy();
}
main().catch((e) => { console.error(e); process.exit(1); });
