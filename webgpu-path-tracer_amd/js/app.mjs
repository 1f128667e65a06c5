// app.mjs — headless Node application on the shipped host code (no reference checkout needed):
//   node app.mjs --golden <dir>/c2 --width 160 --height 90 --frames 4 --bounces 8 --camera cornell --raw out.f32 [--out out.ppm]
//   node app.mjs --obj model.obj --scale 0.6 --translate 0,-0.4,0 ...       (Cornell box + one OBJ mesh)
//   ... --bvh device | sah : build the tree on the GPU over the uploaded triangles instead of using the uploaded one — Scene.create_bvh()'s median split
//   (ptmi_build_scene_bvh), or the reference's other, never-called builder (lib/BVH/bvhNode.js:108-283; ptmi_build_scene_bvh_sah; give --stack above its depth)
// Loads a scene (golden typed arrays or built with lib/scene.mjs), drives Renderer -> WebGPU shim -> ptmi.node.
import fs from 'fs';
import { Ptmi, loadNative } from './ptmi.mjs';
import { GPUDeviceNode } from './webgpu_node.mjs';
import { Renderer } from './renderer.mjs';
import { Camera, ObjReader } from './lib/scene.mjs';
import { CAMERAS, c1Scene, c2Scene, sceneBuffers } from './lib/scenes.mjs';

const args = process.argv.slice(2);
const opt = (k, d) => { const i = args.indexOf('--' + k); return i >= 0 ? args[i + 1] : d; };
const W = Number(opt('width', 320)), H = Number(opt('height', 180)), frames = Number(opt('frames', 4));

function loadGolden(prefix) {
  const rd = (name, T) => { const f = `${prefix}_${name}.bin`; if (!fs.existsSync(f)) return new T(0); const b = fs.readFileSync(f); return new T(b.buffer, b.byteOffset, b.length / 4); };
  return { spheres: rd('spheres', Float32Array), quads: rd('quads', Float32Array), triangles: rd('triangles', Float32Array), meshes: rd('meshes', Int32Array),
    transforms: rd('transforms', Float32Array), materials: rd('materials', Float32Array), bvh: rd('bvh', Float32Array) };
}

async function main() {
  let buffers;
  if (opt('golden', null)) buffers = loadGolden(opt('golden'));
  else if (opt('obj', null)) {
    const sc = c2Scene(ObjReader.parse(fs.readFileSync(opt('obj'), 'utf8')));
    sc.native = loadNative();
    buffers = await sceneBuffers(sc);
  } else buffers = await sceneBuffers(c1Scene());
  // --device N = one GPU; --devices a,b,.. = one context over several GPUs (ptmi_create_multi: tiles sharded, one RCCL reduce on read-back)
  const backend = new Ptmi(opt('devices', null) ? opt('devices').split(',').map(Number) : Number(opt('device', 0)));
  const canvas = { wantPixels: false, pixels: null };
  const device = new GPUDeviceNode(backend, canvas);
  const camera = new Camera();
  const [eye, center] = CAMERAS[opt('camera', 'cornell')];
  camera.set_camera(eye, center, [0, 1, 0]);
  const renderer = await Renderer.create(device);
  const module = renderer.createShaderModule({ MAX_BOUNCES: Number(opt('bounces', 8)), IMPORTANCE_SAMPLING: args.includes('--is'), STACK_SIZE: Number(opt('stack', 20)) });
  await renderer.initBuffers(null, camera, W, H, buffers);
  if (opt('bvh', null) === 'sah') backend.buildSceneBVHSAH();
  else if (opt('bvh', null) === 'device') backend.buildSceneBVH();
  renderer.createComputePipeline(module);
  renderer.createRenderPipeline(module);
  renderer.setRenderParameters({}, camera, W, H);
  renderer.renderAnimation(frames);
  backend.synchronize();
  if (opt('raw', null)) fs.writeFileSync(opt('raw'), Buffer.from(backend.readFramebuffer().buffer));
  if (opt('out', null)) {
    const px = backend.resolveRGBA8(frames);
    const rgb = Buffer.alloc(W * H * 3);
    for (let i = 0; i < W * H; i++) { rgb[3 * i] = px[4 * i]; rgb[3 * i + 1] = px[4 * i + 1]; rgb[3 * i + 2] = px[4 * i + 2]; }
    fs.writeFileSync(opt('out'), Buffer.concat([Buffer.from(`P6\n${W} ${H}\n255\n`), rgb]));
  }
  console.log(JSON.stringify({ frames, stats: backend.stats() }));
  backend.destroy();
}
main().catch((e) => { console.error(e.message || e); process.exit(1); });
