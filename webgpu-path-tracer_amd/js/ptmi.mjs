// ptmi.mjs — loads the N-API addon (ptmi.node -> libptmi.so) and wraps it in a small class.
// There is no JavaScript fallback for rendering: if the addon or the GPU is missing, this throws.
import { createRequire } from 'module';

const require_ = createRequire(import.meta.url);
let native_ = null;

export function loadNative() {
  if (!native_) native_ = require_('./ptmi.node');
  return native_;
}

export const BUFFER_NAMES = ['spheres', 'quads', 'triangles', 'meshes', 'transforms', 'materials', 'bvh'];

// One integrator context (include/ptmi.h).  `device` is a GPU index, or an array of them: one context over several GPUs of
// the node (ptmi_create_multi) — pixel tiles sharded across them, one RCCL reduce inside readFramebuffer().
export class Ptmi {
  constructor(device = 0, native = loadNative()) {
    this.native = native;
    this.h = native.create(device);
    this.width = 0;
    this.height = 0;
  }
  destroy() { if (this.h) { this.native.destroy(this.h); this.h = null; } }
  setParams(p) { return this.native.setParams(this.h, p); }
  upload(name, typedArray) { this.native.upload(this.h, this.native.BUF[name], typedArray); }
  uploadScene(buffers) { for (const k of BUFFER_NAMES) this.upload(k, buffers[k]); }
  resize(w, h) { this.native.resize(this.h, w, h); this.width = w; this.height = h; }
  clear() { this.native.clear(this.h); }
  setShard(rank, world, tile) { this.native.setShard(this.h, rank, world, tile); }
  renderFrame(uniforms20) { this.native.renderFrame(this.h, uniforms20); }
  render(view16, firstFrame, nFrames) { this.native.render(this.h, view16, firstFrame, nFrames); }
  synchronize() { this.native.synchronize(this.h); }
  prepare() { this.native.prepare(this.h); }
  buildSceneBVHSAH() { this.native.buildSceneBVHSAH(this.h); }   // the same with the reference's never-called SAH builder (lib/BVH/bvhNode.js:108-283): opt-in
  buildSceneBVH() { this.native.buildSceneBVH(this.h); }   // Scene.create_bvh() (lib/scene.js:253-259) on the GPU, over the uploaded unordered triangles
  readFramebuffer(out = new Float32Array(this.width * this.height * 4)) { return this.native.readFramebuffer(this.h, out); }
  writeFramebuffer(src) { this.native.writeFramebuffer(this.h, src); }
  resolveRGBA8(frameNum, out = new Uint8Array(this.width * this.height * 4)) { return this.native.resolveRGBA8(this.h, frameNum, out); }
  setCounters(on) { this.native.setCounters(this.h, on); }
  setTiming(on) { this.native.setTiming(this.h, on); }
  stats() { return this.native.stats(this.h); }
  resetStats() { this.native.resetStats(this.h); }
}
