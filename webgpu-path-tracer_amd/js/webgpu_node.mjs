// webgpu_node.mjs — a `navigator.gpu`-shaped device for Node whose compute dispatch is the HIP integrator.
//
// The reference drives its hot path through WebGPU only (webgpu-utils.js:15-69,111-148, renderer.js:91-124,
// 184-191): create buffers from typed arrays, write 80 bytes of uniforms, dispatch `computeFrameBuffer`, then a
// fullscreen draw.  This module implements exactly that slice of the WebGPU API on top of a backend with the
// surface of ptmi.mjs's Ptmi class, so the reference's OWN webgpu-utils.js / renderer.js / lib/*.js / index.js
// run unchanged under Node (see run_reference.mjs and INTEGRATION.md):
//
//   createBuffer + queue.writeBuffer   -> staged copy; uploaded (ptmi_upload) when first bound in a dispatch
//   createShaderModule({code})         -> the WGSL is not compiled; its header constants (NUM_SAMPLES, MAX_BOUNCES,
//                                         STRATIFY, IMPORTANCE_SAMPLING, STACK_SIZE, background_color) become ptmi_params
//   dispatchWorkgroups (compute pass)  -> ptmi_render_frame(uniforms)         [bindings of shaders/header.wgsl:15-23]
//   draw (render pass)                 -> display pass: optional RGBA8 resolve into canvas.pixels, and the
//                                         framebuffer clear the fragment shader performs when resetBuffer == 1
//                                         (shaders/fragment.js:30-33)
const BINDING_NAMES = { 1: 'spheres', 2: 'quads', 5: 'triangles', 6: 'meshes', 7: 'transforms', 8: 'materials', 9: 'bvh' };

export const GPUBufferUsage = { MAP_READ: 1, MAP_WRITE: 2, COPY_SRC: 4, COPY_DST: 8, INDEX: 16, VERTEX: 32, UNIFORM: 64, STORAGE: 128, INDIRECT: 256, QUERY_RESOLVE: 512 };

export function paramsFromWGSL(code) {
  const p = {};
  const num = (re) => { const m = re.exec(code); return m ? Number(m[1]) : undefined; };
  const bool = (re) => { const m = re.exec(code); return m ? (m[1] === 'true' ? 1 : 0) : undefined; };
  const set = (k, v) => { if (v !== undefined && !Number.isNaN(v)) p[k] = v; };
  set('num_samples', num(/const\s+NUM_SAMPLES\s*=\s*([0-9]+)/));
  set('max_bounces', num(/const\s+MAX_BOUNCES\s*=\s*([0-9]+)/));
  set('stack_size', num(/const\s+STACK_SIZE\s*=\s*([0-9]+)/));
  set('stratify', bool(/const\s+STRATIFY\s*=\s*(true|false)/));
  set('importance_sampling', bool(/const\s+IMPORTANCE_SAMPLING\s*=\s*(true|false)/));
  const bg = /background_color\s*=\s*vec3f\(\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)\s*\)/.exec(code);
  if (bg) p.background = [Number(bg[1]), Number(bg[2]), Number(bg[3])];
  return p;
}

class GPUBufferNode {
  constructor(desc) { this.label = desc.label; this.size = desc.size; this.usage = desc.usage; this.bytes = new Uint8Array(desc.size); this.version = 0; }
  destroy() {}
}

class ComputePass {
  constructor(enc) { this.enc = enc; this.pipeline = null; this.groups = {}; }
  setPipeline(p) { this.pipeline = p; }
  setBindGroup(i, g) { this.groups[i] = g; }
  dispatchWorkgroups(x, y = 1, z = 1) { this.enc.cmds.push({ kind: 'dispatch', pipeline: this.pipeline, group: this.groups[0], x, y, z }); }
  end() {}
}
class RenderPass {
  constructor(enc, desc) { this.enc = enc; this.desc = desc; this.pipeline = null; this.groups = {}; }
  setPipeline(p) { this.pipeline = p; }
  setBindGroup(i, g) { this.groups[i] = g; }
  setVertexBuffer() {}
  draw(n) { this.enc.cmds.push({ kind: 'draw', pipeline: this.pipeline, group: this.groups[0], n }); }
  end() {}
}
class CommandEncoder {
  constructor(device) { this.device = device; this.cmds = []; }
  beginComputePass() { return new ComputePass(this); }
  beginRenderPass(desc) { return new RenderPass(this, desc); }
  finish() { return { cmds: this.cmds }; }
}

export class GPUDeviceNode {
  constructor(backend, canvas) {
    this.backend = backend;
    this.canvas = canvas;
    this.limits = { maxTextureDimension2D: 16384, maxComputeWorkgroupsPerDimension: 65535 };
    this.lost = new Promise(() => {});
    this.uploaded = new Map();   // binding name -> {buffer, version}
    this.fbBuffer = null;
    this.fbSeeded = -1;
    this.shaderParams = {};
    this.frames = 0;
    const self = this;
    this.queue = {
      writeBuffer(buf, offset, data, dataOffset = 0, size) {
        const src = ArrayBuffer.isView(data) ? new Uint8Array(data.buffer, data.byteOffset, data.byteLength) : new Uint8Array(data);
        const n = size === undefined ? src.length - dataOffset : size;
        buf.bytes.set(src.subarray(dataOffset, dataOffset + n), offset);
        buf.version++;
      },
      submit(cmdBuffers) { for (const cb of cmdBuffers) for (const c of cb.cmds) self._execute(c); },
      onSubmittedWorkDone() { self.backend.synchronize(); return Promise.resolve(); },
    };
  }
  createBuffer(desc) { return new GPUBufferNode(desc); }
  createShaderModule(desc) { Object.assign(this.shaderParams, paramsFromWGSL(desc.code || '')); return { code: desc.code }; }
  createComputePipeline(desc) { return { kind: 'compute', desc, getBindGroupLayout: (i) => ({ index: i }) }; }
  createRenderPipeline(desc) { return { kind: 'render', desc, getBindGroupLayout: (i) => ({ index: i }) }; }
  createBindGroup(desc) { const m = {}; for (const e of desc.entries) m[e.binding] = e.resource.buffer; return { label: desc.label, buffers: m }; }
  createCommandEncoder() { return new CommandEncoder(this); }

  _uniformsOf(group) { const b = group.buffers[0]; return new Float32Array(b.bytes.buffer.slice(0, 80)); }

  _execute(c) {
    if (c.kind === 'dispatch') {
      const g = c.group, u = this._uniformsOf(g);
      const W = u[0], H = u[1];
      if (Object.keys(this.shaderParams).length && !this.paramsApplied) { this.backend.setParams(this.shaderParams); this.paramsApplied = true; }
      for (const [binding, name] of Object.entries(BINDING_NAMES)) {
        const buf = g.buffers[binding];
        if (!buf) throw new Error(`compute bind group lacks binding ${binding} (${name})`);
        const have = this.uploaded.get(name);
        if (!have || have.buffer !== buf || have.version !== buf.version) {
          const ctor = name === 'meshes' ? Int32Array : Float32Array;
          this.backend.upload(name, new ctor(buf.bytes.buffer.slice(0, buf.size)));
          this.uploaded.set(name, { buffer: buf, version: buf.version });
        }
      }
      const fb = g.buffers[3];
      if (!fb) throw new Error('compute bind group lacks binding 3 (framebuffer)');
      if (fb.size !== W * H * 16) throw new Error(`framebuffer binding holds ${fb.size} bytes, uniforms say ${W}x${H}`);
      if (this.fbBuffer !== fb) { this.backend.resize(W, H); this.fbBuffer = fb; this.fbSeeded = -1; }
      if (this.fbSeeded !== fb.version) {   // host wrote the framebuffer (renderer.js:88,100 writes zeros)
        const f = new Float32Array(fb.bytes.buffer);
        if (f.some((x) => x !== 0)) this.backend.writeFramebuffer(f); else if (this.fbSeeded !== -1) this.backend.clear();
        this.fbSeeded = fb.version;
      }
      this.backend.renderFrame(u);
      this.frames++;
    } else if (c.kind === 'draw') {
      const u = this._uniformsOf(c.group);
      if (this.canvas && this.canvas.wantPixels) this.canvas.pixels = this.backend.resolveRGBA8(u[2], this.canvas.pixels);
      if (u[3] === 1) this.backend.clear();   // shaders/fragment.js:30-33
    }
  }
}

// Minimal browser environment for the reference's index.js / renderer.js / lib/camera.js.
export function installBrowserShims({ backend, width, height, root = null, maxFrames = 1, onDone = null }) {
  const listeners = {};
  const canvas = {
    clientWidth: width, clientHeight: height, width, height, wantPixels: false, pixels: null,
    addEventListener(t, f) { (listeners[t] = listeners[t] || []).push(f); }, removeEventListener() {},
    getContext() { return { configure() {}, getCurrentTexture() { return { createView() { return {}; } }; } }; },
  };
  const device = new GPUDeviceNode(backend, canvas);
  const g = globalThis;
  g.navigator = { gpu: { requestAdapter: async () => ({ requestDevice: async () => device }), getPreferredCanvasFormat: () => 'bgra8unorm' } };
  g.GPUBufferUsage = GPUBufferUsage;
  g.document = { querySelector: () => canvas, addEventListener() {}, body: { appendChild() {} } };
  g.Stats = class { constructor() { this.dom = {}; } showPanel() {} begin() {} end() {} };
  g.ResizeObserver = class {
    constructor(cb) { this.cb = cb; }
    observe(target) { setImmediate(() => this.cb([{ target, contentBoxSize: [{ inlineSize: width, blockSize: height }] }])); }
  };
  let stopped = false;
  g.requestAnimationFrame = (cb) => {
    if (device.frames >= maxFrames) { if (!stopped) { stopped = true; if (onDone) onDone(device, canvas); } return 0; }
    setImmediate(cb);
    return 1;
  };
  if (root) {
    g.fetch = async (p) => {
      const fs = await import('fs');
      const path = await import('path');
      const file = path.resolve(root, String(p).replace(/^\.\//, ''));
      return { text: async () => fs.readFileSync(file, 'utf8'), ok: true };
    };
  }
  if (!g.performance) import('perf_hooks').then((m) => { g.performance = m.performance; });
  return { device, canvas };
}
