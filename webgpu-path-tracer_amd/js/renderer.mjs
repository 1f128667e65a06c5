// renderer.mjs — the Node host's Renderer: same method surface as the reference's renderer.js (create,
// createShaderModule, initBuffers, createComputePipeline, createRenderPipeline, setRenderParameters,
// renderAnimation, renderSingleFrame) written against the WebGPU-shaped device of webgpu_node.mjs.
import { GPUBufferUsage } from './webgpu_node.mjs';

export function flattenToFloat32Array(obj) {   // renderer.js:265-278
  const array = [];
  for (const key in obj) {
    if (Array.isArray(obj[key]) || obj[key] instanceof Float32Array) array.push(...obj[key]); else array.push(obj[key]);
  }
  return new Float32Array(array);
}

export class Renderer {
  constructor(device) { this.device = device; this.frameNum = 0.0; }
  static async create(device) { return new Renderer(device); }

  // The WGSL is not compiled here; a header in the reference's syntax carries the run-time knobs.
  createShaderModule(params = {}) {
    const p = Object.assign({ NUM_SAMPLES: 1, MAX_BOUNCES: 100, STRATIFY: false, IMPORTANCE_SAMPLING: false, STACK_SIZE: 20, background: [0, 1, 1] }, params);
    const code = `const NUM_SAMPLES = ${p.NUM_SAMPLES};\nconst MAX_BOUNCES = ${p.MAX_BOUNCES};\nconst STRATIFY = ${p.STRATIFY};\nconst IMPORTANCE_SAMPLING = ${p.IMPORTANCE_SAMPLING};\n` +
      `const STACK_SIZE = ${p.STACK_SIZE};\nlet background_color = vec3f(${p.background.join(', ')});\n`;
    return this.device.createShaderModule({ code });
  }

  _buffer(label, array, usage) {   // webgpu-utils.js:15-69
    const buffer = this.device.createBuffer({ label, size: Math.max(array.byteLength, 0), usage });
    this.device.queue.writeBuffer(buffer, 0, array);
    return { data: array, buffer };
  }

  async initBuffers(scene, camera, WIDTH, HEIGHT, prebuilt = null) {   // renderer.js:68-103
    this.uniforms = { screenDims: [WIDTH, HEIGHT], frameNum: 0, resetBuffer: 0, viewMatrix: camera.viewMatrix };
    let b = prebuilt;
    if (!b) {
      await scene.init_mesh_data();
      scene.create_meshes();
      b = { meshes: scene.get_meshes(), spheres: scene.get_spheres(), quads: scene.get_quads(), materials: scene.get_materials(), transforms: scene.get_transforms() };
      scene.create_bvh();
      b.bvh = scene.get_bvh();
      b.triangles = scene.get_triangles();
    }
    const S = GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_DST;
    this.buffers = {
      uniforms: this._buffer('uniformBuffer', flattenToFloat32Array(this.uniforms), GPUBufferUsage.UNIFORM | GPUBufferUsage.COPY_DST),
      spheres: this._buffer('sphere buffer', b.spheres, S), meshes: this._buffer('mesh buffer', b.meshes, S), quads: this._buffer('quads buffer', b.quads, S),
      materials: this._buffer('material buffer', b.materials, S), transforms: this._buffer('transform buffer', b.transforms, S),
      bvh: this._buffer('bvh buffer', b.bvh, S), triangles: this._buffer('tri buffer', b.triangles, S),
      frameBuffer: this._buffer('frameNum buffer', new Float32Array(WIDTH * HEIGHT * 4), S | GPUBufferUsage.COPY_SRC),
    };
    this.WIDTH = WIDTH; this.HEIGHT = HEIGHT; this.camera = camera;
  }

  createComputePipeline(module) {   // renderer.js:105-124
    this.computePipeline = this.device.createComputePipeline({ label: 'Compute pipeline', layout: 'auto', compute: { module, entryPoint: 'computeFrameBuffer' } });
    const b = this.buffers;
    this.bindGroupCompute = this.device.createBindGroup({
      label: 'bindGroup for work buffer', layout: this.computePipeline.getBindGroupLayout(0),
      entries: [[0, b.uniforms], [1, b.spheres], [2, b.quads], [3, b.frameBuffer], [5, b.triangles], [6, b.meshes], [7, b.transforms], [8, b.materials], [9, b.bvh]]
        .map(([binding, x]) => ({ binding, resource: { buffer: x.buffer } })),
    });
  }
  createRenderPipeline(module) {   // renderer.js:126-139
    this.renderPipeline = this.device.createRenderPipeline({ label: 'render pipeline', layout: 'auto', vertex: { module, entryPoint: 'vs' }, fragment: { module, entryPoint: 'fs' } });
    this.bindGroup = this.device.createBindGroup({ layout: this.renderPipeline.getBindGroupLayout(0),
      entries: [{ binding: 0, resource: { buffer: this.buffers.uniforms.buffer } }, { binding: 3, resource: { buffer: this.buffers.frameBuffer.buffer } }] });
  }
  setRenderParameters(renderParams, camera, WIDTH, HEIGHT) { this.renderParams = renderParams; this.camera = camera; this.frameNum = 0.0; }

  // one iteration of renderer.js:163-215 (no frame pacing under Node)
  renderSingleFrame() {
    this.frameNum += 1.0;
    this.uniforms.frameNum = this.frameNum;
    this.uniforms.resetBuffer = (this.camera.MOVING || this.camera.keyPress) ? 1 : 0;
    if (this.camera.MOVING || this.camera.keyPress) { this.frameNum = 1; this.camera.keyPress = 0; }
    this.uniforms.viewMatrix = this.camera.viewMatrix;
    this.buffers.uniforms.data = flattenToFloat32Array(this.uniforms);
    this.device.queue.writeBuffer(this.buffers.uniforms.buffer, 0, this.buffers.uniforms.data);
    let enc = this.device.createCommandEncoder({ label: 'compute encoder' });
    const pass = enc.beginComputePass({ label: 'compute pass' });
    pass.setPipeline(this.computePipeline);
    pass.setBindGroup(0, this.bindGroupCompute);
    pass.dispatchWorkgroups((this.WIDTH * this.HEIGHT) / 64 + 1, 1, 1);
    pass.end();
    this.device.queue.submit([enc.finish()]);
    enc = this.device.createCommandEncoder({ label: 'render encoder' });
    const rp = enc.beginRenderPass({ label: 'renderPass', colorAttachments: [{ clearValue: [0.3, 0.3, 0.3, 1], loadOp: 'clear', storeOp: 'store' }] });
    rp.setPipeline(this.renderPipeline);
    rp.setBindGroup(0, this.bindGroup);
    rp.draw(6);
    rp.end();
    this.device.queue.submit([enc.finish()]);
  }
  renderAnimation(frames = 1) { for (let i = 0; i < frames; i++) this.renderSingleFrame(); }
}
