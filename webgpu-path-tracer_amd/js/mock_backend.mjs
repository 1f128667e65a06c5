// A recording backend with the surface of Ptmi (ptmi.mjs), for exercising the WebGPU shim without a GPU.
export class MockBackend {
  constructor() { this.calls = []; this.uploads = {}; this.frames = []; this.params = null; this.width = 0; this.height = 0; }
  setParams(p) { this.params = Object.assign({}, p); this.calls.push(['setParams']); return p; }
  upload(name, arr) { this.uploads[name] = arr; this.calls.push(['upload', name, arr.length]); }
  resize(w, h) { this.width = w; this.height = h; this.calls.push(['resize', w, h]); }
  clear() { this.calls.push(['clear']); }
  writeFramebuffer() { this.calls.push(['writeFramebuffer']); }
  renderFrame(u) { this.frames.push(Array.from(u)); this.calls.push(['renderFrame', u[2], u[3]]); }
  resolveRGBA8(fn, out) { this.calls.push(['resolve', fn]); return out || new Uint8Array(this.width * this.height * 4); }
  synchronize() {}
}
