// Paste into the reference page's console (WebGPU browser) after N frames; needs `window.__renderer = renderer` in
// index.js start().  Copies the accumulation buffer (created with COPY_SRC, webgpu-utils.js:43-55) to a mappable buffer
// and downloads it as raw little-endian f32 RGBA sums: reference.f32
(async () => {
  const r = window.__renderer, dev = r.webGPU.device, src = r.buffers.frameBuffer.buffer;   // renderer.js:100
  const dst = dev.createBuffer({ size: src.size, usage: GPUBufferUsage.COPY_DST | GPUBufferUsage.MAP_READ });
  const enc = dev.createCommandEncoder();
  enc.copyBufferToBuffer(src, 0, dst, 0, src.size);
  dev.queue.submit([enc.finish()]);
  await dst.mapAsync(GPUMapMode.READ);
  const blob = new Blob([dst.getMappedRange().slice(0)], { type: 'application/octet-stream' });
  const a = document.createElement('a');
  a.href = URL.createObjectURL(blob); a.download = 'reference.f32'; a.click();
  console.log('frames accumulated:', r.frameNum, 'bytes:', src.size);
})();
