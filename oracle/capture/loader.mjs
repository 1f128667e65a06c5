// ESM loader hooks for Node 12 (--experimental-loader): lets capture.mjs import the reference's host
// JavaScript UNCHANGED from /root/reference, offline.  Nothing is fetched: URL specifiers are remapped
// before Node's resolver sees them.  Used only in the build container to generate tests/golden/*.
import { pathToFileURL } from 'url';
import path from 'path';
const here = path.dirname(new URL(import.meta.url).pathname);
const GLM = pathToFileURL(path.resolve(here, '../../webgpu-path-tracer_amd/js/glmatrix.mjs')).href;
const STUB = pathToFileURL(path.resolve(here, 'stub.mjs')).href;

export async function resolve(spec, ctx, next) {
  if (spec.startsWith('https://cdn.skypack.dev/gl-matrix')) return { url: GLM };
  if (spec.startsWith('https://')) return { url: STUB };
  return next(spec, ctx, next);
}
export async function getFormat(url, ctx, next) {
  if (url.startsWith('file:///root/reference/') && url.endsWith('.js')) return { format: 'module' };
  return next(url, ctx, next);
}
export async function transformSource(src, ctx, next) {
  // Node 12 lacks ?. and ?? (used at lib/BVH/bvhBuilder.js:24-25): rewrite those two expressions only.
  if (ctx.url.endsWith('/lib/BVH/bvhBuilder.js')) {
    return {
      source: src.toString()
        .replace('obj[10]?.id ?? -1', '((obj[10] != null && obj[10].id != null) ? obj[10].id : -1)')
        .replace('obj[3]?.id ?? -1', '((obj[3] != null && obj[3].id != null) ? obj[3].id : -1)'),
    };
  }
  return next(src, ctx, next);
}
