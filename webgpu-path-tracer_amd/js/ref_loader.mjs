// ESM loader hooks (Node 12: --experimental-loader) that let the REFERENCE's unchanged browser modules load
// under Node: CDN imports are remapped to local modules (nothing is fetched), the reference directory is
// treated as ESM although it has no package.json, and — on Node < 14 only — the three optional-chaining
// expressions in the reference (lib/BVH/bvhBuilder.js:24-25, webgpu-utils.js:191) are rewritten.
// The reference root comes from $PTMI_REFERENCE_ROOT.
import { pathToFileURL } from 'url';
import path from 'path';

const here = path.dirname(new URL(import.meta.url).pathname);
const GLM = pathToFileURL(path.join(here, 'glmatrix.mjs')).href;
const STUB = pathToFileURL(path.join(here, 'empty_module.mjs')).href;
const ROOT = pathToFileURL(path.resolve(process.env.PTMI_REFERENCE_ROOT || '/root/reference')).href + '/';

export async function resolve(spec, ctx, next) {
  if (spec.startsWith('https://cdn.skypack.dev/gl-matrix')) return { url: GLM };
  if (spec.startsWith('https://')) return { url: STUB };
  return next(spec, ctx, next);
}
export async function getFormat(url, ctx, next) {
  if (url.startsWith(ROOT) && url.endsWith('.js')) return { format: 'module' };
  return next(url, ctx, next);
}
export async function transformSource(src, ctx, next) {
  if (!ctx.url.startsWith(ROOT)) return next(src, ctx, next);
  let s = src.toString();
  if (ctx.url.endsWith('/lib/BVH/bvhBuilder.js')) {
    s = s.replace('obj[10]?.id ?? -1', '((obj[10] != null && obj[10].id != null) ? obj[10].id : -1)')
         .replace('obj[3]?.id ?? -1', '((obj[3] != null && obj[3].id != null) ? obj[3].id : -1)');
  }
  if (ctx.url.endsWith('/webgpu-utils.js')) s = s.replace('adapter?.requestDevice()', 'adapter.requestDevice()');
  return { source: s };
}
