// check_host.mjs <golden_dir> <assets_dir> — builds c1/c2/c2m with lib/scene.mjs (JS builder and native builder) and
// compares every array with the goldens byte for byte; prints a JSON report.  Used by tests/test_js_host.py.
import fs from 'fs';
import path from 'path';
import { ObjReader, Camera } from './lib/scene.mjs';
import { c1Scene, c2Scene, c2mScene, sceneBuffers, CAMERAS } from './lib/scenes.mjs';
import { loadNative } from './ptmi.mjs';

const [golden, assets] = process.argv.slice(2);
const names = ['spheres', 'quads', 'triangles', 'meshes', 'transforms', 'materials', 'bvh'];
const rd = (f) => { if (!fs.existsSync(f)) return new Uint8Array(0); const b = fs.readFileSync(f); return new Uint8Array(b.buffer, b.byteOffset, b.length); };
const same = (a, g) => { const x = new Uint8Array(a.buffer, a.byteOffset, a.byteLength); if (x.length !== g.length) return false; for (let i = 0; i < x.length; i++) if (x[i] !== g[i]) return false; return true; };
const obj = (f) => ObjReader.parse(fs.readFileSync(path.join(assets, f), 'utf8'));
const report = {};
async function check(tag, make) {
  for (const useNative of [false, true]) {
    const sc = make();
    if (useNative) sc.native = loadNative();
    const b = await sceneBuffers(sc);
    for (const n of names) report[`${tag}.${n}.${useNative ? 'native' : 'js'}`] = same(b[n], rd(path.join(golden, `${tag}_${n}.bin`)));
  }
}
(async () => {
  await check('c1', () => c1Scene());
  if (assets && fs.existsSync(path.join(assets, 'monkey_968.obj'))) {
    await check('c2', () => c2Scene(obj('monkey_968.obj')));
    await check('c2m', () => c2mScene(obj('icosphere.obj'), obj('cube.obj')));
  }
  if (assets && fs.existsSync(path.join(assets, 'monkey_968.obj'))) {   // the opt-in SAH builder through the addon
    const sc = c2Scene(obj('monkey_968.obj'));
    sc.native = loadNative();
    sc.useSAH = true;
    const b = await sceneBuffers(sc);
    report['c2sah.bvh.native'] = same(b.bvh, rd(path.join(golden, 'c2sah_bvh.bin')));
  }
  if (assets && fs.existsSync(path.join(assets, 'monkey_968.obj'))) {
    for (const f of ['cube.obj', 'monkey_968.obj', 'hole.obj']) {
      const text = fs.readFileSync(path.join(assets, f), 'utf8');
      const a = ObjReader.parse(text), b = loadNative().parseObj(text);
      report[`objparse.${f}`] = same(a.vertices, new Uint8Array(b.vertices.buffer)) && same(a.normals, new Uint8Array(b.normals.buffer));
    }
  }
  const man = JSON.parse(fs.readFileSync(path.join(golden, 'manifest.json'), 'utf8'));
  for (const [k, [eye, center]] of Object.entries(CAMERAS)) {
    const c = new Camera(); c.set_camera(eye, center, [0, 1, 0]);
    report[`camera.${k}`] = Array.from(c.viewMatrix).every((v, i) => v === Math.fround(man.cameras[k].viewMatrix[i]));
  }
  console.log(JSON.stringify(report));
})().catch((e) => { console.error(e); process.exit(1); });
