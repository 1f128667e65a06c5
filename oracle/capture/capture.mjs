// Generates tests/golden/* by running the reference's own host JavaScript (lib/scene.js and friends,
// imported unchanged from /root/reference) under Node:
//     cd oracle/capture && node --experimental-loader ./loader.mjs capture.mjs
// Outputs are DATA only (typed-array bytes + a JSON manifest).  Scenes other than "default" are made by
// overriding Scene.prototype.create_spheres / create_quads / create_meshes, the three methods in which
// the reference hard-codes its scene (lib/scene.js:36-251).
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { performance } from 'perf_hooks';

const REF = '/root/reference/';
const OUT = path.resolve(path.dirname(new URL(import.meta.url).pathname), '../../tests/golden');
globalThis.performance = performance;
globalThis.fetch = async (p) => ({ text: async () => fs.readFileSync(REF + p.replace(/^\.\//, ''), 'utf8') });
const quiet = console.log; console.log = () => {};

const sha = (ta) => crypto.createHash('sha256').update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest('hex');
const manifest = {};
function save(scene, name, ta, keepBytes = true) {
  manifest[scene] = manifest[scene] || {};
  manifest[scene][name] = { dtype: ta instanceof Int32Array ? 'i32' : 'f32', length: ta.length, sha256: sha(ta), stored: keepBytes };
  if (keepBytes) fs.writeFileSync(path.join(OUT, `${scene}_${name}.bin`), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength));
}

async function main() {
  const { Scene } = await import('file://' + REF + 'lib/scene.js');
  const { Sphere } = await import('file://' + REF + 'lib/primitives/sphere.js');
  const { Quad } = await import('file://' + REF + 'lib/primitives/quad.js');
  const { Mesh } = await import('file://' + REF + 'lib/primitives/mesh.js');
  const { ObjReader } = await import('file://' + REF + 'lib/primitives/objReader.js');
  const { mat4 } = await import('../../webgpu-path-tracer_amd/js/glmatrix.mjs');
  const orig = { s: Scene.prototype.create_spheres, q: Scene.prototype.create_quads, m: Scene.prototype.create_meshes, i: Scene.prototype.init_mesh_data };

  // exact call order of renderer.js:78-87
  async function dump(name, scene, keep = true) {
    await scene.init_mesh_data();
    scene.create_meshes();
    save(name, 'meshes', scene.get_meshes(), keep);
    save(name, 'spheres', scene.get_spheres(), keep);
    save(name, 'quads', scene.get_quads(), keep);
    save(name, 'materials', scene.get_materials(), keep);
    save(name, 'transforms', scene.get_transforms(), keep);
    if (scene.triangles.length > 0) {
      scene.create_bvh();
      save(name, 'bvh', scene.get_bvh(), keep);
      save(name, 'triangles', scene.get_triangles(), keep);
    }
  }

  // ---- default: the reference scene as shipped (19 spheres, 8 quads, rotated cube) ----
  await dump('default', new Scene());

  // ---- Cornell box pieces shared by c1 / c2 / c2m (SURVEY.md §8d) ----
  function cornell_quads() {
    this.add_material('red', 0, [0.75, 0.1, 0.1], [0.75, 0.1, 0.1], [0, 0, 0], 0.05, 0.95, 0);
    this.add_material('green', 0, [0.05, 0.55, 0.05], [0.05, 0.55, 0.05], [0, 0, 0], 0.05, 0.95, 0);
    this.add_material('blue', 0, [0.05, 0.05, 0.55], [0.05, 0.05, 0.55], [0, 0, 0], 0.05, 0.95, 0);
    this.add_material('white', 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.05, 0.95, 0);
    this.add_material('glossywhite', 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.3, 0.1, 0);
    this.add_material('black', 0, [0.2, 0.2, 0.2], [0.2, 0.2, 0.2], [0, 0, 0], 0.05, 0.95, 0);
    this.add_material('glass', 1, [0.95, 0.95, 0.95], [0, 0, 0], [0, 0, 0], 0, 0, 0);
    this.quads.push(
      new Quad([-0.35, 0.9999, -0.3], [0.7, 0, 0], [0, 0, 0.6], this.global_id++, this.quad_id++, this.add_material('light', 0, [0, 0, 0], [0, 0, 0], [10, 10, 10], 0, 0, 0)),
      new Quad([-1, -1, -1], [2, 0, 0], [0, 2, 0], this.global_id++, this.quad_id++, this.material_dict['black']),
      new Quad([-1, -1, 1], [0, 0, -2], [0, 2, 0], this.global_id++, this.quad_id++, this.material_dict['red']),
      new Quad([1, -1, -1], [0, 0, 2], [0, 2, 0], this.global_id++, this.quad_id++, this.material_dict['green']),
      new Quad([-1, 1, -1], [2, 0, 0], [0, 0, 2], this.global_id++, this.quad_id++, this.material_dict['white']),
      new Quad([1, -1, -1], [-2, 0, 0], [0, 0, 2], this.global_id++, this.quad_id++, this.material_dict['glossywhite']),
    );
    this.lights.push(this.quads[0]);
    this.objs.push(this.quads.flat());
  }
  function no_spheres() {
    this.add_material('default', 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0);
    this.objs.push(this.spheres.flat());
  }
  function finish_meshes() {   // tail of lib/scene.js create_meshes (:245-248)
    this.triangles = this.meshes.map(mesh => mesh.triangles).flat();
    this.meshes.forEach(mesh => { mesh.calc_bbox(mesh.transform); });
    this.triangle_data = this.triangles.map(tri => tri.data);
    this.objs.push(this.meshes.flat());
  }
  function add_mesh(scene, data, material_id) {
    const m = new Mesh(data, scene.triangle_offset, scene.global_id++, scene.mesh_id++, scene.triangle_id, material_id);
    scene.meshes.push(m);
    scene.triangle_id += m.numTriangle;
    scene.triangle_offset += m.numTriangle;
    return m;
  }

  // ---- c1: Cornell + mirror sphere + glass sphere, no triangles ----
  Scene.prototype.create_spheres = function () {
    this.add_material('default', 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0);
    this.spheres.push(
      new Sphere([-0.5, -0.7, -0.5], 0.3, this.global_id++, this.sphere_id++, this.add_material('mirror_ball', 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0, 0)),
      new Sphere([0.6, -0.75, 0.5], 0.25, this.global_id++, this.sphere_id++, this.add_material('glass_ball', 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5)),
    );
    this.objs.push(this.spheres.flat());
  };
  Scene.prototype.create_quads = cornell_quads;
  Scene.prototype.create_meshes = function () { finish_meshes.call(this); };
  Scene.prototype.init_mesh_data = async function () { this.mesh_data = {}; };
  await dump('c1', new Scene());

  // ---- c2: Cornell + monkey_968, scale 0.6, translate (0,-0.4,0), dragonMat ----
  Scene.prototype.create_spheres = no_spheres;
  Scene.prototype.init_mesh_data = async function () {
    this.mesh_data = { monkey1: await ObjReader.load_model('./assets/monkey_968.obj') };
  };
  Scene.prototype.create_meshes = function () {
    const mat = this.add_material('dragonMat', 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5);
    const m = add_mesh(this, this.mesh_data['monkey1'], mat);
    m.transform.update(m.transform.scale(0.6, 0.6, 0.6), m.transform.translate(0, -0.4, 0));
    finish_meshes.call(this);
  };
  await dump('c2', new Scene());

  // ---- c2m: Cornell + two meshes (icosphere with the dragon transform of lib/scene.js:216-220, and a
  //      cube rotated about a non-unit axis) + one fog sphere pair: pins multi-mesh ids and fromRotation ----
  Scene.prototype.create_spheres = function () {
    this.add_material('default', 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0);
    this.spheres.push(
      new Sphere([-0.45, -0.6, 0.45], 0.3, this.global_id++, this.sphere_id++, this.add_material('fog', 3, [0.56, 0.93, 0.56], [0, 0, 0], [0, 0, 0], 0.00001, -1 / 4, 0)),
      new Sphere([-0.45, -0.6, 0.45], 0.3, this.global_id++, this.sphere_id++, this.add_material('gg4t', 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5)),
    );
    this.objs.push(this.spheres.flat());
  };
  Scene.prototype.init_mesh_data = async function () {
    this.mesh_data = { ico: await ObjReader.load_model('./assets/icosphere.obj'), cube: await ObjReader.load_model('./assets/cube.obj') };
  };
  Scene.prototype.create_meshes = function () {
    const mat = this.add_material('dragonMat', 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5);
    const a = add_mesh(this, this.mesh_data['ico'], mat);
    const b = add_mesh(this, this.mesh_data['cube'], this.add_material('box2', 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0.05, 1.5));
    a.transform.update(a.transform.scale(0.35, 0.35, 0.35), a.transform.rotate(Math.PI / 4, [0, 1, 0]), a.transform.translate(0.45, -0.64, 0));
    b.transform.update(b.transform.scale(0.2, 0.3, 0.2), b.transform.rotate(-Math.PI / 4, [1, 1, 0]), b.transform.translate(-0.1, -0.55, -0.4));
    finish_meshes.call(this);
  };
  await dump('c2m', new Scene());

  // ---- BVH-only pins on larger meshes (hashes only; bytes not stored) ----
  for (const [tag, file] of [['m5802', 'monkey_5802.obj'], ['m15744', 'monkey_smooth_15744.obj']]) {
    Scene.prototype.create_spheres = no_spheres;
    Scene.prototype.init_mesh_data = async function () { this.mesh_data = { m: await ObjReader.load_model('./assets/' + file) }; };
    Scene.prototype.create_meshes = function () {
      const mat = this.add_material('dragonMat', 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5);
      const m = add_mesh(this, this.mesh_data['m'], mat);
      m.transform.update(m.transform.scale(1.1, 1.1, 1.1), m.transform.rotate(Math.PI / 4, [0, 1, 0]), m.transform.translate(0.65, -0.64, 0));
      finish_meshes.call(this);
    };
    await dump(tag, new Scene(), false);
  }

  // ---- the reference's OTHER builder: BVH.generate_bvh_heirarchy_SAH (lib/BVH/bvhNode.js:108-283) is never called by the
  //      reference (create_bvh -> generate_bvh_heirarchy); routing the static create_bvh to it runs that code unchanged
  //      through the same populate_links / flattenBVH.  Pins ptmi_build_bvh_sah. ----
  {
    const { BVH } = await import('file://' + REF + 'lib/BVH/bvhNode.js');
    const median = BVH.create_bvh;
    BVH.create_bvh = (objs) => ({ bvh: BVH.generate_bvh_heirarchy_SAH(objs, 0, objs.length - 1), objs: objs });
    for (const [tag, file, keep] of [['c2sah', 'monkey_968.obj', true], ['m5802sah', 'monkey_5802.obj', false], ['m15744sah', 'monkey_smooth_15744.obj', false]]) {
      Scene.prototype.create_spheres = no_spheres;
      Scene.prototype.init_mesh_data = async function () { this.mesh_data = { m: await ObjReader.load_model('./assets/' + file) }; };
      Scene.prototype.create_meshes = function () {
        const mat = this.add_material('dragonMat', 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5);
        const m = add_mesh(this, this.mesh_data['m'], mat);
        if (keep) m.transform.update(m.transform.scale(0.6, 0.6, 0.6), m.transform.translate(0, -0.4, 0));
        else m.transform.update(m.transform.scale(1.1, 1.1, 1.1), m.transform.rotate(Math.PI / 4, [0, 1, 0]), m.transform.translate(0.65, -0.64, 0));
        finish_meshes.call(this);
      };
      const sc = new Scene();
      await sc.init_mesh_data();
      sc.create_meshes();
      sc.create_bvh();
      save(tag, 'bvh', sc.get_bvh(), keep);
      save(tag, 'triangles', sc.get_triangles(), false);
    }
    BVH.create_bvh = median;
  }

  // ---- cameras: mat4.targetTo exactly as lib/camera.js:32 calls it (raw JS arrays, not the f32 copies) ----
  const cams = {};
  for (const [k, eye, center, up] of [['default', [0.5, 0, 2.5], [0.5, 0, 0], [0, 1, 0]], ['cornell', [0, 0, 2.5], [0, 0, 0], [0, 1, 0]], ['oblique', [1.2, 0.4, 2.1], [0.1, -0.2, 0], [0, 1, 0]]]) {
    const m = mat4.create();
    mat4.targetTo(m, eye, center, up);
    cams[k] = { eye, center, up, viewMatrix: Array.from(m) };
  }
  manifest.cameras = cams;

  // ---- OBJ parse pin: de-indexed arrays of cube.obj straight from ObjReader.load_model ----
  const cube = await ObjReader.load_model('./assets/cube.obj');
  save('objcube', 'vertices', cube.vertices);
  save('objcube', 'normals', cube.normals);

  fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest, null, 1));
  Object.assign(Scene.prototype, { create_spheres: orig.s, create_quads: orig.q, create_meshes: orig.m, init_mesh_data: orig.i });
  console.log = quiet;
  console.log('wrote goldens to', OUT);
}
main().catch((e) => { console.error(e); process.exit(1); });
