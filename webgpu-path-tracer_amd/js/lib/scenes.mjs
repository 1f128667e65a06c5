// Canonical scenes (SURVEY.md §8d), the JS twin of webgpu-path-tracer_amd/scenes.py.
import { Scene } from './scene.mjs';

export const DRAGON_MAT = ['dragonMat', 0, [0.0, 0.37, 0.20], [0.0, 0.95, 0.95], [0, 0, 0], 0.4, 0.3, 2.5];   // lib/scene.js:166
export const CAMERAS = { default: [[0.5, 0, 2.5], [0.5, 0, 0]], cornell: [[0, 0, 2.5], [0, 0, 0]], oblique: [[1.2, 0.4, 2.1], [0.1, -0.2, 0]] };

function cornellMaterials(sc) {   // lib/scene.js:107-113
  sc.add_material('red', 0, [0.75, 0.1, 0.1], [0.75, 0.1, 0.1], [0, 0, 0], 0.05, 0.95, 0);
  sc.add_material('green', 0, [0.05, 0.55, 0.05], [0.05, 0.55, 0.05], [0, 0, 0], 0.05, 0.95, 0);
  sc.add_material('blue', 0, [0.05, 0.05, 0.55], [0.05, 0.05, 0.55], [0, 0, 0], 0.05, 0.95, 0);
  sc.add_material('white', 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.05, 0.95, 0);
  sc.add_material('glossywhite', 0, [0.76, 0.70, 0.51], [0.76, 0.70, 0.51], [0, 0, 0], 0.3, 0.1, 0);
  sc.add_material('black', 0, [0.2, 0.2, 0.2], [0.2, 0.2, 0.2], [0, 0, 0], 0.05, 0.95, 0);
  sc.add_material('glass', 1, [0.95, 0.95, 0.95], [0, 0, 0], [0, 0, 0], 0, 0, 0);
}

export class CornellScene extends Scene {
  constructor(spheresFn = null, meshesFn = null) { CornellScene.pending = [spheresFn, meshesFn]; super(); }
  create_spheres() {
    [this.spheresFn, this.meshesFn] = CornellScene.pending;
    this.add_material('default', 0, [1, 0, 0], [0, 0, 0], [0, 0, 0], 0, 0, 0);
    if (this.spheresFn) this.spheresFn(this);
    this.objs.push(...this.spheres);
  }
  create_quads() {
    cornellMaterials(this);
    const d = this.material_dict;
    this.add_quad([-0.35, 0.9999, -0.3], [0.7, 0, 0], [0, 0, 0.6], this.add_material('light', 0, [0, 0, 0], [0, 0, 0], [10, 10, 10], 0, 0, 0));
    this.add_quad([-1, -1, -1], [2, 0, 0], [0, 2, 0], d.black);
    this.add_quad([-1, -1, 1], [0, 0, -2], [0, 2, 0], d.red);
    this.add_quad([1, -1, -1], [0, 0, 2], [0, 2, 0], d.green);
    this.add_quad([-1, 1, -1], [2, 0, 0], [0, 0, 2], d.white);
    this.add_quad([1, -1, -1], [-2, 0, 0], [0, 0, 2], d.glossywhite);
    this.lights.push(this.quads[0]);
    this.objs.push(...this.quads);
  }
  create_meshes() { if (this.meshesFn) this.meshesFn(this); this.finish_meshes(); }
}

export function c1Scene() {
  return new CornellScene((sc) => {
    sc.add_sphere([-0.5, -0.7, -0.5], 0.3, sc.add_material('mirror_ball', 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0, 0));
    sc.add_sphere([0.6, -0.75, 0.5], 0.25, sc.add_material('glass_ball', 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5));
  });
}
export function c2Scene(monkey) {
  return new CornellScene(null, (sc) => {
    const m = sc.add_mesh(monkey, sc.add_material(...DRAGON_MAT));
    m.transform.update(m.transform.scale(0.6, 0.6, 0.6), m.transform.translate(0, -0.4, 0));
  });
}
export function c2mScene(ico, cube) {
  return new CornellScene((sc) => {
    sc.add_sphere([-0.45, -0.6, 0.45], 0.3, sc.add_material('fog', 3, [0.56, 0.93, 0.56], [0, 0, 0], [0, 0, 0], 0.00001, -1 / 4, 0));
    sc.add_sphere([-0.45, -0.6, 0.45], 0.3, sc.add_material('gg4t', 2, [1, 1, 1], [0, 0, 0], [0, 0, 0], 0, 0, 1.5));
  }, (sc) => {
    const a = sc.add_mesh(ico, sc.add_material(...DRAGON_MAT));
    const b = sc.add_mesh(cube, sc.add_material('box2', 1, [0.95, 0.95, 0.95], [0.95, 0.95, 0.95], [0, 0, 0], 0, 0.05, 1.5));
    a.transform.update(a.transform.scale(0.35, 0.35, 0.35), a.transform.rotate(Math.PI / 4, [0, 1, 0]), a.transform.translate(0.45, -0.64, 0));
    b.transform.update(b.transform.scale(0.2, 0.3, 0.2), b.transform.rotate(-Math.PI / 4, [1, 1, 0]), b.transform.translate(-0.1, -0.55, -0.4));
  });
}

// renderer.js:78-87 call order -> the seven uploadable arrays
export async function sceneBuffers(scene) {
  await scene.init_mesh_data();
  scene.create_meshes();
  const out = { meshes: scene.get_meshes(), spheres: scene.get_spheres(), quads: scene.get_quads(), materials: scene.get_materials(), transforms: scene.get_transforms() };
  scene.create_bvh();
  out.bvh = scene.get_bvh();
  out.triangles = scene.get_triangles();
  return out;
}
