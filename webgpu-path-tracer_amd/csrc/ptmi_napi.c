/*
 * ptmi_napi.c — raw N-API (node_api.h, no node-addon-api) binding of include/ptmi.h for the Node host.
 *
 * The reference's host is browser JavaScript talking to WebGPU (webgpu-utils.js); under Node the same
 * typed arrays go through this addon instead.  The addon dlopen()s libptmi.so at load time (path from
 * $PTMI_LIB, else ../libptmi.so next to this file), so it builds with plain gcc:
 *     gcc -O2 -fPIC -shared -I/usr/include/node -Iinclude -o js/ptmi.node csrc/ptmi_napi.c -ldl
 * Every failing ptmi_* call becomes a thrown JS Error carrying ptmi_last_error().  Typed arrays are
 * only read/written during the call (ptmi copies on upload), no external ArrayBuffer lifetimes.
 */
#define NAPI_VERSION 4
#define _GNU_SOURCE
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptmi.h"

static void* g_lib;
#define FN(name) static __typeof__(name)* p_##name;
FN(ptmi_version)
FN(ptmi_last_error)
FN(ptmi_create)
FN(ptmi_create_multi)
FN(ptmi_prepare)
FN(ptmi_destroy)
FN(ptmi_default_params)
FN(ptmi_set_params)
FN(ptmi_get_params)
FN(ptmi_upload)
FN(ptmi_resize)
FN(ptmi_clear_framebuffer)
FN(ptmi_set_shard)
FN(ptmi_render_frame)
FN(ptmi_render)
FN(ptmi_synchronize)
FN(ptmi_read_framebuffer)
FN(ptmi_write_framebuffer)
FN(ptmi_resolve_rgba8)
FN(ptmi_set_counters)
FN(ptmi_set_timing)
FN(ptmi_get_stats)
FN(ptmi_reset_stats)
FN(ptmi_build_bvh)
FN(ptmi_build_bvh_sah)
FN(ptmi_build_bvh_device)
FN(ptmi_build_scene_bvh)
FN(ptmi_build_scene_bvh_sah)
FN(ptmi_obj_parse)
FN(ptmi_free)
FN(ptmi_device_count)
FN(ptmi_reduce_info)

static int load_lib(char* err, size_t errlen) {
  if (g_lib) return 0;
  char path[4096];
  const char* env = getenv("PTMI_LIB");
  if (env && *env) {
    snprintf(path, sizeof path, "%s", env);
  } else {
    Dl_info info;
    if (!dladdr((void*)&load_lib, &info) || !info.dli_fname) {
      snprintf(err, errlen, "cannot locate ptmi.node on disk");
      return -1;
    }
    snprintf(path, sizeof path, "%s", info.dli_fname);
    char* slash = strrchr(path, '/');
    if (slash) *slash = 0;
    strncat(path, "/../libptmi.so", sizeof path - strlen(path) - 1);
  }
  g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!g_lib) {
    snprintf(err, errlen, "dlopen(%s): %s", path, dlerror());
    return -1;
  }
#define LOAD(name)                                            \
  p_##name = (__typeof__(name)*)dlsym(g_lib, #name);          \
  if (!p_##name) {                                            \
    snprintf(err, errlen, "libptmi.so lacks symbol " #name); \
    return -1;                                                \
  }
  LOAD(ptmi_version) LOAD(ptmi_last_error) LOAD(ptmi_create) LOAD(ptmi_create_multi) LOAD(ptmi_prepare) LOAD(ptmi_destroy) LOAD(ptmi_default_params) LOAD(ptmi_set_params)
  LOAD(ptmi_get_params) LOAD(ptmi_upload) LOAD(ptmi_resize) LOAD(ptmi_clear_framebuffer) LOAD(ptmi_set_shard) LOAD(ptmi_render_frame)
  LOAD(ptmi_render) LOAD(ptmi_synchronize) LOAD(ptmi_read_framebuffer) LOAD(ptmi_write_framebuffer) LOAD(ptmi_resolve_rgba8)
  LOAD(ptmi_set_counters) LOAD(ptmi_set_timing) LOAD(ptmi_get_stats) LOAD(ptmi_reset_stats) LOAD(ptmi_build_bvh) LOAD(ptmi_build_bvh_sah) LOAD(ptmi_build_bvh_device) LOAD(ptmi_build_scene_bvh) LOAD(ptmi_build_scene_bvh_sah)
  LOAD(ptmi_obj_parse) LOAD(ptmi_free) LOAD(ptmi_device_count) LOAD(ptmi_reduce_info)
  return 0;
}

#define CHECK_NAPI(call)                                   \
  if ((call) != napi_ok) {                                 \
    napi_throw_error(env, NULL, "N-API call failed: " #call); \
    return NULL;                                           \
  }

static napi_value throw_status(napi_env env, ptmi_ctx* c, int st, const char* what) {
  char msg[1024];
  snprintf(msg, sizeof msg, "%s: ptmi status %d: %s", what, st, p_ptmi_last_error(c));
  napi_throw_error(env, NULL, msg);
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "wrong number of arguments");
    return -1;
  }
  return 0;
}

static ptmi_ctx* ctx_of(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "expected a ptmi context handle");
    return NULL;
  }
  ptmi_ctx** box = (ptmi_ctx**)p;
  if (!*box) {
    napi_throw_error(env, NULL, "ptmi context already destroyed");
    return NULL;
  }
  return *box;
}

static int typed(napi_env env, napi_value v, napi_typedarray_type want, const char* what, void** data, size_t* len) {
  bool is = false;
  napi_typedarray_type ty;
  napi_value ab;
  size_t off;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is || napi_get_typedarray_info(env, v, &ty, len, data, &ab, &off) != napi_ok || ty != want) {
    char msg[128];
    snprintf(msg, sizeof msg, "%s: wrong typed array type", what);
    napi_throw_type_error(env, NULL, msg);
    return -1;
  }
  return 0;
}

static void finalize_ctx(napi_env env, void* data, void* hint) {
  (void)env;
  (void)hint;
  ptmi_ctx** box = (ptmi_ctx**)data;
  if (*box) p_ptmi_destroy(*box);
  free(box);
}

static napi_value js_version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  CHECK_NAPI(napi_create_int32(env, p_ptmi_version(), &v));
  return v;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  /* create(deviceId) = one GPU; create([id0, id1, ...]) = one context over several GPUs of the node (ptmi_create_multi):
   * tiles sharded across them, one RCCL reduce inside readFramebuffer */
  ptmi_ctx* c = NULL;
  bool is_array = false;
  CHECK_NAPI(napi_is_array(env, a[0], &is_array));
  if (is_array) {
    uint32_t n = 0;
    CHECK_NAPI(napi_get_array_length(env, a[0], &n));
    if (n < 1 || n > 64) {
      napi_throw_range_error(env, NULL, "create: need 1..64 device ids");
      return NULL;
    }
    int ids[64];
    for (uint32_t i = 0; i < n; i++) {
      napi_value e;
      int32_t v = 0;
      CHECK_NAPI(napi_get_element(env, a[0], i, &e));
      CHECK_NAPI(napi_get_value_int32(env, e, &v));
      ids[i] = v;
    }
    int st = p_ptmi_create_multi(&c, ids, (int)n);
    if (st) return throw_status(env, NULL, st, "ptmi_create_multi");
  } else {
    int32_t dev = 0;
    CHECK_NAPI(napi_get_value_int32(env, a[0], &dev));
    int st = p_ptmi_create(&c, dev);
    if (st) return throw_status(env, NULL, st, "ptmi_create");
  }
  ptmi_ctx** box = (ptmi_ctx**)malloc(sizeof *box);
  *box = c;
  napi_value ext;
  CHECK_NAPI(napi_create_external(env, box, finalize_ctx, NULL, &ext));
  return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  void* p = NULL;
  if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
    ptmi_ctx** box = (ptmi_ctx**)p;
    if (*box) p_ptmi_destroy(*box);
    *box = NULL;
  }
  return NULL;
}

static napi_value set_i32(napi_env env, napi_value o, const char* k, int32_t v) {
  napi_value x;
  napi_create_int32(env, v, &x);
  napi_set_named_property(env, o, k, x);
  return o;
}
static napi_value set_f64(napi_env env, napi_value o, const char* k, double v) {
  napi_value x;
  napi_create_double(env, v, &x);
  napi_set_named_property(env, o, k, x);
  return o;
}

static napi_value params_to_js(napi_env env, const ptmi_params* p) {
  napi_value o, bg;
  CHECK_NAPI(napi_create_object(env, &o));
  set_i32(env, o, "num_samples", p->num_samples);
  set_i32(env, o, "max_bounces", p->max_bounces);
  set_i32(env, o, "stratify", p->stratify);
  set_i32(env, o, "importance_sampling", p->importance_sampling);
  set_i32(env, o, "stack_size", p->stack_size);
  set_f64(env, o, "fov_degrees", p->fov_degrees);
  set_i32(env, o, "frames_in_flight", p->frames_in_flight);
  set_f64(env, o, "tmin", p->tmin);
  set_f64(env, o, "light_mix", p->light_mix);
  CHECK_NAPI(napi_create_array_with_length(env, 3, &bg));
  for (uint32_t i = 0; i < 3; i++) {
    napi_value x;
    napi_create_double(env, p->background[i], &x);
    napi_set_element(env, bg, i, x);
  }
  napi_set_named_property(env, o, "background", bg);
  return o;
}

static napi_value js_default_params(napi_env env, napi_callback_info info) {
  (void)info;
  ptmi_params p;
  p_ptmi_default_params(&p);
  return params_to_js(env, &p);
}

static void read_i32(napi_env env, napi_value o, const char* k, int32_t* dst) {
  bool has = false;
  napi_value v;
  if (napi_has_named_property(env, o, k, &has) == napi_ok && has && napi_get_named_property(env, o, k, &v) == napi_ok) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) == napi_ok) {
      if (t == napi_boolean) {
        bool b;
        napi_get_value_bool(env, v, &b);
        *dst = b ? 1 : 0;
      } else if (t == napi_number) {
        napi_get_value_int32(env, v, dst);
      }
    }
  }
}

static napi_value js_set_params(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  ptmi_params p;
  p_ptmi_get_params(c, &p);
  read_i32(env, a[1], "num_samples", &p.num_samples);
  read_i32(env, a[1], "max_bounces", &p.max_bounces);
  read_i32(env, a[1], "stratify", &p.stratify);
  read_i32(env, a[1], "importance_sampling", &p.importance_sampling);
  read_i32(env, a[1], "stack_size", &p.stack_size);
  read_i32(env, a[1], "frames_in_flight", &p.frames_in_flight);
  bool has = false;
  napi_value v;
  const char* fkeys[3] = {"fov_degrees", "tmin", "light_mix"};
  float* fdst[3] = {&p.fov_degrees, &p.tmin, &p.light_mix};
  for (int k = 0; k < 3; k++) {
    if (napi_has_named_property(env, a[1], fkeys[k], &has) == napi_ok && has && napi_get_named_property(env, a[1], fkeys[k], &v) == napi_ok) {
      double d;
      if (napi_get_value_double(env, v, &d) == napi_ok) *fdst[k] = (float)d;
    }
  }
  if (napi_has_named_property(env, a[1], "background", &has) == napi_ok && has && napi_get_named_property(env, a[1], "background", &v) == napi_ok) {
    for (uint32_t i = 0; i < 3; i++) {
      napi_value e;
      double d;
      if (napi_get_element(env, v, i, &e) == napi_ok && napi_get_value_double(env, e, &d) == napi_ok) p.background[i] = (float)d;
    }
  }
  int st = p_ptmi_set_params(c, &p);
  if (st) return throw_status(env, c, st, "ptmi_set_params");
  return params_to_js(env, &p);
}

static napi_value js_upload(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int32_t which;
  CHECK_NAPI(napi_get_value_int32(env, a[1], &which));
  void* data;
  size_t len;
  if (typed(env, a[2], which == PTMI_BUF_MESHES ? napi_int32_array : napi_float32_array, "upload", &data, &len)) return NULL;
  int st = p_ptmi_upload(c, which, data, len * 4);
  if (st) return throw_status(env, c, st, "ptmi_upload");
  return NULL;
}

static napi_value js_resize(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int32_t w, h;
  CHECK_NAPI(napi_get_value_int32(env, a[1], &w));
  CHECK_NAPI(napi_get_value_int32(env, a[2], &h));
  int st = p_ptmi_resize(c, w, h);
  if (st) return throw_status(env, c, st, "ptmi_resize");
  return NULL;
}

static napi_value js_clear(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_clear_framebuffer(c);
  if (st) return throw_status(env, c, st, "ptmi_clear_framebuffer");
  return NULL;
}

static napi_value js_set_shard(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (get_args(env, info, 4, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int32_t r, w, t;
  CHECK_NAPI(napi_get_value_int32(env, a[1], &r));
  CHECK_NAPI(napi_get_value_int32(env, a[2], &w));
  CHECK_NAPI(napi_get_value_int32(env, a[3], &t));
  int st = p_ptmi_set_shard(c, r, w, t);
  if (st) return throw_status(env, c, st, "ptmi_set_shard");
  return NULL;
}

static napi_value js_render_frame(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  void* data;
  size_t len;
  if (typed(env, a[1], napi_float32_array, "renderFrame(uniforms)", &data, &len)) return NULL;
  if (len != 20) {
    napi_throw_range_error(env, NULL, "renderFrame: uniforms must hold 20 floats [W,H,frameNum,resetBuffer,viewMatrix]");
    return NULL;
  }
  int st = p_ptmi_render_frame(c, (const float*)data);
  if (st) return throw_status(env, c, st, "ptmi_render_frame");
  return NULL;
}

static napi_value js_render(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (get_args(env, info, 4, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  void* data;
  size_t len;
  if (typed(env, a[1], napi_float32_array, "render(view)", &data, &len)) return NULL;
  if (len != 16) {
    napi_throw_range_error(env, NULL, "render: view matrix must hold 16 floats");
    return NULL;
  }
  uint32_t first, n;
  CHECK_NAPI(napi_get_value_uint32(env, a[2], &first));
  CHECK_NAPI(napi_get_value_uint32(env, a[3], &n));
  int st = p_ptmi_render(c, (const float*)data, first, n);
  if (st) return throw_status(env, c, st, "ptmi_render");
  return NULL;
}

static napi_value js_synchronize(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_synchronize(c);
  if (st) return throw_status(env, c, st, "ptmi_synchronize");
  return NULL;
}

static napi_value js_prepare(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_prepare(c);
  if (st) return throw_status(env, c, st, "ptmi_prepare");
  return NULL;
}

/* Scene.create_bvh() on the GPU over the uploaded (unordered) triangles, meshes and transforms: lib/scene.js:253-259 */
static napi_value js_build_scene_bvh(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_build_scene_bvh(c);
  if (st) return throw_status(env, c, st, "ptmi_build_scene_bvh");
  return NULL;
}

/* the same with the reference's other builder, BVH.generate_bvh_heirarchy_SAH (lib/BVH/bvhNode.js:108-283), which its renderer never calls: an opt-in */
static napi_value js_build_scene_bvh_sah(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_build_scene_bvh_sah(c);
  if (st) return throw_status(env, c, st, "ptmi_build_scene_bvh_sah");
  return NULL;
}

static napi_value js_read_fb(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  void* data;
  size_t len;
  if (typed(env, a[1], napi_float32_array, "readFramebuffer(out)", &data, &len)) return NULL;
  int st = p_ptmi_read_framebuffer(c, (float*)data, len * 4);
  if (st) return throw_status(env, c, st, "ptmi_read_framebuffer");
  return a[1];
}

static napi_value js_write_fb(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  void* data;
  size_t len;
  if (typed(env, a[1], napi_float32_array, "writeFramebuffer(src)", &data, &len)) return NULL;
  int st = p_ptmi_write_framebuffer(c, (const float*)data, len * 4);
  if (st) return throw_status(env, c, st, "ptmi_write_framebuffer");
  return NULL;
}

static napi_value js_resolve(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  double fn;
  CHECK_NAPI(napi_get_value_double(env, a[1], &fn));
  void* data;
  size_t len;
  if (typed(env, a[2], napi_uint8_array, "resolveRGBA8(out)", &data, &len)) return NULL;
  int st = p_ptmi_resolve_rgba8(c, (float)fn, (uint8_t*)data, len);
  if (st) return throw_status(env, c, st, "ptmi_resolve_rgba8");
  return a[2];
}

static napi_value js_set_flag(napi_env env, napi_callback_info info, int which) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  bool on = false;
  napi_coerce_to_bool(env, a[1], &a[1]);
  napi_get_value_bool(env, a[1], &on);
  int st = which ? p_ptmi_set_timing(c, on) : p_ptmi_set_counters(c, on);
  if (st) return throw_status(env, c, st, "ptmi_set_counters/timing");
  return NULL;
}
static napi_value js_set_counters(napi_env env, napi_callback_info info) { return js_set_flag(env, info, 0); }
static napi_value js_set_timing(napi_env env, napi_callback_info info) { return js_set_flag(env, info, 1); }

static napi_value js_stats(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  ptmi_stats s;
  int st = p_ptmi_get_stats(c, &s);
  if (st) return throw_status(env, c, st, "ptmi_get_stats");
  napi_value o;
  CHECK_NAPI(napi_create_object(env, &o));
  set_f64(env, o, "rays", (double)s.rays);
  set_f64(env, o, "paths", (double)s.paths);
  set_f64(env, o, "node_visits", (double)s.node_visits);
  set_f64(env, o, "tri_tests", (double)s.tri_tests);
  set_f64(env, o, "sphere_tests", (double)s.sphere_tests);
  set_f64(env, o, "quad_tests", (double)s.quad_tests);
  set_f64(env, o, "mat_fetches", (double)s.mat_fetches);
  set_f64(env, o, "frames", (double)s.frames);
  set_f64(env, o, "intersect_launches", (double)s.intersect_launches);
  set_f64(env, o, "shade_launches", (double)s.shade_launches);
  set_f64(env, o, "render_ms", s.render_ms);
  set_f64(env, o, "intersect_ms", s.intersect_ms);
  set_f64(env, o, "shade_ms", s.shade_ms);
  set_f64(env, o, "other_ms", s.other_ms);
  set_f64(env, o, "prims_ms", s.prims_ms);
  set_f64(env, o, "bvh_ms", s.bvh_ms);
  set_f64(env, o, "generate_ms", s.generate_ms);
  set_f64(env, o, "accumulate_ms", s.accumulate_ms);
  set_f64(env, o, "tail_ms", s.tail_ms);
  set_f64(env, o, "tail_launches", (double)s.tail_launches);
  set_f64(env, o, "devices", (double)s.devices);
  set_f64(env, o, "bvh_node_visits", (double)s.bvh_node_visits);
  set_f64(env, o, "bvh_mat_fetches", (double)s.bvh_mat_fetches);
  set_f64(env, o, "reduce_mode", (double)s.reduce_mode);
  set_f64(env, o, "peer_links", (double)s.peer_links);
  set_f64(env, o, "placement_sets", (double)s.placement_sets);
  set_f64(env, o, "placement_ms", s.placement_ms);
  return o;
}

static napi_value js_reset_stats(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  int st = p_ptmi_reset_stats(c);
  if (st) return throw_status(env, c, st, "ptmi_reset_stats");
  return NULL;
}

/* deviceCount() -> GPUs this process sees; reduceInfo(ctx) -> one line about how a multi-device context sums its devices' buffers (RCCL, add kernel,
 * or "FALLBACK: hipMemcpyPeer + add (<what failed>)" — API v4: an RCCL failure no longer fails the host) */
static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  CHECK_NAPI(napi_create_int32(env, p_ptmi_device_count(), &v));
  return v;
}
static napi_value js_reduce_info(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  ptmi_ctx* c = ctx_of(env, a[0]);
  if (!c) return NULL;
  napi_value v;
  CHECK_NAPI(napi_create_string_utf8(env, p_ptmi_reduce_info(c), NAPI_AUTO_LENGTH, &v));
  return v;
}

/* buildBVH(bmin: Float64Array(3n), bmax: Float64Array(3n), primType) -> { nodes: Float32Array, order: Int32Array } */
static napi_value build_bvh_common(napi_env env, napi_callback_info info, int sah) { /* sah: 0 median (host), 1 SAH (host), 2 median on the GPU of ctx */
  napi_value all[4];
  if (get_args(env, info, sah == 2 ? 4 : 3, all)) return NULL;
  ptmi_ctx* c = NULL;
  if (sah == 2) {
    c = ctx_of(env, all[0]);
    if (!c) return NULL;
  }
  napi_value* a = all + (sah == 2 ? 1 : 0);
  void *bmin, *bmax;
  size_t l0, l1;
  if (typed(env, a[0], napi_float64_array, "buildBVH(bmin)", &bmin, &l0) || typed(env, a[1], napi_float64_array, "buildBVH(bmax)", &bmax, &l1)) return NULL;
  if (l0 != l1 || l0 % 3) {
    napi_throw_range_error(env, NULL, "buildBVH: bmin/bmax must both hold 3*n doubles");
    return NULL;
  }
  int32_t pt = 2;
  CHECK_NAPI(napi_get_value_int32(env, a[2], &pt));
  size_t n = l0 / 3, nn = n ? 2 * n - 1 : 0;
  napi_value ab_nodes, ab_order, nodes, order, out;
  void *pn = NULL, *po = NULL;
  CHECK_NAPI(napi_create_arraybuffer(env, nn * 48, &pn, &ab_nodes));
  CHECK_NAPI(napi_create_arraybuffer(env, n * 4, &po, &ab_order));
  int64_t* tmp = (int64_t*)malloc((n ? n : 1) * sizeof(int64_t));
  if (!tmp) {
    napi_throw_error(env, NULL, "buildBVH: out of memory");
    return NULL;
  }
  size_t rows = nn;
  int st = sah == 1   ? p_ptmi_build_bvh_sah(n, (const double*)bmin, (const double*)bmax, pt, (float*)pn, tmp, &rows)
           : sah == 2 ? p_ptmi_build_bvh_device(c, n, (const double*)bmin, (const double*)bmax, pt, (float*)pn, tmp)
                      : p_ptmi_build_bvh(n, (const double*)bmin, (const double*)bmax, pt, (float*)pn, tmp);
  if (st) {
    free(tmp);
    return throw_status(env, c, st, sah == 1 ? "ptmi_build_bvh_sah" : sah == 2 ? "ptmi_build_bvh_device" : "ptmi_build_bvh");
  }
  for (size_t i = 0; i < n; i++) ((int32_t*)po)[i] = (int32_t)tmp[i];
  free(tmp);
  CHECK_NAPI(napi_create_typedarray(env, napi_float32_array, rows * 12, ab_nodes, 0, &nodes));  /* SAH trees may have fewer rows */
  CHECK_NAPI(napi_create_typedarray(env, napi_int32_array, n, ab_order, 0, &order));
  CHECK_NAPI(napi_create_object(env, &out));
  napi_set_named_property(env, out, "nodes", nodes);
  napi_set_named_property(env, out, "order", order);
  return out;
}

static napi_value js_build_bvh(napi_env env, napi_callback_info info) { return build_bvh_common(env, info, 0); }
/* buildBVHSAH(...): same arguments; the reference's binned-SAH builder (lib/BVH/bvhNode.js:108-283), opt-in */
static napi_value js_build_bvh_sah(napi_env env, napi_callback_info info) { return build_bvh_common(env, info, 1); }
/* buildBVHDevice(ctx, bmin, bmax, primType): the median-split build on the context's GPU, same result as buildBVH */
static napi_value js_build_bvh_device(napi_env env, napi_callback_info info) { return build_bvh_common(env, info, 2); }

/* parseObj(text: string | Uint8Array) -> { vertices: Float32Array, normals: Float32Array }  (objReader.js grammar) */
static napi_value js_parse_obj(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  char* text = NULL;
  size_t len = 0;
  int owned = 0;
  napi_valuetype ty;
  CHECK_NAPI(napi_typeof(env, a[0], &ty));
  if (ty == napi_string) {
    CHECK_NAPI(napi_get_value_string_utf8(env, a[0], NULL, 0, &len));
    text = (char*)malloc(len + 1);
    if (!text) {
      napi_throw_error(env, NULL, "parseObj: out of memory");
      return NULL;
    }
    owned = 1;
    if (napi_get_value_string_utf8(env, a[0], text, len + 1, &len) != napi_ok) {
      free(text);
      napi_throw_error(env, NULL, "parseObj: cannot read string");
      return NULL;
    }
  } else {
    void* data;
    if (typed(env, a[0], napi_uint8_array, "parseObj(text)", &data, &len)) return NULL;
    text = (char*)data;
  }
  float *v = NULL, *n = NULL;
  size_t nv = 0, nn = 0;
  int st = p_ptmi_obj_parse(text, len, &v, &nv, &n, &nn);
  if (owned) free(text);
  if (st) return throw_status(env, NULL, st, "ptmi_obj_parse");
  napi_value abv, abn, tv, tn, out;
  void *pv = NULL, *pn = NULL;
  napi_status s1 = napi_create_arraybuffer(env, nv * 4, &pv, &abv), s2 = napi_create_arraybuffer(env, nn * 4, &pn, &abn);
  if (s1 == napi_ok && s2 == napi_ok) {
    if (nv) memcpy(pv, v, nv * 4);
    if (nn) memcpy(pn, n, nn * 4);
  }
  p_ptmi_free(v);
  p_ptmi_free(n);
  if (s1 != napi_ok || s2 != napi_ok) {
    napi_throw_error(env, NULL, "parseObj: cannot allocate result");
    return NULL;
  }
  CHECK_NAPI(napi_create_typedarray(env, napi_float32_array, nv, abv, 0, &tv));
  CHECK_NAPI(napi_create_typedarray(env, napi_float32_array, nn, abn, 0, &tn));
  CHECK_NAPI(napi_create_object(env, &out));
  napi_set_named_property(env, out, "vertices", tv);
  napi_set_named_property(env, out, "normals", tn);
  return out;
}

static napi_value init(napi_env env, napi_value exports) {
  char err[4400];
  if (load_lib(err, sizeof err)) {
    napi_throw_error(env, NULL, err);
    return NULL;
  }
  static const struct {
    const char* name;
    napi_callback fn;
  } fns[] = {
      {"version", js_version}, {"create", js_create}, {"destroy", js_destroy}, {"defaultParams", js_default_params}, {"setParams", js_set_params},
      {"upload", js_upload}, {"resize", js_resize}, {"clear", js_clear}, {"setShard", js_set_shard}, {"renderFrame", js_render_frame},
      {"render", js_render}, {"synchronize", js_synchronize}, {"prepare", js_prepare}, {"buildSceneBVH", js_build_scene_bvh}, {"buildSceneBVHSAH", js_build_scene_bvh_sah}, {"readFramebuffer", js_read_fb}, {"writeFramebuffer", js_write_fb},
      {"resolveRGBA8", js_resolve}, {"setCounters", js_set_counters}, {"setTiming", js_set_timing}, {"stats", js_stats},
      {"resetStats", js_reset_stats}, {"buildBVH", js_build_bvh}, {"buildBVHSAH", js_build_bvh_sah}, {"buildBVHDevice", js_build_bvh_device}, {"parseObj", js_parse_obj},
      {"deviceCount", js_device_count}, {"reduceInfo", js_reduce_info},
  };
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok || napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
      napi_throw_error(env, NULL, "ptmi.node: cannot register function");
      return NULL;
    }
  }
  napi_value buf;
  napi_create_object(env, &buf);
  set_i32(env, buf, "spheres", PTMI_BUF_SPHERES);
  set_i32(env, buf, "quads", PTMI_BUF_QUADS);
  set_i32(env, buf, "triangles", PTMI_BUF_TRIANGLES);
  set_i32(env, buf, "meshes", PTMI_BUF_MESHES);
  set_i32(env, buf, "transforms", PTMI_BUF_TRANSFORMS);
  set_i32(env, buf, "materials", PTMI_BUF_MATERIALS);
  set_i32(env, buf, "bvh", PTMI_BUF_BVH);
  napi_set_named_property(env, exports, "BUF", buf);
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
