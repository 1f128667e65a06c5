// isa_pieces.hip — the building blocks of k_generate / k_shade as stand-alone kernels, so that tools/isa_histogram.py --pieces can count what each one costs
// (static instruction counts of mostly straight-line code; device only, never linked into the product).
#include "../webgpu-path-tracer_amd/csrc/ptmi_kernels.h"
using namespace ptmi;
extern "C" {
__global__ void p_norm3(float4* io) { f3 v = mk3(io[threadIdx.x]); v = norm3(v); io[threadIdx.x] = make_float4(v.x, v.y, v.z, 0); }
__global__ void p_div3(float4* io) { float4 a = io[threadIdx.x]; f3 v = mk3(a) / a.w; io[threadIdx.x] = make_float4(v.x, v.y, v.z, 0); }
__global__ void p_sqrt_ieee(float* io) { io[threadIdx.x] = ptm_sqrt(io[threadIdx.x]); }
__global__ void p_sqrt_exact(float* io) { io[threadIdx.x] = sqrt_exact(io[threadIdx.x]); }
__global__ void p_rcp_exact(float* io) { io[threadIdx.x] = rcp_exact_il(io[threadIdx.x]); }
__global__ void p_rcp3_exact(float4* io) { f3 v = rcp3_exact_il(mk3(io[threadIdx.x])); io[threadIdx.x] = make_float4(v.x, v.y, v.z, 0); }
__global__ void p_div_ieee(float* io) { io[threadIdx.x] = io[threadIdx.x + 64] / io[threadIdx.x]; }
__global__ void p_rcp_ieee(float* io) { io[threadIdx.x] = 1.0f / io[threadIdx.x]; }
__global__ void p_rand(uint32_t* s, float* o) { uint32_t r = s[threadIdx.x]; o[threadIdx.x] = rand2D(r); s[threadIdx.x] = r; }
__global__ void p_sincos(float* io) { float p = io[threadIdx.x]; io[threadIdx.x] = ptm_sin(p); io[threadIdx.x + 64] = ptm_cos(p); }
__global__ void p_cosine_sample(uint32_t* s, float4* o) { uint32_t r = s[threadIdx.x]; f3 v = cosine_sampling_wrt_Z(r); o[threadIdx.x] = make_float4(v.x, v.y, v.z, 0); s[threadIdx.x] = r; }
__global__ void p_onb(float4* io) { Onb b = onb_build_from_w(mk3(io[threadIdx.x])); io[threadIdx.x] = make_float4(b.u.x, b.u.y, b.u.z, b.v.x); io[threadIdx.x + 64] = make_float4(b.v.y, b.v.z, b.w.x, b.w.y); }
__global__ void p_scatter_lambert(DevScene S, float4* io, uint32_t* s) {
  uint32_t r = s[threadIdx.x];
  Material m = load_material(S, 3);
  HitGeom g; g.p = mk3(io[threadIdx.x]); g.n = mk3(io[threadIdx.x + 64]); g.front = true;
  float ds; bool sk; f3 uw;
  f3 d = material_scatter(BIN_LAMBERTIAN, m, g, mk3(io[threadIdx.x + 128]), r, ds, sk, uw);
  io[threadIdx.x] = make_float4(d.x, d.y, d.z, ds); s[threadIdx.x] = r;
}
__global__ void p_prims(DevScene S, float4* io, uint32_t* s, float2* tp, uint32_t* hm) {
  uint32_t r = s[threadIdx.x]; Counters cn = {0,0,0,0,0};
  float2 t; uint32_t h;
  prims_for_ray<false>(S, mk3(io[threadIdx.x]), mk3(io[threadIdx.x + 64]), r, t, h, cn);
  tp[threadIdx.x] = t; hm[threadIdx.x] = h; s[threadIdx.x] = r;
}
__global__ void p_resolve_quad(DevScene S, float4* io) {
  HitGeom g = resolve_hit(S, mk3(io[threadIdx.x]), mk3(io[threadIdx.x + 64]), io[threadIdx.x].w, 0, 0, (K_QUAD << 28) | 2u);
  io[threadIdx.x] = make_float4(g.p.x, g.p.y, g.p.z, g.front); io[threadIdx.x + 64] = make_float4(g.n.x, g.n.y, g.n.z, 0);
}
__global__ void p_resolve_tri(DevScene S, float4* io) {
  HitGeom g = resolve_hit(S, mk3(io[threadIdx.x]), mk3(io[threadIdx.x + 64]), io[threadIdx.x].w, 0.3f, 0.2f, (K_TRI << 28) | (threadIdx.x));
  io[threadIdx.x] = make_float4(g.p.x, g.p.y, g.p.z, g.front); io[threadIdx.x + 64] = make_float4(g.n.x, g.n.y, g.n.z, 0);
}
__global__ void p_camera(RenderConst rc, float4* o) { uint32_t r = threadIdx.x * 7u; f3 a, b; camera_ray(rc, threadIdx.x, 0, r, a, b); o[threadIdx.x] = make_float4(a.x, a.y, a.z, b.x); o[threadIdx.x + 64] = make_float4(b.y, b.z, __uint_as_float(r), 0); }
}
