// valu_peak.hip — measured wave64 vector-instruction issue rates on MI355X (gfx950), per instruction class and per
// number of resident waves per SIMD.  Settles the denominator of the `valu_issue` roofline in bench.py / DESIGN.md §5:
// MI355X_MICROARCH.md ("Wave scheduling", cycle-constants row `v_fma_f32`) says a wave64 VALU instruction issues over
// 2 cycles on CDNA4's SIMD-32 when >= 2 waves share the SIMD and over 4 for a lone wave.
//
// Every wave runs kIter iterations of 64 instructions of one opcode — 8 independent chains x 8, inline asm so that nothing is
// folded, one taken branch per 64 instructions.  Placement is forced, not hoped for: a case with W waves per SIMD launches
// one block of 256*W threads per CU (two blocks of 256*W/2 for W > 4) and every block asks for so much LDS that exactly that
// many fit a CU; each wave records HW_ID / XCC_ID and the host counts the waves that really shared each SIMD.
// Reported per case: cycles per wave-instruction per SIMD from inside the kernel (median wave's s_memtime span over its
// instruction count, divided by the waves on its SIMD), the same from the wall clock (HIP events around 20 launches,
// launch gaps and tails included), and the shader clock (s_memtime / s_memrealtime).
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_peak tools/valu_peak.hip     run: tools/valu_peak > profiles/valu_peak.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

constexpr int kIter = 2048, kPerIter = 64;

enum Op { FMA32, MUL32, MIN32, CNDMASK, MOV, PKFMA32, PKMUL32, SQRT32, RCP32, RSQ32, FMA64, MUL64, ADD64, RCP64, CVT_F64_F32, CVT_F32_F64, CMP_CLASS32, CMP_GT32, DIVSCALE32, DIVFMAS32, DIVFIXUP32, MAD_U32, MUL_LO_U32, ADD_U32, AND_B32, LSHLREV, XOR_B32, ADD_F32, CVT_F32_U32, CNDMASK_SGPR, MAX3, FMA32_HALF, MIN32_HALF, N_OPS };
const char* kNames[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_min_f32", "v_cndmask_b32", "v_mov_b32", "v_pk_fma_f32", "v_pk_mul_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_fma_f64",
                             "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cmp_class_f32", "v_cmp_gt_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
                             "v_mad_u32_u24", "v_mul_lo_u32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_xor_b32", "v_add_f32", "v_cvt_f32_u32", "v_cndmask_b32 (sgpr-pair mask)", "v_max3_f32", "v_fma_f32, EXEC = lanes 0-31 only", "v_min_f32, EXEC = lanes 0-31 only"};

struct WaveRec {
  unsigned long long cycles, realtime;
  uint32_t hw_id, xcc_id;
};

template <int OP>
__global__ __launch_bounds__(1024) void k_issue(float* __restrict__ sink, WaveRec* __restrict__ rec) {
  extern __shared__ int pad_lds[];  // only its size matters: it limits the blocks per CU
  float a[8];
  double d[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
  unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float b = 1.0000001f, c = 1e-9f;
  const double db = 1.0000000001, dc = 1e-12;
  for (int i = 0; i < 8; i++) {
    a[i] = 1.0f + (float)(threadIdx.x + i) * 1e-3f;
    d[i] = 1.0 + (double)(threadIdx.x + i) * 1e-3;
    p[i] = v2f{a[i], a[i] + 1.0f};
  }
  const unsigned long long lane_mask = __ballot(threadIdx.x & 1);  // a real per-lane mask in an SGPR pair
  __syncthreads();  // all waves of the block start together
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // partial-EXEC cases: does the SIMD skip the 32-lane pass whose lanes are all masked off?
  const int live = (OP == FMA32_HALF || OP == MIN32_HALF) ? 32 : 64;
  if ((int)(threadIdx.x & 63) < live)
#pragma unroll 1
  for (int it = 0; it < kIter; it++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == FMA32 || OP == FMA32_HALF) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MIN32 || OP == MIN32_HALF) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));  // reads vcc only (a clobber would make the compiler pad every one with s_nop)
        if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
        if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (OP == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        if (OP == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
        if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
        if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
        if (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        // (masks go to SGPR pairs, not vcc: after an asm that clobbers vcc the compiler pads with s_nop)
        if (OP == CMP_CLASS32) asm volatile("v_cmp_class_f32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(0x267));
        if (OP == CMP_GT32) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b));
        if (OP == DIVSCALE32) asm volatile("v_div_scale_f32 %0, %1, %0, %2, %0" : "+v"(a[i]), "=s"(m[i]) : "v"(b));
        if (OP == DIVFMAS32) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));  // reads vcc
        if (OP == DIVFIXUP32) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MAD_U32) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
        if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
        if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(lane_mask));
        if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  for (int i = 0; i < 8; i++) s += a[i] + (float)d[i] + p[i].x + p[i].y + (float)m[i];
  if (s == 12345.678f) sink[0] = s + (float)pad_lds[0];  // keeps the chains (and the LDS allocation) alive
  if ((threadIdx.x & 63) == 0) {
    WaveRec w;
    w.cycles = t1 - t0;
    w.realtime = r1 - r0;
    w.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
    w.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    rec[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;
  }
}

template <int OP>
int run_case(int n_cu, int wps, float* sink, WaveRec* rec, std::vector<WaveRec>& h, std::string& json) {
  const int blocks_per_cu = wps > 4 ? 2 : 1;
  const int threads = 256 * wps / blocks_per_cu;  // waves of a block are dealt to the four SIMDs in turn
  const int grid = n_cu * blocks_per_cu;
  const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : 60 * 1024;  // 160 KB per CU: one block of 100 KB fits, or two of 60 KB, never more
  CK(hipFuncSetAttribute((const void*)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float warm_ms = 0.0f;  // back-to-back launches first, so that the clock has settled under this load
  CK(hipEventRecord(e0));
  while (warm_ms < 300.0f) {
    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(threads), lds, 0, sink, rec);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&warm_ms, e0, e1));
  }
  const int n = 20;
  CK(hipEventRecord(e0));
  for (int k = 0; k < n; k++) hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(threads), lds, 0, sink, rec);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const int n_waves = grid * threads / 64;
  CK(hipMemcpy(h.data(), rec, (size_t)n_waves * sizeof(WaveRec), hipMemcpyDeviceToHost));
  std::vector<double> ghz, cyc;
  std::map<uint64_t, int> per_simd;  // (xcc, se, sh, cu, simd) -> waves of the last launch
  for (int w = 0; w < n_waves; w++) {
    if (h[w].realtime) ghz.push_back((double)h[w].cycles / (double)h[w].realtime * 0.1);  // s_memrealtime ticks at 100 MHz
    cyc.push_back((double)h[w].cycles);
    const uint32_t id = h[w].hw_id;
    per_simd[((uint64_t)(h[w].xcc_id & 0xf) << 32) | (id & 0xfff0u)]++;  // everything but the wave slot
  }
  std::sort(ghz.begin(), ghz.end());
  std::sort(cyc.begin(), cyc.end());
  int lo = 1 << 30, hi = 0;
  for (auto& kv : per_simd) lo = std::min(lo, kv.second), hi = std::max(hi, kv.second);
  const double clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  const double per_wave = (double)kIter * kPerIter;
  const double rate = (double)n_waves * per_wave * n / (ms * 1e-3);  // chip-wide, wall clock
  const double simds = (double)n_cu * 4.0;
  const double cyc_per_instr_wave = cyc[cyc.size() / 2] / per_wave;  // what ONE wave sustains with wps - 1 neighbours
  const double cyc_per_instr_simd = cyc_per_instr_wave / wps;
  char buf[640];
  snprintf(buf, sizeof buf,
           "   {\"op\": \"%s\", \"waves_per_simd\": %d, \"simds_occupied\": %zu, \"waves_per_simd_seen\": [%d, %d], \"clock_ghz\": %.3f, "
           "\"cycles_per_wave_instr_per_simd\": %.3f, \"cycles_between_issues_of_one_wave\": %.2f, \"cycles_per_wave_instr_per_simd_wall\": %.3f, "
           "\"chip_wave_instr_per_s\": %.4e, \"chip_wave_instr_per_s_wall\": %.4e}",
           kNames[OP], wps, per_simd.size(), lo, hi, clock_ghz, cyc_per_instr_simd, cyc_per_instr_wave, simds * clock_ghz * 1e9 / rate, simds * clock_ghz * 1e9 / cyc_per_instr_simd, rate);
  json += buf;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

template <int OP>
int run_op(int n_cu, float* sink, WaveRec* rec, std::vector<WaveRec>& h, std::string& json, bool all_occupancies) {
  const int full[] = {1, 2, 3, 4, 6, 8}, few[] = {1, 4, 8};
  const int* w = all_occupancies ? full : few;
  const int nw = all_occupancies ? 6 : 3;
  for (int i = 0; i < nw; i++) {
    if (!json.empty()) json += ",\n";
    if (run_case<OP>(n_cu, w[i], sink, rec, h, json)) return 1;
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* sink;
  WaveRec* rec;
  const size_t waves = (size_t)n_cu * 32;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&rec, waves * sizeof(WaveRec)));
  std::vector<WaveRec> h(waves);
  std::string json;
  int rc = 0;
  rc |= run_op<FMA32>(n_cu, sink, rec, h, json, true);
  rc |= run_op<MUL32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MIN32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MOV>(n_cu, sink, rec, h, json, false);
  // (v_cndmask_b32 is not listed: in this loop it reads a VCC that nothing ever writes and times at 22 cycles, which the real kernels
  //  contradict — k_bvh is 40 % v_cndmask/v_mov and its SQ_ACTIVE_INST_VALU is 1.00 quad-cycles per instruction)
  rc |= run_op<CMP_GT32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<CMP_CLASS32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MAD_U32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MUL_LO_U32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<ADD_U32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<AND_B32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<LSHLREV>(n_cu, sink, rec, h, json, false);
  rc |= run_op<XOR_B32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<ADD_F32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<CVT_F32_U32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<CNDMASK_SGPR>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MAX3>(n_cu, sink, rec, h, json, false);
  rc |= run_op<FMA32_HALF>(n_cu, sink, rec, h, json, false);
  rc |= run_op<MIN32_HALF>(n_cu, sink, rec, h, json, false);
  rc |= run_op<PKFMA32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<PKMUL32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<SQRT32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<RCP32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<RSQ32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<DIVSCALE32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<DIVFMAS32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<DIVFIXUP32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<FMA64>(n_cu, sink, rec, h, json, true);
  rc |= run_op<MUL64>(n_cu, sink, rec, h, json, false);
  rc |= run_op<ADD64>(n_cu, sink, rec, h, json, false);
  rc |= run_op<RCP64>(n_cu, sink, rec, h, json, false);
  rc |= run_op<CVT_F64_F32>(n_cu, sink, rec, h, json, false);
  rc |= run_op<CVT_F32_F64>(n_cu, sink, rec, h, json, false);
  if (rc) return 1;
  printf("{\n \"device\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"simds\": %d,\n", prop.name, prop.gcnArchName, n_cu, n_cu * 4);
  printf(" \"method\": \"per wave: %d iterations x %d instructions of one opcode (8 independent chains x 8, inline asm); W waves per SIMD = one block of 256*W threads "
         "per CU (two of 128*W for W > 4), blocks per CU pinned by their LDS size, placement verified from HW_ID/XCC_ID (simds_occupied, waves_per_simd_seen = "
         "[min, max]); clock = s_memtime / s_memrealtime (100 MHz), median over waves, after >= 0.3 s of back-to-back launches; cycles_per_wave_instr_per_simd = "
         "median wave span / instructions per wave / W; *_wall = the same from HIP events around 20 launches (launch gaps and tails included) and are the figures to use as throughput; the compiler pads inline asm that writes an SGPR pair (v_cmp*, v_div_scale) with one s_nop 0 each and every loop body with a few, so those rows are upper bounds\",\n \"cases\": [\n%s\n ]\n}\n",
         kIter, kPerIter, json.c_str());
  return 0;
}
