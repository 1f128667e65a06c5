// gather_probe2.hip — what does a wave pay for a 64-byte record gather when only some of its lanes take part, and for other widths?
// (k_bvh's inner phases run with ~47 of 64 lanes, its triangle phases with ~18.)  Per-lane 4 x global_load_dwordx4 at a random record of an
// L1-resident (64 KB) or L2-resident (2 MB) table, dependent chain per lane, 5 waves per SIMD; `active` lanes picked scattered (lane % k == 0)
// or contiguous (lane < n).  Also: the same 64 bytes as 2 records of 32 bytes (2 x dwordx4), and the record held in LDS (4 x ds_read_b128).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe2 tools/gather_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
// WIDTH: float4 loads per record (4 = 64 B, 2 = 32 B, 1 = 16 B)
template <int WIDTH>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ table, uint32_t mask, uint32_t steps, int stride, int limit, float* __restrict__ sink, int share = 1) {
  const int lane = threadIdx.x & 63;
  const bool on = (lane % stride == 0) && lane < limit;
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t r = pcg(s) & mask;
  float acc = 0.0f;
  if (on) {
    for (uint32_t k = 0; k < steps; k++) {
      const uint32_t rr = share > 1 ? (uint32_t)__shfl((int)r, lane & ~(share - 1), 64) : r;  // `share` neighbouring lanes fetch the same record
      const float4* rec = table + 4 * (size_t)rr;
      float4 a = rec[0], b = a, c = a, d = a;
      if (WIDTH >= 2) b = rec[1];
      if (WIDTH >= 4) c = rec[2], d = rec[3];
      acc += a.x + b.y + c.z + d.w;
      r = (pcg(s) ^ __float_as_uint(a.x) ^ __float_as_uint(d.w)) & mask;
    }
  }
  if (acc == 12345.0f) sink[0] = acc;
}
// the table in LDS (48 KB per block of 256 threads: 768 records), 4 x ds_read_b128 per lane
__global__ __launch_bounds__(256) void probe_lds(const float4* __restrict__ table, uint32_t steps, int stride, int limit, float* __restrict__ sink) {
  __shared__ float4 t[4 * 512];
  for (int i = threadIdx.x; i < 4 * 512; i += 256) t[i] = table[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool on = (lane % stride == 0) && lane < limit;
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t r = pcg(s) & 511u;
  float acc = 0.0f;
  if (on) {
    for (uint32_t k = 0; k < steps; k++) {
      const float4* rec = t + 4 * r;
      const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
      acc += a.x + b.y + c.z + d.w;
      r = (pcg(s) ^ __float_as_uint(a.x) ^ __float_as_uint(d.w)) & 511u;
    }
  }
  if (acc == 12345.0f) sink[0] = acc;
}
// one dwordx4 per lane; groups of G lanes read consecutive pieces of a random (G*16)-byte-aligned block
__global__ __launch_bounds__(256) void probe_contig(const float4* __restrict__ table, uint32_t mask16, uint32_t steps, int G, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t r = pcg(s);
  float acc = 0.0f;
  for (uint32_t k = 0; k < steps; k++) {
    const uint32_t rr = (uint32_t)__shfl((int)r, lane & ~(G - 1), 64);     // the group's block
    const uint32_t idx = ((rr * (uint32_t)G) + (uint32_t)(lane & (G - 1))) & mask16;
    const float4 a = table[idx];
    acc += a.x + a.w;
    r = pcg(s) ^ __float_as_uint(a.x);
  }
  if (acc == 12345.0f) sink[0] = acc;
}

int main() {
  float4* table;
  float* sink;
  CK(hipMalloc(&table, (size_t)2 << 20));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, (size_t)2 << 20));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Pat { const char* name; int stride, limit, lanes; };
  const Pat pats[] = {{"64 lanes", 1, 64, 64}, {"48 contiguous", 1, 48, 48}, {"32 scattered", 2, 64, 32}, {"32 contiguous", 1, 32, 32}, {"16 scattered", 4, 64, 16}, {"16 contiguous", 1, 16, 16}, {"8 scattered", 8, 64, 8}};
  const uint32_t steps = 2000;
  const dim3 grid((unsigned)(cus * 5)), block(256);
  printf("{\"cases\": [\n");
  bool first = true;
  for (int width : {4, 2, 1, 0}) {
    for (size_t bytes : {(size_t)64 << 10, (size_t)2 << 20}) {
      if (width == 0 && bytes != ((size_t)64 << 10)) continue;
      for (const Pat& p : pats) {
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0, 0));
          const uint32_t mask = (uint32_t)(bytes / 64 - 1);
          if (width == 4) hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, table, mask, steps, p.stride, p.limit, sink);
          else if (width == 2) hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, table, mask, steps, p.stride, p.limit, sink);
          else if (width == 1) hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, table, mask, steps, p.stride, p.limit, sink);
          else hipLaunchKernelGGL(probe_lds, grid, block, 0, 0, table, steps, p.stride, p.limit, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)cus * 5 * 4 * steps;  // wave-level fetches
        const double clk_per_wave_fetch = ms * 1e-3 * (prop.clockRate * 1e3) / (winstr / cus);
        printf("%s {\"what\": \"%s\", \"bytes_per_lane\": %d, \"table_kb\": %zu, \"lanes\": \"%s\", \"ms\": %.3f, \"clk_per_wave_fetch_per_cu\": %.1f, \"clk_per_record_per_cu\": %.2f}", first ? "" : ",\n",
               width ? "global" : "lds", (width ? width : 4) * 16, bytes >> 10, p.name, ms, clk_per_wave_fetch, clk_per_wave_fetch / p.lanes);
        first = false;
        fflush(stdout);
      }
    }
  }
  // lanes sharing records: groups of 4 / 16 / 64 neighbouring lanes fetch the SAME 64-byte record (what coherent rays do at the top of a tree)
  for (int share : {1, 4, 16, 64}) {
    for (size_t bytes : {(size_t)64 << 10, (size_t)2 << 20}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, table, (uint32_t)(bytes / 64 - 1), steps, 1, 64, sink, share);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
      }
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double winstr = (double)cus * 5 * 4 * steps;
      printf(",\n {\"what\": \"global, %d lanes per record\", \"bytes_per_lane\": 64, \"table_kb\": %zu, \"lanes\": \"64 lanes\", \"ms\": %.3f, \"clk_per_wave_fetch_per_cu\": %.1f, \"clk_per_record_per_cu\": %.2f}",
             share, bytes >> 10, ms, ms * 1e-3 * (prop.clockRate * 1e3) / (winstr / cus), ms * 1e-3 * (prop.clockRate * 1e3) / (winstr / cus) / 64);
    }
  }
  // contiguity: groups of G neighbouring lanes read G consecutive 16-byte pieces (one dwordx4 per lane) of a random G*16-byte block
  for (int G : {1, 4, 8, 16, 64}) {
    for (size_t bytes : {(size_t)64 << 10, (size_t)2 << 20}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe_contig, grid, block, 0, 0, table, (uint32_t)(bytes / 16 - 1), steps, G, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
      }
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double winstr = (double)cus * 5 * 4 * steps;  // one load per step here
      printf(",\n {\"what\": \"global, 1 x dwordx4 per lane, %d lanes contiguous\", \"bytes_per_lane\": 16, \"table_kb\": %zu, \"lanes\": \"64 lanes\", \"ms\": %.3f, \"clk_per_wave_fetch_per_cu\": %.1f, \"clk_per_record_per_cu\": %.2f}",
             G, bytes >> 10, ms, ms * 1e-3 * (prop.clockRate * 1e3) / (winstr / cus), ms * 1e-3 * (prop.clockRate * 1e3) / (winstr / cus) / 64);
    }
  }
  printf("\n]}\n");
  return 0;
}
