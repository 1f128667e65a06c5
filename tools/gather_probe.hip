// gather_probe.hip — how fast can a CU pull 64-byte records from random addresses (k_bvh's access pattern), and does it depend on
// HOW the 64 bytes are asked for?
//   lane      every lane reads its own record with 4 x global_load_dwordx4 (what k_bvh does): each wave-instruction touches 64
//             different cache lines with 16 bytes each
//   quad      the four lanes of a quad read ONE record per instruction (lane q of the quad reads bytes 16q..16q+15), four
//             instructions for the quad's four records: each wave-instruction touches 16 lines with 64 contiguous bytes each
//   quad_lds  `quad`, then a 4x4 transpose through LDS (4 x ds_write_b128, 4 x ds_read_b128) so that every lane ends up with
//             its own record in registers — the full price of the cooperative form
// Every lane follows a dependent chain (the next index depends on the loaded data, as a traversal's does); tables of 8 KB and 64 KB
// (L1-resident), 512 KB and 2 MB (L2-resident per XCD), 32 MB, 128 MB (Infinity Cache) and 2 GB; 5 single-wave-per-SIMD-slot blocks of 256 threads per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ table, uint32_t mask, uint32_t steps, float* __restrict__ sink) {
  __shared__ float4 tr[MODE == 2 ? 4 * 256 : 1];
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t r = pcg(s) & mask;
  float acc = 0.0f;
  const int lane = threadIdx.x & 63, q = lane & 3, qbase = lane & ~3;
  for (uint32_t k = 0; k < steps; k++) {
    float4 a, b, c, d;
    if (MODE == 0) {
      const float4* rec = table + 4 * (size_t)r;
      a = rec[0], b = rec[1], c = rec[2], d = rec[3];
    } else {
      // record index of quad member m, broadcast within the quad
      const uint32_t r0 = (uint32_t)__shfl((int)r, qbase + 0, 64), r1 = (uint32_t)__shfl((int)r, qbase + 1, 64);
      const uint32_t r2 = (uint32_t)__shfl((int)r, qbase + 2, 64), r3 = (uint32_t)__shfl((int)r, qbase + 3, 64);
      a = table[4 * (size_t)r0 + q];  // piece q of member 0's record
      b = table[4 * (size_t)r1 + q];
      c = table[4 * (size_t)r2 + q];
      d = table[4 * (size_t)r3 + q];
      if (MODE == 2) {
        // transpose through LDS: lane (quad member m) gets pieces 0..3 of its own record
        float4* w = tr + 4 * (threadIdx.x & ~3);  // the quad's 16 float4: [member][piece]
        w[0 * 4 + q] = a, w[1 * 4 + q] = b, w[2 * 4 + q] = c, w[3 * 4 + q] = d;
        __builtin_amdgcn_wave_barrier();
        const float4* rd = w + 4 * q;
        a = rd[0], b = rd[1], c = rd[2], d = rd[3];
        __builtin_amdgcn_wave_barrier();
      }
    }
    acc += a.x + b.y + c.z + d.w;
    r = (pcg(s) ^ __float_as_uint(a.x) ^ __float_as_uint(d.w)) & mask;  // dependent chain (the table holds zeros)
  }
  if (acc == 12345.0f) sink[0] = acc;
}

int main() {
  const size_t max_bytes = (size_t)2 << 30;
  float4* table;
  float* sink;
  CK(hipMalloc(&table, max_bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, max_bytes));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t sizes[] = {(size_t)8 << 10, (size_t)64 << 10, (size_t)512 << 10, (size_t)2 << 20, (size_t)32 << 20, (size_t)128 << 20, (size_t)2 << 30};
  const char* names[] = {"lane", "quad", "quad_lds"};
  printf("{\"cus\": %d, \"clock_mhz\": %d, \"cases\": [\n", cus, prop.clockRate / 1000);
  bool first = true;
  for (int bpc : {5, 8}) {
    for (size_t bytes : sizes) {
      const uint32_t mask = (uint32_t)(bytes / 64 - 1);
      for (int mode = 0; mode < 3; mode++) {
        const uint32_t steps = 2000;
        const dim3 grid((unsigned)(cus * bpc)), block(256);
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0, 0));
          if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, table, mask, steps, sink);
          else if (mode == 1) hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, table, mask, steps, sink);
          else hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, table, mask, steps, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double recs = (double)cus * bpc * 256 * steps;
        printf("%s {\"mode\": \"%s\", \"table_kb\": %zu, \"waves_per_simd\": %d, \"ms\": %.3f, \"grecords_per_s\": %.2f, \"tb_per_s\": %.2f, \"cycles_per_record_per_cu\": %.2f}",
               first ? "" : ",\n", names[mode], bytes >> 10, bpc, ms, recs / ms / 1e6, recs * 64 / ms / 1e9, ms * 1e-3 * (prop.clockRate * 1e3) / (recs / cus));
        first = false;
        fflush(stdout);
      }
    }
  }
  printf("\n]}\n");
  return 0;
}
