// fetch_calib.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the two access patterns of the integrator:
//   stream   every lane reads consecutive float4 (k_generate / k_shade / k_accumulate state streams)
//   gather   every lane reads ONE 64-byte record (4 x float4, 64-byte aligned) at a random index of a table far larger than
//            L2 + Infinity Cache (k_bvh's pair / triangle records)
// The program prints the bytes each kernel really asked for; run it under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE)
// and tools/fetch_calib.sh divides.  MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// build: hipcc --offload-arch=gfx950 -O2 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ src, size_t n, float* __restrict__ sink) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.0f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather64(const float4* __restrict__ table, uint32_t n_records, uint32_t per_lane, float* __restrict__ sink) {
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.0f;
  for (uint32_t k = 0; k < per_lane; k++) {
    s = s * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    const uint32_t r = ((w >> 22u) ^ w) % n_records;
    const float4* rec = table + 4 * (size_t)r;
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 12345.0f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_store(float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_float4(1.0f, 2.0f, 3.0f, (float)i);
}

int main() {
  const size_t table_bytes = (size_t)8 << 30;  // 8 GiB: 32x the Infinity Cache
  float4* table;
  float* sink;
  CK(hipMalloc(&table, table_bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, table_bytes));
  const size_t n4 = table_bytes / 16;
  const uint32_t n_records = (uint32_t)(table_bytes / 64);
  const uint32_t grid = 256 * 16, per_lane = 64;
  hipLaunchKernelGGL(calib_stream, dim3(grid), dim3(256), 0, 0, table, n4, sink);
  hipLaunchKernelGGL(calib_gather64, dim3(grid), dim3(256), 0, 0, table, n_records, per_lane, sink);
  hipLaunchKernelGGL(calib_store, dim3(grid), dim3(256), 0, 0, table, n4);
  CK(hipDeviceSynchronize());
  printf("{\"calib_stream\": {\"read_bytes\": %zu}, \"calib_gather64\": {\"read_bytes\": %zu, \"records\": %zu}, \"calib_store\": {\"write_bytes\": %zu}}\n",
         table_bytes, (size_t)grid * 256 * per_lane * 64, (size_t)grid * 256 * per_lane, table_bytes);
  return 0;
}
