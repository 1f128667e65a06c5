// launch_gap_probe — what does a kernel boundary cost on one stream?  N back-to-back launches of a kernel that does nothing (and of
// one that touches one cache line), timed with events; the same sequence captured in a hipGraph and replayed.
// hipcc --offload-arch=gfx950 -O2 -o launch_gap_probe tools/launch_gap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_nothing(int* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void k_touch(int* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  int* d;
  CK(hipMalloc(&d, 4096));
  CK(hipMemset(d, 0, 4096));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int N = 2000;
  printf("{");
  for (int which = 0; which < 2; which++) {
    for (int grid : {1, 4096}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; i++) {
          if (which) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(64), 0, s, d);
          else hipLaunchKernelGGL(k_nothing, dim3(grid), dim3(64), 0, s, d);
        }
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("\"stream_%s_grid%d_us_per_launch\": %.2f, ", which ? "touch" : "nothing", grid, ms * 1e3 / N);
      }
    }
  }
  // the same as a graph: 26 kernels per graph (one render step of configs[1]), replayed
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 26; i++) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(64), 0, s, d);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 100; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep) printf("\"graph_26_touch_grid4096_us_per_kernel\": %.2f", ms * 1e3 / (100 * 26));
  }
  printf("}\n");
  return 0;
}
