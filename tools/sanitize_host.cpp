// Sanitizer driver for the host natives of libptmi (ptmi_host.cpp): both BVH builders (1 thread vs many) and the OBJ parser.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -pthread tools/sanitize_host.cpp webgpu-path-tracer_amd/csrc/ptmi_host.cpp -o /tmp/san/asan && /tmp/san/asan
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread ... -o /tmp/san/tsan && /tmp/san/tsan
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/ptmi.h"

int main() {
  std::mt19937_64 rng(3);
  std::uniform_real_distribution<double> U(-1, 1), E(0, 0.02);
  for (size_t n : {size_t(1), size_t(2), size_t(777), size_t(90000)}) {
    std::vector<double> lo(3 * n), hi(3 * n);
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        double c = (i % 7 == 0 && k == 0) ? 0.25 : U(rng), e = E(rng);
        lo[3 * i + k] = c - e, hi[3 * i + k] = c + e;
      }
    std::vector<float> a(12 * (2 * n - 1)), b(a.size()), s(a.size());
    std::vector<int64_t> oa(n), ob(n), os(n);
    setenv("PTMI_BUILD_THREADS", "1", 1);
    if (ptmi_build_bvh(n, lo.data(), hi.data(), 2, a.data(), oa.data())) return 1;
    setenv("PTMI_BUILD_THREADS", "8", 1);
    if (ptmi_build_bvh(n, lo.data(), hi.data(), 2, b.data(), ob.data())) return 1;
    if (memcmp(a.data(), b.data(), a.size() * 4) || oa != ob) {
      printf("thread-count dependence at n=%zu\n", n);
      return 2;
    }
    size_t rows = 0, rows1 = 0;
    std::vector<float> s1(a.size());
    std::vector<int64_t> os1(n);
    if (ptmi_build_bvh_sah(n, lo.data(), hi.data(), 2, s.data(), os.data(), &rows) || rows == 0 || rows > 2 * n - 1) return 3;  // 8 threads: subtrees fork where a thread is free
    setenv("PTMI_BUILD_THREADS", "1", 1);
    if (ptmi_build_bvh_sah(n, lo.data(), hi.data(), 2, s1.data(), os1.data(), &rows1) || rows1 != rows || memcmp(s.data(), s1.data(), rows * 48) || os != os1) {
      printf("SAH: thread-count dependence at n=%zu\n", n);
      return 5;
    }
  }
  std::string obj = "# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf 1//1 9//1 3//7\nv 1e3 0x10 -Infinity\nf 4 4 4\nv\n";
  float *v = nullptr, *nn = nullptr;
  size_t nv = 0, nnn = 0;
  if (ptmi_obj_parse(obj.data(), obj.size(), &v, &nv, &nn, &nnn)) return 4;
  ptmi_free(v);
  ptmi_free(nn);
  puts("host natives: sanitizer run clean");
  return 0;
}
