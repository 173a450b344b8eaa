// Vector-load pointer chase: one wave, L lanes active (1, 8, 64), every lane follows its OWN random chain
// (fully divergent: L distinct 128-byte lines per load instruction).  Cycles per dependent hop, first pass (data
// written by a copy: Infinity Cache / HBM) and second pass (own L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const unsigned* __restrict__ buf, int hops, const unsigned* __restrict__ starts, int lanes, unsigned long long* out) {
  if ((int)threadIdx.x >= lanes) return;
  unsigned i = starts[threadIdx.x];
  const unsigned long long t0 = clock64();
  for (int k = 0; k < hops; ++k) i = buf[(size_t)i * 32];
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  out[1 + threadIdx.x] = i;
}
int main() {
  unsigned long long* out; hipMalloc(&out, 8 * 80);
  std::mt19937 rng(1);
  for (size_t bytes : {size_t(1) << 20, size_t(64) << 20, size_t(2) << 30}) {
    const size_t n = bytes / 128;
    std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<unsigned> host(n * 32, 0u);
    for (size_t k = 0; k < n; ++k) host[(size_t)perm[k] * 32] = perm[(k + 1) % n];
    unsigned* dev; hipMalloc(&dev, bytes);
    std::vector<unsigned> st(64); for (int l = 0; l < 64; ++l) st[l] = perm[(n / 64) * l];
    unsigned* dst; hipMalloc(&dst, 256); hipMemcpy(dst, st.data(), 256, hipMemcpyHostToDevice);
    for (int lanes : {1, 8, 64}) {
      const int hops = (int)std::min<size_t>(n / 64, 2000);
      hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice);   // rewrite: lines leave the L2s
      unsigned long long h[2];
      chase<<<1, 64>>>(dev, hops, dst, lanes, out); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double cold = (double)h[0] / hops;
      chase<<<1, 64>>>(dev, hops, dst, lanes, out); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%8.1f MiB  lanes %2d  hops %5d  first pass %7.1f clk/hop   second pass %7.1f clk/hop\n", bytes / 1048576.0, lanes, hops, cold, (double)h[0] / hops);
    }
    hipFree(dev); hipFree(dst);
  }
  return 0;
}
