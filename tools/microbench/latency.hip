// Pointer-chase latency of the MI355X memory hierarchy (one lane, dependent loads, 128-byte stride, random cycle).
// build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency ; prints cycles (s_memtime) and ns per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const unsigned* __restrict__ buf, int hops, unsigned start, unsigned long long* out) {
  unsigned i = start;
  const unsigned long long t0 = clock64();
  for (int k = 0; k < hops; ++k) i = buf[(size_t)i * 32];
  const unsigned long long t1 = clock64();
  out[0] = t1 - t0; out[1] = i;
}
__global__ void touch(const unsigned* __restrict__ buf, size_t n, unsigned* sink) {  // other waves pull the lines into THEIR L2
  unsigned acc = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += buf[i * 32];
  if (acc == 0xdeadbeef) *sink = acc;
}
int main() {
  unsigned long long* out; hipMalloc(&out, 16);
  unsigned* sink; hipMalloc(&sink, 4);
  std::mt19937 rng(1);
  for (size_t bytes : {size_t(16) << 10, size_t(256) << 10, size_t(2) << 20, size_t(16) << 20, size_t(128) << 20, size_t(1) << 30, size_t(4) << 30}) {
    const size_t n = bytes / 128;
    std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<unsigned> host(n * 32, 0u);
    for (size_t k = 0; k < n; ++k) host[(size_t)perm[k] * 32] = perm[(k + 1) % n];
    unsigned* dev; hipMalloc(&dev, bytes); hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice);
    const int hops = (int)std::min<size_t>(n, 20000);
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) touch<<<1024, 256>>>(dev, n, sink);   // second run: lines were just touched by all XCDs (in MALL / other L2s)
      chase<<<1, 1>>>(dev, hops, perm[0], out);
      unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double coldClk = (double)h[0] / hops;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a); chase<<<1, 1>>>(dev, hops, perm[0], out); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%8.2f MiB  hops %6d  first pass %7.1f clk/hop  second pass %7.1f clk/hop  %7.1f ns/hop (event)  %s\n", bytes / 1048576.0, hops, coldClk, (double)h[0] / hops, ms * 1e6 / hops,
             rep ? "after all-XCD touch" : "");
    }
    hipFree(dev);
  }
  return 0;
}
