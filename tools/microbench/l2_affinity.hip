// Do lines WRITTEN by one kernel stay in the writer XCD's L2 for the next kernel, and is workgroup -> XCD
// placement the same round-robin in both launches?  Kernel A: workgroup b builds pointer chain b (device writes).
// Kernel B: workgroup b chases chain (b + shift) % G.  L2-hit latency for shift % 8 == 0 only => yes to both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int LINES = 4096;   // 512 KiB per chain
__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__global__ void build(unsigned* buf, unsigned* xcc) {
  unsigned* c = buf + (size_t)blockIdx.x * LINES * 32;
  for (int k = threadIdx.x; k < LINES; k += blockDim.x) c[(size_t)k * 32] = (unsigned)((k * 1237u + 331u) % LINES);  // full cycle (1237 odd, LINES power of two)
  if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
}
__global__ void chase(const unsigned* buf, int shift, unsigned long long* out, unsigned* xcc) {
  const int g = (blockIdx.x + shift) % gridDim.x;
  const unsigned* c = buf + (size_t)g * LINES * 32;
  if (threadIdx.x != 0) return;
  unsigned i = 0;
  const unsigned long long t0 = clock64();
  for (int k = 0; k < 1024; ++k) i = __builtin_nontemporal_load(&c[(size_t)i * 32]) ;
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = i; xcc[blockIdx.x] = xcc_id();
}
int main() {
  const int G = 32;
  unsigned* buf; hipMalloc(&buf, (size_t)G * LINES * 128);
  unsigned long long* out; hipMalloc(&out, G * 16);
  unsigned *xa, *xb; hipMalloc(&xa, G * 4); hipMalloc(&xb, G * 4);
  for (int shift : {0, 8, 0, 1, 16, 0, 9}) {
    build<<<G, 256>>>(buf, xa);
    chase<<<G, 64>>>(buf, shift, out, xb);
    std::vector<unsigned long long> h(G * 2); std::vector<unsigned> a(G), b(G);
    hipMemcpy(h.data(), out, G * 16, hipMemcpyDeviceToHost); hipMemcpy(a.data(), xa, G * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), xb, G * 4, hipMemcpyDeviceToHost);
    printf("shift %d:", shift);
    for (int g = 0; g < G; ++g) printf("  wg%d[xcc %u->%u] %.0f", g, a[(g + shift) % G], b[g], (double)h[g * 2] / 1024);
    printf("\n");
  }
  return 0;
}
