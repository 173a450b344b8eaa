// wave_utils.h -- wave64 / workgroup scan and reduce helpers for 256-thread workgroups.
#pragma once

#include <hip/hip_runtime.h>

namespace itm {

constexpr int kWave = 64;

__device__ inline int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ inline int wave_id() { return threadIdx.x / kWave; }

// inclusive scan across the 64 lanes of a wave
__device__ inline int wave_inclusive_scan(int v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int o = __shfl_up(v, d, kWave);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ inline int wave_reduce_sum(int v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_down(v, d, kWave);
  return __shfl(v, 0, kWave);
}

// Exclusive scan of one int per thread over a workgroup of NW waves (blockDim.x == NW*64).
// `lds` needs NW+1 ints.  Returns the exclusive prefix; *total receives the workgroup sum.
template <int NW>
__device__ inline int block_exclusive_scan(int v, int* lds, int* total) {
  int inc = wave_inclusive_scan(v);
  const int lane = lane_id(), w = wave_id();
  __syncthreads();  // protect lds reuse across consecutive calls
  if (lane == kWave - 1) lds[w] = inc;
  __syncthreads();
  int base = 0, sum = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    int t = lds[i];
    if (i < w) base += t;
    sum += t;
  }
  *total = sum;
  return base + inc - v;
}

template <int NW>
__device__ inline int block_reduce_sum(int v, int* lds) {
  int s = wave_reduce_sum(v);
  __syncthreads();
  if (lane_id() == 0) lds[wave_id()] = s;
  __syncthreads();
  int sum = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) sum += lds[i];
  return sum;
}

}  // namespace itm
