// wave_utils.h -- wave64 / workgroup scan and reduce helpers for 256-thread workgroups.
#pragma once

#include <hip/hip_runtime.h>

namespace itm {

constexpr int kWave = 64;

__device__ inline int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ inline int wave_id() { return threadIdx.x / kWave; }

// Cross-lane steps as data-parallel primitives (DPP): the shift happens in the ALU's operand path.  The __shfl_* forms they
// replace are ds_bpermute instructions, one LDS round trip per step (a 6-step reduction cost 0.4-0.9 us of pure latency in the
// ordered sweeps and in the tracker's reduction).
//   row_shr:n   (0x110 + n)  lane i reads lane i - n of its row of 16; lanes without a source read 0 (bound_ctrl)
//   row_bcast:15 (0x142)     lane 15 of every row to the next row      (row mask 0xa: rows 1 and 3 take it)
//   row_bcast:31 (0x143)     lane 31 to the rows above it               (row mask 0xc: rows 2 and 3 take it)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ inline int dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, BOUND); }

// inclusive scan across the 64 lanes of a wave
__device__ inline int wave_inclusive_scan(int v) {
  v += dpp_int<0x111, 0xf, true>(v);
  v += dpp_int<0x112, 0xf, true>(v);
  v += dpp_int<0x114, 0xf, true>(v);
  v += dpp_int<0x118, 0xf, true>(v);      // every row scanned
  v += dpp_int<0x142, 0xa, false>(v);     // rows 1, 3 += total of the row before
  v += dpp_int<0x143, 0xc, false>(v);     // rows 2, 3 += total of rows 0 + 1
  return v;
}

// sum over the wave, the same value in every lane
__device__ inline int wave_reduce_sum(int v) { return __builtin_amdgcn_readlane(wave_inclusive_scan(v), kWave - 1); }

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
