// range_device.h -- block projection for the expected-depth range image, shared by the stand-alone
// CreateExpectedDepths kernels (visualise.hip) and the fused integrate + projection launch (integrate.hip).
//
// Reference: ProjectSingleBlock  DeviceAgnostic/ITMVisualisationEngine.h:28-90,
//            CreateExpectedDepths DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:79-152
#pragma once

#include "itm_internal.h"
#include "wave_utils.h"

namespace itm {

struct ProjParams {
  Mat4 M;
  float fx, fy, cx, cy;
  float voxelSize;
  int W, H;
  int maxBlocks;
};

struct Projected { int ulx, uly, lrx, lry; float z0, z1; int n; };

// ProjectSingleBlock + the tile count of CreateExpectedDepths (:128-131)
__device__ inline Projected project_block(const HashEntry& e, const ProjParams& p) {
  Projected r;
  r.ulx = p.W / 8; r.uly = p.H / 8; r.lrx = -1; r.lry = -1; r.z0 = 999999.9f; r.z1 = 0.05f; r.n = 0;
  if (e.ptr < 0) return r;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int16_t tx = (int16_t)(e.px + ((corner & 1) ? 1 : 0));
    const int16_t ty = (int16_t)(e.py + ((corner & 2) ? 1 : 0));
    const int16_t tz = (int16_t)(e.pz + ((corner & 4) ? 1 : 0));
    const float x = (float)tx * (float)kBlockSide * p.voxelSize;
    const float y = (float)ty * (float)kBlockSide * p.voxelSize;
    const float z = (float)tz * (float)kBlockSide * p.voxelSize;
    const Vec3 q = transform_point(p.M, x, y, z);
    if ((double)q.z < 1e-6) continue;  // double literal in the reference
    const float u = (p.fx * q.x / q.z + p.cx) / 8;
    const float v = (p.fy * q.y / q.z + p.cy) / 8;
    if ((float)r.ulx > floorf(u)) r.ulx = (int)floorf(u);
    if ((float)r.lrx < ceilf(u)) r.lrx = (int)ceilf(u);
    if ((float)r.uly > floorf(v)) r.uly = (int)floorf(v);
    if ((float)r.lry < ceilf(v)) r.lry = (int)ceilf(v);
    if (r.z0 > q.z) r.z0 = q.z;
    if (r.z1 < q.z) r.z1 = q.z;
  }
  if (r.ulx < 0) r.ulx = 0;
  if (r.uly < 0) r.uly = 0;
  if (r.lrx >= p.W) r.lrx = p.W - 1;
  if (r.lry >= p.H) r.lry = p.H - 1;
  if (r.ulx > r.lrx) return r;
  if (r.uly > r.lry) return r;
  if (r.z0 < 0.05f) r.z0 = 0.05f;
  if (r.z1 < 0.05f) return r;
  const int nx = (int)ceilf((float)(r.lrx - r.ulx + 1) / 16.0f);
  const int ny = (int)ceilf((float)(r.lry - r.uly + 1) / 16.0f);
  r.n = nx * ny;
  return r;
}

// LDS variant of the box merge: kRangeParts workgroups, each projects a slice of the visible list and
// min/max-merges the boxes into its own LDS copy of the [0,W/8)x[0,H/8) region with LDS atomics, then
// writes that partial image to HBM (reduced by range_reduce_kernel).  Cells outside the region (the
// reference clamps boxes to the FULL image size, a quirk that only touches cells no ray ever reads) go
// through global atomics.
// (re-measured at the end of round 4, BASELINE configs[1] frames/s through the four calls / configs[4]: 4 parts 10.8-10.9 k / 2 800 -- the
// projection becomes the integration launch's long pole, 31 us --, 8: 12.0 k / 2 902, 16: 12.46-12.54 k / 2 901-2 906, 32: 12.28-12.37 k /
// 2 892-2 899, 64: 12.1-12.2 k / 2 857: every ray-cast workgroup reduces 4 cells x parts partial values in its prologue.  With 5-12 times as
// many visible blocks at 640 x 480 (2 mm / 1.5 mm voxels, tools/fine_voxel_bench.py) 16 parts are still no slower than 32: 4 765 / 2 867 against
// 4 670 / 2 857 frames/s)
constexpr int kRangeParts = 16;   // 16, 32 or 64 (the ray-cast prologue reduces 4 cells x kRangeParts partials with <= 256 lanes)

// Rendering-block cap of the reference reached (numRenderingBlocks >= MAX_RENDERING_BLOCKS): replays the sequential
// accept / skip decisions and rebuilds the whole image from the accepted boxes.  One workgroup of `nthreads` lanes.
__device__ inline void range_replay_capped(int tid, int nthreads, uint2* cells, RenderCounters* __restrict__ rc, float2* __restrict__ range,
                                           uint4* __restrict__ projBuf, const ProjParams& p, int RW, int RH) {
  const int nCells = RW * RH;
  const int nv = rc->noVisibleEntries;
  const uint2 initCell = make_uint2(__float_as_uint(999999.9f), __float_as_uint(0.05f));
  for (int i = tid; i < nCells; i += nthreads) cells[i] = initCell;
  for (int i = tid; i < p.W * p.H; i += nthreads) range[i] = make_float2(999999.9f, 0.05f);
  if (tid == 0) {
    int count = 0;
    for (int e = 0; e < nv; ++e) {
      uint4 b = projBuf[2 * e + 1];
      const int n = (int)b.z;
      if (n == 0) continue;
      if (count + n >= p.maxBlocks) b.w = 0u; else { b.w = 1u; count += n; }
      projBuf[2 * e + 1] = b;
    }
    rc->renderingBlocksAccepted = count;
  }
  __threadfence();
  __syncthreads();
  for (int e = tid; e < nv; e += nthreads) {
    const uint4 a = projBuf[2 * e], b = projBuf[2 * e + 1];
    if (b.z == 0u || b.w == 0u) continue;
    for (int y = (int)a.y; y <= (int)a.w; ++y)
      for (int x = (int)a.x; x <= (int)a.z; ++x) {
        if (x < RW && y < RH) {
          atomicMin(&cells[x + y * RW].x, b.x);
          atomicMax(&cells[x + y * RW].y, b.y);
        } else {
          uint32_t* px = (uint32_t*)&range[x + y * p.W];
          atomicMin(px, b.x);
          atomicMax(px + 1, b.y);
        }
      }
  }
  __syncthreads();
  for (int i = tid; i < nCells; i += nthreads) {
    const int y = i / RW, x = i - y * RW;
    const uint2 c = cells[i];
    range[x + y * p.W] = make_float2(__uint_as_float(c.x), __uint_as_float(c.y));
  }
}

// min / max over the kRangeParts partial images for one cell
__device__ inline void range_reduce_cell(int i, const uint2* __restrict__ partials, float2* __restrict__ range, int nCells, int RW, int W) {
  uint2 c[kRangeParts];
#pragma unroll
  for (int g = 0; g < kRangeParts; ++g) c[g] = partials[(size_t)g * nCells + i];
  uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int g = 0; g < kRangeParts; ++g) { lo = c[g].x < lo ? c[g].x : lo; hi = c[g].y > hi ? c[g].y : hi; }
  const int y = i / RW, x = i - y * RW;
  range[x + y * W] = make_float2(__uint_as_float(lo), __uint_as_float(hi));
}

// (Letting the kRangeParts workgroups meet at an in-kernel counter and reduce the partial images themselves, to
// save the reduction launch, was measured slower: the agent-scope fences write back / invalidate the L2s the
// integration workgroups are using -- fused launch 26.7 -> 41.4 us for 6.7 us saved.)
__device__ inline void project_partial_body(int part, uint2* cells, const int32_t* __restrict__ ids, RenderCounters* __restrict__ rc,
                                            const uint4* __restrict__ hash, float2* __restrict__ range, uint4* __restrict__ projBuf,
                                            uint2* __restrict__ partials, const ProjParams& p, int RW, int RH) {
  __shared__ int lds[8];
  const int tid = threadIdx.x;
  const int nCells = RW * RH;
  const uint2 initCell = make_uint2(__float_as_uint(999999.9f), __float_as_uint(0.05f));
  for (int i = tid; i < nCells; i += 512) cells[i] = initCell;
  __syncthreads();
  const int nv = rc->noVisibleEntries;
  int need = 0;
  for (int e = part * 512 + tid; e < nv; e += kRangeParts * 512) {
    const HashEntry he = unpack_entry(hash[ids[e]]);
    const Projected r = project_block(he, p);
    projBuf[2 * e] = make_uint4((uint32_t)r.ulx, (uint32_t)r.uly, (uint32_t)r.lrx, (uint32_t)r.lry);
    projBuf[2 * e + 1] = make_uint4(__float_as_uint(r.z0), __float_as_uint(r.z1), (uint32_t)r.n, 1u);
    if (r.n == 0) continue;
    need += r.n;
    const uint32_t z0 = __float_as_uint(r.z0), z1 = __float_as_uint(r.z1);
    for (int y = r.uly; y <= r.lry; ++y)
      for (int x = r.ulx; x <= r.lrx; ++x) {
        if (x < RW && y < RH) {
          atomicMin(&cells[x + y * RW].x, z0);
          atomicMax(&cells[x + y * RW].y, z1);
        } else {
          uint32_t* px = (uint32_t*)&range[x + y * p.W];
          atomicMin(px, z0);
          atomicMax(px + 1, z1);
        }
      }
  }
  const int sum = block_reduce_sum<8>(need, lds);
  if (tid == 0 && sum) atomicAdd(&rc->noRenderingBlocks, sum);
  __syncthreads();
  uint2* mine = partials + (size_t)part * nCells;
  for (int i = tid; i < nCells; i += 512) mine[i] = cells[i];
}

}  // namespace itm
