// visualise.hip -- expected-depth range image, ray casting, ICP maps and free-view rendering.
//
// Reference behaviour:
//   CreateExpectedDepths   DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:79-91 (dense), :93-152 (hash)
//   ProjectSingleBlock / CreateRenderingBlocks   DeviceAgnostic/ITMVisualisationEngine.h:28-90
//   GenericRaycast         ITMVisualisationEngine_CPU.cpp:154-188
//   CreateICPMaps_common   :266-287, processPixelICP<true>  DeviceAgnostic/ITMVisualisationEngine.h:314-349
//   RenderImage_common     :190-240, processPixelGrey/Colour/Normal  :368-409
//   computeSingleNormalFromSDF / readFromSDF_color4u_interpolated  DeviceAgnostic/ITMRepresentationAccess.h:187-337
//
// MI355X design:
//   * range image: one lane per visible block projects its 8 corners and min/max-merges its
//     bounding box straight into the range image with integer atomics on the float bit patterns
//     (all values are positive, so uint ordering == float ordering).  The intermediate list of
//     16x16 "rendering blocks" of the reference only matters through its cap
//     (MAX_RENDERING_BLOCKS); the cap is honoured exactly by an overflow pass that replays the
//     sequential accept/skip decisions only when the total reaches the cap.
//   * ray casting: one lane per pixel, 16x4 pixels per wave (two 8x8 range cells per wave), a
//     per-lane block cache as in the reference; 2-byte sdf gathers from HBM/L2.
//   * ICP maps: one lane per pixel over the ray-hit map.
#include <cstdlib>
#include <cstring>

#include "itm_internal.h"
#include "alloc_device.h"
#include "shading_device.h"
#include "range_device.h"
#include "wave_utils.h"


namespace itm {

#ifndef ITM_MIRROR_FLOAT_TYPES
#define ITM_MIRROR_FLOAT_TYPES 0     // (scene.hip decides whether a float scene gets a mirror; the same switch must be given to every file)
#endif

int g_debug_force_global_range = 0;
int g_debug_no_directory = 0;
int g_debug_no_sdf_mirror = 0;      // debug key 12: ray casting reads voxels through the directory / table although the scene has an sdf mirror
int g_debug_no_fused_range_reduce = 0;
int g_debug_single_pass_raycast = 0;
int g_debug_no_side_projection = 0;
int g_debug_dense_range_refill = 0;

// ---------------------------------------------------------------------------------------------
// expected depth range
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) range_init_kernel(float2* __restrict__ img, int n, float a, float b, RenderCounters* rc) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) img[i] = make_float2(a, b);
  if (blockIdx.x == 0 && threadIdx.x == 0) { rc->noRenderingBlocks = 0; rc->renderingBlocksAccepted = -1; }
}

__device__ inline void merge_box(float2* __restrict__ range, int W, const Projected& r) {
  const uint32_t z0 = __float_as_uint(r.z0), z1 = __float_as_uint(r.z1);
  for (int y = r.uly; y <= r.lry; ++y)
    for (int x = r.ulx; x <= r.lrx; ++x) {
      uint32_t* px = (uint32_t*)&range[x + y * W];
      atomicMin(px, z0);
      atomicMax(px + 1, z1);
    }
}

__global__ void __launch_bounds__(256) project_fill_kernel(const int32_t* __restrict__ ids, RenderCounters* __restrict__ rc,
                                                           const uint4* __restrict__ hash, float2* __restrict__ range,
                                                           uint4* __restrict__ projBuf, ProjParams p) {
  const int nv = rc->noVisibleEntries;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nv; e += gridDim.x * blockDim.x) {
    const HashEntry he = unpack_entry(hash[ids[e]]);
    const Projected r = project_block(he, p);
    projBuf[2 * e] = make_uint4((uint32_t)r.ulx, (uint32_t)r.uly, (uint32_t)r.lrx, (uint32_t)r.lry);
    projBuf[2 * e + 1] = make_uint4(__float_as_uint(r.z0), __float_as_uint(r.z1), (uint32_t)r.n, 0u);
    if (r.n > 0) {
      atomicAdd(&rc->noRenderingBlocks, r.n);
      merge_box(range, p.W, r);  // optimistic: undone by range_overflow_kernel if the cap is reached
    }
  }
}

// Only does work when the cap of the reference's rendering-block list would have been hit:
// "if (numRenderingBlocks + required >= MAX) skip this block" is order dependent, so it is replayed
// sequentially over the (ascending) visible list and the range image is rebuilt.
__global__ void __launch_bounds__(256) range_overflow_kernel(RenderCounters* __restrict__ rc, float2* __restrict__ range,
                                                             uint4* __restrict__ projBuf, ProjParams p) {
  if (rc->noRenderingBlocks < p.maxBlocks) return;
  const int nv = rc->noVisibleEntries;
  const int n = p.W * p.H;
  for (int i = threadIdx.x; i < n; i += blockDim.x) range[i] = make_float2(999999.9f, 0.05f);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    int count = 0;
    for (int e = 0; e < nv; ++e) {
      uint4 b = projBuf[2 * e + 1];
      const int need = (int)b.z;
      if (need == 0) continue;
      if (count + need >= p.maxBlocks) { b.w = 0u; } else { b.w = 1u; count += need; }
      projBuf[2 * e + 1] = b;
    }
    rc->renderingBlocksAccepted = count;
  }
  __threadfence();
  __syncthreads();
  for (int e = threadIdx.x; e < nv; e += blockDim.x) {
    const uint4 a = projBuf[2 * e], b = projBuf[2 * e + 1];
    if (b.z == 0u || b.w == 0u) continue;
    Projected r;
    r.ulx = (int)a.x; r.uly = (int)a.y; r.lrx = (int)a.z; r.lry = (int)a.w;
    r.z0 = __uint_as_float(b.x); r.z1 = __uint_as_float(b.y); r.n = (int)b.z;
    merge_box(range, p.W, r);
  }
}

// LDS variant, used whenever the sub-sampled range image fits in LDS (160 KiB per CU: up to ~19k
// cells, i.e. 1280x960 frames).  Global atomics on the 4800-cell image are contention bound
// (~29 us for 10k blocks) and a single workgroup is ALU bound on one CU (~77 us), so the work is
// split in two launches:
//   project_partial_kernel : kRangeParts workgroups, each projects a slice of the visible list and
//       min/max-merges the boxes into its own LDS copy of the [0,W/8)x[0,H/8) region with LDS
//       atomics, then writes that partial image to HBM.  Cells outside the region (the reference
//       clamps boxes to the FULL image size, a quirk that only touches cells no ray ever reads) go
//       through global atomics.
//   range_reduce_kernel    : one workgroup reduces the partial images into the range image; if the
//       rendering-block cap of the reference was reached it instead replays the sequential
//       accept / skip decisions and rebuilds the image from the accepted boxes.
__global__ void __launch_bounds__(512) project_partial_kernel(const int32_t* __restrict__ ids, RenderCounters* __restrict__ rc,
                                                              const uint4* __restrict__ hash, float2* __restrict__ range,
                                                              uint4* __restrict__ projBuf, uint2* __restrict__ partials,
                                                              ProjParams p, int RW, int RH) {
  extern __shared__ uint2 cells[];
  project_partial_body(blockIdx.x, cells, ids, rc, hash, range, projBuf, partials, p, RW, RH);
}

__global__ void __launch_bounds__(256) range_reduce_kernel(RenderCounters* __restrict__ rc, float2* __restrict__ range,
                                                            uint4* __restrict__ projBuf, const uint2* __restrict__ partials,
                                                            ProjParams p, int RW, int RH) {
  extern __shared__ uint2 cells[];
  const int tid = threadIdx.x;
  const int nCells = RW * RH;
  if (rc->noRenderingBlocks < p.maxBlocks) {
    const int i = blockIdx.x * 256 + tid;   // one cell per lane, kRangeParts independent loads
    if (i < nCells) range_reduce_cell(i, partials, range, nCells, RW, p.W);
    return;
  }
  if (blockIdx.x != 0) return;
  // cap reached (uniform branch): sequential replay, then rebuild the whole image
  range_replay_capped(tid, 256, cells, rc, range, projBuf, p, RW, RH);
}

// true when the projection can ride in the integration launch (integrate.hip): four workgroups per CU must keep
// their LDS copy of the sub-sampled range image
bool can_fuse_projection(const itm_scene* s, const itm_render_state* rs) {
  const size_t ldsBytes = (size_t)((rs->w + 7) / 8) * ((rs->h + 7) / 8) * sizeof(uint2);
  return !g_debug_no_fused_projection && s->cfg.indexType == ITM_INDEX_HASH && ldsBytes <= 39 * 1024 && rs->rangePartials && !g_debug_force_global_range;
}

static int ensure_range_lds(const itm_scene* s) {
  // per device (a function attribute belongs to the code object loaded on ONE device)
  static bool attrSet[64] = {};
  const int dev = (s->device >= 0 && s->device < 64) ? s->device : 0;
  if (!attrSet[dev]) {
    ITM_HIP(hipFuncSetAttribute((const void*)project_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    ITM_HIP(hipFuncSetAttribute((const void*)range_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attrSet[dev] = true;
  }
  return ITM_OK;
}

static ProjParams make_proj_params(const itm_scene* s, const float* M, const float* intr, const itm_render_state* rs) {
  ProjParams p;
  memcpy(p.M.m, M, 64);
  p.fx = intr[0]; p.fy = intr[1]; p.cx = intr[2]; p.cy = intr[3];
  p.voxelSize = s->prm.voxelSize;
  p.W = rs->w; p.H = rs->h;
  p.maxBlocks = s->cfg.maxRenderingBlocks;
  return p;
}

// The projection half of CreateExpectedDepths beside the integration instead of after it, for images whose sub-sampled range
// image is too large for the fused launch (can_fuse_projection: 1280x960 needs 150 KB of LDS per projecting workgroup): it
// depends on the visible list only, so it runs on a stream of the render state's own between the allocation and the reduction --
// 28 us of config 5's frame that used to sit between the integration and the ray cast.  Returns 1 when it was launched (the caller
// then passes projected = true to launch_expected_depths), 0 when this path does not apply, < 0 on error.
int launch_projection_beside(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, hipStream_t st) {
  if (s->cfg.indexType != ITM_INDEX_HASH || g_debug_force_global_range || g_debug_no_side_projection || !rs->rangePartials) return 0;
  const int RW = (rs->w + 7) / 8, RH = (rs->h + 7) / 8;
  const size_t ldsBytes = (size_t)RW * RH * sizeof(uint2);
  if (ldsBytes > 150 * 1024) return 0;
  if (!rs->sideStream) {
    // the highest stream priority the device offers: the projection's 16 workgroups each need a whole compute unit's LDS and two wave
    // slots on every SIMD at once; beside an integration launch of thousands of small workgroups that is dispatched from a queue of
    // equal rank they are placed only when that launch runs out (measured, BASELINE configs[4], integration in 128-lane workgroups:
    // integration 114 us, frame 324 us instead of 293) -- from the higher-priority queue they are placed first and the integration
    // fills the rest of the device
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    if (hipStreamCreateWithPriority(&rs->sideStream, hipStreamNonBlocking, greatest) != hipSuccess || hipEventCreateWithFlags(&rs->listReady, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&rs->projectionDone, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) {
      (void)hipGetLastError();
      if (rs->sideStream) { (void)hipStreamDestroy(rs->sideStream); rs->sideStream = nullptr; }
      return 0;
    }
  }
  int rc = ensure_range_lds(s);
  if (rc) return rc;
  const ProjParams p = make_proj_params(s, M, intr, rs);
  ITM_HIP(hipEventRecord(rs->listReady, st));
  ITM_HIP(hipStreamWaitEvent(rs->sideStream, rs->listReady, 0));
  project_partial_kernel<<<kRangeParts, 512, ldsBytes, rs->sideStream>>>(rs->visibleIds, rs->counters, s->hash, rs->range, rs->projBuf, rs->rangePartials, p, RW, RH);
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipEventRecord(rs->projectionDone, rs->sideStream));
  return 1;
}

// `projected`: project_partial already ran inside the integration launch, only the reduction is left
int launch_expected_depths(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, bool rangeAlreadyInit, hipStream_t st, bool projected) {
  const int P = rs->w * rs->h;
  KernelTimer tk(s, ITM_TK_RANGE, st);
  if (s->cfg.indexType == ITM_INDEX_DENSE) {
    if (rs->denseRangeReady && !g_debug_dense_range_refill) return ITM_OK;        // the image already holds the constant (5 us per frame of config 3)
    range_init_kernel<<<512, 256, 0, st>>>(rs->range, P, 0.2f, 3.0f, rs->counters);
    ITM_LAUNCH_CHECK();
    rs->denseRangeReady = true;
    return ITM_OK;
  }
  const ProjParams p = make_proj_params(s, M, intr, rs);
  if (!rangeAlreadyInit) range_init_kernel<<<512, 256, 0, st>>>(rs->range, P, 999999.9f, 0.05f, rs->counters);
  const int RW = (rs->w + 7) / 8, RH = (rs->h + 7) / 8;
  const size_t ldsBytes = (size_t)RW * RH * sizeof(uint2);
  const bool forceGlobal = g_debug_force_global_range != 0;  // test hook for the fallback path
  if (ldsBytes <= 150 * 1024 && !forceGlobal && rs->rangePartials) {
    int rc = ensure_range_lds(s);
    if (rc) return rc;
    if (!projected) project_partial_kernel<<<kRangeParts, 512, ldsBytes, st>>>(rs->visibleIds, rs->counters, s->hash, rs->range, rs->projBuf, rs->rangePartials, p, RW, RH);
    range_reduce_kernel<<<(RW * RH + 255) / 256, 256, ldsBytes, st>>>(rs->counters, rs->range, rs->projBuf, rs->rangePartials, p, RW, RH);
  } else {
    project_fill_kernel<<<128, 256, 0, st>>>(rs->visibleIds, rs->counters, s->hash, rs->range, rs->projBuf, p);
    range_overflow_kernel<<<1, 256, 0, st>>>(rs->counters, rs->range, rs->projBuf, p);
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

// ---------------------------------------------------------------------------------------------
// ray casting
// ---------------------------------------------------------------------------------------------
// One workgroup = 16x16 pixels; wave w covers rows 4w..4w+3 (16x4 pixels, two 8x8 range cells).
//
// REDUCE (itm_process_frame on hash scenes): the workgroup first reduces the kRangeParts partial range images of ITS 2x2 cells
// (the second half of CreateExpectedDepths, otherwise range_reduce_kernel), stores them to the range image and casts its rays
// from the LDS copy -- one launch and one 1.2 MB round trip through memory less per frame.  When the rendering-block cap of
// the reference was reached (noRenderingBlocks >= MAX_RENDERING_BLOCKS, i.e. > 262 144 tiles: never with real data, but defined
// behaviour) every workgroup replays the sequential accept / skip decisions over the visible list itself; slow, exact.
struct RangeFuse {
  const uint2* partials;      // [kRangeParts][RW * RH]
  RenderCounters* rc;
  const uint4* projBuf;       // per visible entry: box, z range, tile count
  float2* range;              // range image (row stride W)
  int RW, RH, maxBlocks;
};

__device__ inline void reduce_own_cells(const RangeFuse& f, int tx, int ty, int W, float2* cellRange) {
  const int tid = threadIdx.x;
  const int nCells = f.RW * f.RH;
  // 4 cells x kRangeParts partials: one load per lane (kRangeParts 32: waves 0 and 1, min / max over each 32-lane half; 64: all
  // four waves, one cell per wave).  Requested before the counter that decides whether they are used: one round trip, not two, at the
  // head of every tile
  const int c = tid / kRangeParts, part = tid % kRangeParts;
  const int cx = tx * 2 + (c & 1), cy = ty * 2 + (c >> 1);
  const bool inside = tid < 4 * kRangeParts && cx < f.RW && cy < f.RH;
  uint2 v = make_uint2(0xffffffffu, 0u);
  if (inside) v = f.partials[(size_t)part * nCells + cx + cy * f.RW];
  const int total = f.rc->noRenderingBlocks;
  if (total < f.maxBlocks) {
    if (tid < 4 * kRangeParts) {
#pragma unroll
      for (int o = kRangeParts / 2; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor(v.x, o, 64), hi = __shfl_xor(v.y, o, 64);
        v.x = lo < v.x ? lo : v.x; v.y = hi > v.y ? hi : v.y;
      }
      if (part == 0) {
        const float2 r = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
        cellRange[c] = r;
        if (inside) f.range[cx + cy * W] = r;
      }
    }
    return;
  }
  // ---- cap reached: sequential replay by wave 0, 64 entries per round ----
  __shared__ uint32_t capCells[8];   // (min bits, max bits) x 4 cells
  if (tid < 4) { capCells[2 * tid] = __float_as_uint(999999.9f); capCells[2 * tid + 1] = __float_as_uint(0.05f); }
  __syncthreads();
  if (tid < 64) {
    const int nv = f.rc->noVisibleEntries;
    int count = 0;
    for (int base = 0; base < nv; base += 64) {
      const int e = base + tid;
      uint4 box = make_uint4(0, 0, 0, 0), zr = make_uint4(0, 0, 0, 0);
      if (e < nv) { box = f.projBuf[2 * e]; zr = f.projBuf[2 * e + 1]; }
      const int need = (int)zr.z;
      unsigned long long accept = 0ull;
      for (int l = 0; l < 64; ++l) {                        // the reference's loop, one entry at a time (uniform control flow)
        const int n = __shfl(need, l, 64);
        if (n != 0 && count + n < f.maxBlocks) { count += n; accept |= 1ull << l; }
      }
      if ((accept >> tid) & 1ull) {
        for (int c = 0; c < 4; ++c) {
          const int cx = tx * 2 + (c & 1), cy = ty * 2 + (c >> 1);
          if (cx >= (int)box.x && cx <= (int)box.z && cy >= (int)box.y && cy <= (int)box.w) {
            atomicMin(&capCells[2 * c], zr.x);
            atomicMax(&capCells[2 * c + 1], zr.y);
          }
        }
      }
    }
    if (blockIdx.x == 0 && tid == 0) f.rc->renderingBlocksAccepted = count;
  }
  __syncthreads();
  if (tid < 4) {
    const int cx = tx * 2 + (tid & 1), cy = ty * 2 + (tid >> 1);
    const float2 r = make_float2(__uint_as_float(capCells[2 * tid]), __uint_as_float(capCells[2 * tid + 1]));
    cellRange[tid] = r;
    if (cx < f.RW && cy < f.RH) f.range[cx + cy * W] = r;
  }
}

// PARK (hash scenes with a block directory): rays that are crossing empty space -- in BASELINE configs[1] the 16 % of the rays
// that pass the sphere's silhouette and walk ~45 "no block" steps to the wall -- are parked by phase 1 in a queue in LDS.  Mixed
// into waves with ordinary rays they made those waves 3-5x longer than the median wave (sphere-side steps, then the run, then
// the wall-side steps, one after the other for the whole wave), and the launch lasts as long as its slowest wave.  Phase 2 of
// the same workgroup re-packs the parked rays 64 per wave: all lanes are then in the same phase, and the look-ahead of
// march_ray turns the ~45 dependent round trips of the run into ~8.  (As a second LAUNCH over a global queue the parked rays
// took 52 us on their own -- one wave per SIMD, other XCDs' cold L2s -- against 34 us for the first pass; measured, dropped.)
constexpr float kSilhouetteSpread = 0.3f;   // metres between the near and far bound of a tile's expected depths from which the tile counts as holding a silhouette
constexpr int kSilhouettePrio = 3;          // s_setprio of such a tile's waves (0.03 m, i.e. nearly every tile: no gain; priority only in phase 2: no gain)
#ifndef ITM_EXP_RAYCAST_STAMPS
#define ITM_EXP_RAYCAST_STAMPS 0   // measurement build: per-wave timeline of the ray-cast launch on the 100 MHz clock (tools/raycast_timeline.py)
#endif
#if ITM_EXP_RAYCAST_STAMPS
__device__ unsigned long long g_rayStamps[8192 * 4];   // per wave: start, after the prologue, end of phase 1, end of phase 2 | parked rays of the tile << 52
#define ITM_RS(...) __VA_ARGS__
#else
#define ITM_RS(...)
#endif

// AHEAD (itm_process_frame_ahead): the workgroups beyond the image's tiles issue the block requests of the NEXT frame.  They are
// dispatched last, i.e. into the compute units that the ordinary tiles leave while the launch waits for its silhouette tiles (81 % of
// the tiles are done after a third of the launch): the request stage of the next frame costs the frame nothing.  It reads the table /
// slot directory / next depth image and writes request keys, visible types and request counters, none of which a ray touches.
struct AheadRequest { RequestArgs ra; AllocParams ap; int rayTiles, reqTilesX; };

// MIRROR: the form of the sdf mirror this instance is compiled for (raycast_device.h, VolumeViewM): 0 none, 1 dense cube, 2 paged
template <class VX, bool DENSE, bool REDUCE, bool PARK, bool AHEAD = false, int MIRROR = 0>
__global__ void __launch_bounds__(256) raycast_kernel(VolumeView volIn, const float2* __restrict__ range, float4* __restrict__ out, RayParams p, RangeFuse fuse, AheadRequest ahead) {
  if constexpr (AHEAD) {
    if ((int)blockIdx.x >= ahead.rayTiles) {
      const int r = (int)blockIdx.x - ahead.rayTiles;
      request_tile<false, false, true>(r % ahead.reqTilesX, r / ahead.reqTilesX, ahead.ra, ahead.ap);
      return;
    }
  }
  VolumeViewM<MIRROR> vol(volIn);
  if constexpr (MIRROR == 0) vol.sdfMirror = nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (An XCD-affine order -- image band b ray-cast by XCD b, with integration placing the blocks of band b on XCD b --
  // recovers 5 of the ~17 us the rays lose to voxel lines written on other XCDs, but costs integration 10 us; and 8x8
  // pixel waves change nothing.  Both measured)
  const int tilesX = (p.W + 15) / 16;
  const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
  // (Dealing a tile's rays to its waves long rays first, by the previous cast's per-pixel read counts, was built and measured in round 4:
  // bit-exact and slower, profiles/r4_raycast_notes.md section 3.)
  const int lx = lane & 15, ly = wave * 4 + (lane >> 4);
  const int x = tx * 16 + lx;
  const int y = ty * 16 + ly;
  const bool inside = x < p.W && y < p.H;
  __shared__ float2 cellRange[4];
  __shared__ float4 parkState[PARK ? 256 : 1];   // (px, py, pz, total) of a parked ray
  __shared__ int parkSource[PARK ? 256 : 1];     // the thread (= pixel of the tile) it belongs to
  __shared__ int parkCount;
  if (PARK && threadIdx.x == 0) parkCount = 0;
  ITM_RS(unsigned long long* stamp = g_rayStamps + (size_t)((blockIdx.x * 4 + wave) & 8191) * 4; if (lane == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();)
  float2 mm = make_float2(0.0f, 0.0f);
  if constexpr (REDUCE) {
    reduce_own_cells(fuse, tx, ty, p.W, cellRange);
    __syncthreads();
    if (inside) mm = cellRange[(lx >> 3) + 2 * (ly >> 3)];
  } else {
    if (PARK) __syncthreads();
    if (inside) mm = range[(x >> 3) + (y >> 3) * p.W];  // floor(x/8) + floor(y/8)*W  (_CPU.cpp:174)
  }
  ITM_RS(if (lane == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();)
  // Tiles whose expected-depth interval is wide hold a silhouette: their rays are the long chains the launch waits for
  // (profiles/r2_raycast_timeline.txt), and while the other ~1 000 tiles are still resident they share each SIMD's issue slots with
  // four other waves.  Raised wave priority lets them issue first: 48.2 -> 46.1 us in frame (config 2; a threshold of 0.03 m, i.e.
  // nearly every tile, gives nothing; delaying the other tiles by 3-14 us instead: no gain, config 5 +4..20 us).
  if (!DENSE && __any(inside && mm.y - mm.x > kSilhouetteSpread)) __builtin_amdgcn_s_setprio(kSilhouettePrio);
  // ---- phase 1: every ray of the tile; rays that turn out to be crossing empty space are parked ----
  bool parked = false;
  if (inside) {
    const float4 r = march_ray<VX, DENSE, DENSE ? kDenseLookahead : 0, PARK>(x, y, vol, p, mm, nullptr, parked);
    if (!parked) out[x + y * p.W] = r;
    else if constexpr (PARK) {
      const int slot = atomicAdd(&parkCount, 1);
      parkState[slot] = r;
      parkSource[slot] = ly * 16 + lx;           // the pixel of the tile the ray belongs to
    }
  }
  ITM_RS(if (lane == 0) { stamp[2] = __builtin_amdgcn_s_memrealtime(); stamp[3] = 0; })
  if constexpr (PARK) {
    __syncthreads();
    // ---- phase 2: the parked rays, re-packed from lane 0 upwards ----
    const int n = parkCount;
    // (the parked rays dealt out to the four waves in turn instead of 64 per wave -- fewer rays, fewer distinct states per wave -- was
    // measured in round 6: BASELINE configs[1] 35.5 -> 37.9 us, configs[4] 112 -> 122.5 us; profiles/r6_notes.md)
    if ((int)threadIdx.x < n) {
      const float4 q = parkState[threadIdx.x];
      const int src = parkSource[threadIdx.x];
      const int qx = tx * 16 + (src & 15), qy = ty * 16 + (src >> 4);
      float2 m2;
      if constexpr (REDUCE) m2 = cellRange[((src & 15) >> 3) + 2 * (src >> 7)];
      else m2 = range[(qx >> 3) + (qy >> 3) * p.W];
      const RayResume rr{q.x, q.y, q.z, q.w};
      bool again;
      out[qx + qy * p.W] = march_ray<VX, DENSE, kParkedLookahead, false>(qx, qy, vol, p, m2, &rr, again);
    }
    ITM_RS(if (lane == 0) stamp[3] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)(wave == 0 ? n : 0) << 52);)
  }
}
#if ITM_EXP_RAYCAST_STAMPS
extern "C" int itm_debug_read_raycast_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rayStamps), (size_t)n * 8); }
#endif

int launch_raycast(const itm_scene* s, const float* invM, const float* intr, itm_render_state* rs, float4* dst, hipStream_t st, bool reduceRange, const AheadRequest* aheadIn) {
  RayParams p; make_ray_params(s, invM, intr, rs->w, rs->h, p);
  const VolumeView vol = make_volume(s);
  const int rayTiles = ((rs->w + 15) / 16) * ((rs->h + 15) / 16);
  const bool dense = s->cfg.indexType == ITM_INDEX_DENSE;
  RangeFuse fuse{rs->rangePartials, rs->counters, rs->projBuf, rs->range, (rs->w + 7) / 8, (rs->h + 7) / 8, s->cfg.maxRenderingBlocks};
  // two phases whenever the block directory is in use
  const bool park = !dense && vol.dirPtr != nullptr && !g_debug_single_pass_raycast;
  AheadRequest ahead;
  memset(&ahead, 0, sizeof ahead);
  int reqTiles = 0;
  if (aheadIn && !dense) { ahead = *aheadIn; ahead.rayTiles = rayTiles; reqTiles = ahead.reqTilesX * ((ahead.ap.H + 15) / 16); }
  const dim3 grid(rayTiles + reqTiles);
  KernelTimer tk(s, ITM_TK_RAYCAST, st);
  // the mirror's form picks the kernel (one form's address arithmetic per instance, see VolumeViewM); scenes without the block directory
  // (no parked rays) have no mirror either
  const int mirrorForm = (!vol.sdfMirror || !park) ? 0 : (vol.org.mMaxPages < 0 ? 1 : 2);
  int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    constexpr bool kMayMirror = VX::kShort || ITM_MIRROR_FLOAT_TYPES;
#define ITM_RC(RED, PRK, AHD, MIR) raycast_kernel<VX, false, RED, PRK, AHD, MIR><<<grid, 256, 0, st>>>(vol, rs->range, dst, p, fuse, ahead)
#define ITM_RC_MIRROR(RED, AHD) do { if (kMayMirror && mirrorForm == 1) ITM_RC(RED, true, AHD, (kMayMirror ? 1 : 0)); else if (kMayMirror && mirrorForm == 2) ITM_RC(RED, true, AHD, (kMayMirror ? 2 : 0)); else ITM_RC(RED, true, AHD, 0); } while (0)
    if (dense) raycast_kernel<VX, true, false, false><<<grid, 256, 0, st>>>(vol, rs->range, dst, p, fuse, ahead);
    else if (reqTiles) {
      if (park) { if (reduceRange) ITM_RC_MIRROR(true, true); else ITM_RC_MIRROR(false, true); }
      else if (reduceRange) ITM_RC(true, false, true, 0);
      else ITM_RC(false, false, true, 0);
    }
    else if (park) { if (reduceRange) ITM_RC_MIRROR(true, false); else ITM_RC_MIRROR(false, false); }
    else if (reduceRange) ITM_RC(true, false, false, 0);
    else ITM_RC(false, false, false, 0);
#undef ITM_RC_MIRROR
#undef ITM_RC
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}
// (the entry points of the other translation units know nothing of requests ahead)
int launch_raycast(const itm_scene* s, const float* invM, const float* intr, itm_render_state* rs, float4* dst, hipStream_t st, bool reduceRange) {
  return launch_raycast(s, invM, intr, rs, dst, st, reduceRange, nullptr);
}

// ---------------------------------------------------------------------------------------------
// ICP maps (normals from the ray-hit map)
// ---------------------------------------------------------------------------------------------
// Streaming outputs (11 MB per frame that nothing on the GPU re-reads soon) are stored with the non-temporal hint so that
// they do not wash the hash lines, occupancy words and voxel lines of the next kernels out of the 4 MB L2s.
__device__ inline void nt_store(float4* p, float4 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
__device__ inline void nt_store(uchar4* p, uchar4 v) { __builtin_nontemporal_store(*(unsigned int*)&v, (unsigned int*)p); }

// A workgroup = one 16x16 tile (wave = 16x4 pixels, as the ray tiles).  The normal of a pixel is made of hits one or two pixels away in
// x and y: the tile and its plus-shaped halo (16x16 + 4 strips of 2x16) are staged in LDS with one round of coalesced loads, instead of
// a centre load followed by four (and often four more) dependent gathers per lane.
__global__ void __launch_bounds__(256) icp_maps_kernel(const float4* __restrict__ rays, float4* __restrict__ points,
                                                       float4* __restrict__ normals, uchar4* __restrict__ image, RayParams p) {
  __shared__ float4 hits[20][20];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = lane & 15, ly = wave * 4 + (lane >> 4);
  const int x = blockIdx.x * 16 + lx, y = blockIdx.y * 16 + ly;
  const bool inside = x < p.W && y < p.H;
  if (inside) hits[2 + ly][2 + lx] = rays[x + y * p.W];
  if (threadIdx.x < 128) {
    const int strip = threadIdx.x >> 5, k = threadIdx.x & 31;      // 0 left, 1 right, 2 above, 3 below; 2 x 16 pixels each
    int hx, hy;
    if (strip < 2) { hx = strip == 0 ? (k & 1) : 18 + (k & 1); hy = 2 + (k >> 1); }
    else { hy = strip == 2 ? (k & 1) : 18 + (k & 1); hx = 2 + (k >> 1); }
    const int gx = (int)blockIdx.x * 16 + hx - 2, gy = (int)blockIdx.y * 16 + hy - 2;
    if (gx >= 0 && gx < p.W && gy >= 0 && gy < p.H) hits[hy][hx] = rays[gx + gy * p.W];
  }
  __syncthreads();
  if (!inside) return;
  const int loc = x + y * p.W;
  const float4 r = hits[2 + ly][2 + lx];
  bool found = r.w > 0.0f;
  float nx = 0, ny = 0, nz = 0, angle = 0;
  const int ox = (int)blockIdx.x * 16 - 2, oy = (int)blockIdx.y * 16 - 2;
  if (found) found = normal_from_hits_at([&](int qx, int qy) { return hits[qy - oy][qx - ox]; }, x, y, p.W, p.H, p.voxelSize, p.lx, p.ly, p.lz, nx, ny, nz, angle);
  if (found) {
    nt_store(&image[loc], grey_pixel(angle));
    nt_store(&points[loc], make_float4(r.x * p.voxelSize, r.y * p.voxelSize, r.z * p.voxelSize, 1.0f));
    nt_store(&normals[loc], make_float4(nx, ny, nz, 0.0f));
  } else {
    const float4 inv = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    nt_store(&points[loc], inv); nt_store(&normals[loc], inv); nt_store(&image[loc], make_uchar4(0, 0, 0, 0));
  }
}

// The ICP maps of a frame whose successor's block requests rode in the ray-cast launch (itm_process_frame_ahead): what the request
// launch of the successor would have initialised -- the range image and the rendering-block counters of its CreateExpectedDepths --
// is initialised here, after this frame's ray cast has read them.
__global__ void __launch_bounds__(256) icp_maps_init_next_kernel(const float4* __restrict__ rays, float4* __restrict__ points, float4* __restrict__ normals,
                                                                 uchar4* __restrict__ image, RayParams p, float2* __restrict__ range, RenderCounters* __restrict__ rcnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { rcnt->noRenderingBlocks = 0; rcnt->renderingBlocksAccepted = -1; }
  const int x = blockIdx.x * 16 + (lane & 15);
  const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
  if (x >= p.W || y >= p.H) return;
  const int loc = x + y * p.W;
  range[loc] = make_float2(999999.9f, 0.05f);
  const float4 r = rays[loc];
  bool found = r.w > 0.0f;
  float nx = 0, ny = 0, nz = 0, angle = 0;
  if (found) found = normal_from_hits(rays, x, y, p.W, p.H, p.voxelSize, p.lx, p.ly, p.lz, nx, ny, nz, angle);
  if (found) {
    nt_store(&image[loc], grey_pixel(angle));
    nt_store(&points[loc], make_float4(r.x * p.voxelSize, r.y * p.voxelSize, r.z * p.voxelSize, 1.0f));
    nt_store(&normals[loc], make_float4(nx, ny, nz, 0.0f));
  } else {
    const float4 inv = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    nt_store(&points[loc], inv); nt_store(&normals[loc], inv); nt_store(&image[loc], make_uchar4(0, 0, 0, 0));
  }
}

int prepare_request_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st, AllocParams& p, RequestArgs& ra, bool& lazy);

// `next` (with the scene as a mutable object): also issue the block requests of that view, in the tail of the ray-cast launch
int launch_icp_maps(const itm_scene* s, const itm_view* v, itm_render_state* rs, float4* points, float4* normals, hipStream_t st, bool reduceRange,
                    itm_scene* sceneForNext, const itm_view* next) {
  float invM[16];
  if (!invert4(v->M_d, invM)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  AheadRequest ahead;
  bool issueAhead = false;
  if (next && sceneForNext) {
    // (placing the cubes for the next view may move them: before this frame's ray cast, which then reads them at their new place)
    bool lazy = false;
    int rc = prepare_request_stage(sceneForNext, next, rs, st, ahead.ap, ahead.ra, lazy);
    if (rc) return rc;
    issueAhead = lazy;          // (otherwise the previous list needs its explicit mark first: the next frame issues its own requests)
    ahead.reqTilesX = (next->w + 15) / 16;
  }
  int rc = launch_raycast(s, invM, v->intr_d, rs, rs->raycast, st, reduceRange, issueAhead ? &ahead : nullptr);
  if (rc) return rc;
  RayParams p; make_ray_params(s, invM, v->intr_d, rs->w, rs->h, p);
  const dim3 grid((rs->w + 15) / 16, (rs->h + 15) / 16);
  KernelTimer tk(s, ITM_TK_ICP_MAPS, st);
  if (issueAhead) {
    icp_maps_init_next_kernel<<<grid, 256, 0, st>>>(rs->raycast, points, normals, rs->image, p, rs->range, rs->counters);
    rs->ahead.tableEpoch = sceneForNext->tableEpoch;
    sceneForNext->aheadRs = rs;
    rs->ahead.valid = true; rs->ahead.depth = next->depth; rs->ahead.w = next->w; rs->ahead.h = next->h; rs->ahead.lazy = true;
    memcpy(rs->ahead.M_d, next->M_d, 64); memcpy(rs->ahead.intr_d, next->intr_d, 16);
  } else {
    icp_maps_kernel<<<grid, 256, 0, st>>>(rs->raycast, points, normals, rs->image, p);
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

// ---------------------------------------------------------------------------------------------
// free-view rendering: SDF-gradient normals, colour lookup
// ---------------------------------------------------------------------------------------------
template <class VX, bool DENSE>
__global__ void __launch_bounds__(256) render_image_kernel(VolumeView vol, const float4* __restrict__ rays, uchar4* __restrict__ out, int type, RayParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (lane & 15);
  const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
  if (x >= p.W || y >= p.H) return;
  const int loc = x + y * p.W;
  const float4 r = rays[loc];
  bool found = r.w > 0;
  float nx = 0, ny = 0, nz = 0, angle = 0;
  if (found) found = normal_from_sdf<VX, DENSE>(vol, r.x, r.y, r.z, p, nx, ny, nz, angle);
  if (!found) { out[loc] = make_uchar4(0, 0, 0, 0); return; }
  if (type == ITM_RENDER_COLOUR_FROM_VOLUME) {
    const float4 c = colour_at<VX, DENSE>(vol, r.x, r.y, r.z);
    out[loc] = make_uchar4((unsigned char)(c.x * 255.0f), (unsigned char)(c.y * 255.0f), (unsigned char)(c.z * 255.0f), 255);
  } else if (type == ITM_RENDER_COLOUR_FROM_NORMAL) {
    // drawPixelNormal writes r,g,b only; the alpha byte keeps its previous value
    unsigned char* o = (unsigned char*)&out[loc];
    o[0] = (unsigned char)((0.3f + (-nx + 1.0f) * 0.35f) * 255.0f);
    o[1] = (unsigned char)((0.3f + (-ny + 1.0f) * 0.35f) * 255.0f);
    o[2] = (unsigned char)((0.3f + (-nz + 1.0f) * 0.35f) * 255.0f);
  } else {
    out[loc] = grey_pixel(angle);
  }
}

int launch_render_image(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, uchar4* out, int type, hipStream_t st) {
  float invM[16];
  if (!invert4(M, invM)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  int rc = launch_raycast(s, invM, intr, rs, rs->raycast, st);
  if (rc) return rc;
  RayParams p; make_ray_params(s, invM, intr, rs->w, rs->h, p);
  const VolumeView vol = make_volume(s);
  const dim3 grid((rs->w + 15) / 16, (rs->h + 15) / 16);
  const bool dense = s->cfg.indexType == ITM_INDEX_DENSE;
  const bool colour = (s->cfg.voxelType == ITM_VOXEL_S_RGB || s->cfg.voxelType == ITM_VOXEL_F_RGB);
  if (type == ITM_RENDER_COLOUR_FROM_VOLUME && !colour) type = ITM_RENDER_SHADED_GREYSCALE;  // _CPU.cpp:205-206
  if (type != ITM_RENDER_COLOUR_FROM_VOLUME && type != ITM_RENDER_COLOUR_FROM_NORMAL) type = ITM_RENDER_SHADED_GREYSCALE;
  rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    if (dense) render_image_kernel<VX, true><<<grid, 256, 0, st>>>(vol, rs->raycast, out, type, p);
    else render_image_kernel<VX, false><<<grid, 256, 0, st>>>(vol, rs->raycast, out, type, p);
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_debug_set(int key, int value) {
  if (key == ITM_DEBUG_FORCE_GLOBAL_RANGE_ATOMICS) { g_debug_force_global_range = value; return ITM_OK; }
  if (key == ITM_DEBUG_EXPLICIT_MARK_PREVIOUS) { g_debug_explicit_mark = value; return ITM_OK; }
  if (key == ITM_DEBUG_INTEGRATE_WORKGROUPS) { g_debug_integrate_wgs = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_FUSED_PROJECTION) { g_debug_no_fused_projection = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_DIRECTORY) { g_debug_no_directory = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_FUSED_RANGE_REDUCE) { g_debug_no_fused_range_reduce = value; return ITM_OK; }
  if (key == ITM_DEBUG_TWO_PASS_VISIBLE_LIST) { g_debug_two_pass_visible_list = value; return ITM_OK; }
  if (key == ITM_DEBUG_SINGLE_PASS_RAYCAST) { g_debug_single_pass_raycast = value; return ITM_OK; }
  if (key == ITM_DEBUG_DENSE_GROUP_CULL) { g_debug_dense_group_cull = value; return ITM_OK; }
  if (key == ITM_DEBUG_TRACKER_LAUNCH_PER_EVALUATION) { g_debug_tracker_launch_per_evaluation = value; return ITM_OK; }
  if (key == ITM_DEBUG_TRACKER_HOST_COMMAND) { g_debug_tracker_host_command = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_SDF_MIRROR) { g_debug_no_sdf_mirror = value; return ITM_OK; }
  if (key == ITM_DEBUG_SEPARATE_SWEEP) { g_debug_separate_sweep = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_SIDE_PROJECTION) { g_debug_no_side_projection = value; return ITM_OK; }
  if (key == ITM_DEBUG_DENSE_RANGE_REFILL) { g_debug_dense_range_refill = value; return ITM_OK; }
  if (key == ITM_DEBUG_DENSE_CLASSIFY) { g_debug_dense_classify = value; return ITM_OK; }
  if (key == ITM_DEBUG_TRACKER_SESSION_UNUSABLE) { g_debug_tracker_session_unusable = value; return ITM_OK; }
  if (key == ITM_DEBUG_DENSE_NO_STRIPS) { g_debug_dense_no_strips = value; return ITM_OK; }
  if (key == ITM_DEBUG_NO_DEFERRED_FUSION) { g_debug_no_deferred_fusion = value; return ITM_OK; }
  if (key == ITM_DEBUG_FORCE_LIST_STUCK) { g_debug_force_list_stuck = value; return ITM_OK; }
  if (key == ITM_DEBUG_EXCHANGE_DEVICE_COPY) { g_debug_exchange_device_copy = value; return ITM_OK; }
  if (key == ITM_DEBUG_EXCHANGE_CORRUPT_WORD) { g_debug_exchange_corrupt_word = value; return ITM_OK; }
  return set_error(ITM_ERR_INVALID, "unknown debug key");
}

int itm_find_surface(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream stream) {
  if (!s || !M || !intr || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  float invM[16];
  if (!invert4(M, invM)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  return launch_raycast(s, invM, intr, rs, rs->raycast, as_stream(stream));
}

int itm_render_image(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, uint8_t* out, int type, itm_stream stream) {
  if (!s || !M || !intr || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  return launch_render_image(s, M, intr, rs, out ? (uchar4*)out : rs->image, type, as_stream(stream));
}

// ITMDenseMapper::ProcessFrame + ITMTrackingController::Prepare (Engine/ITMMainEngine.cpp:123-126):
// the four engine calls on one stream with the range-image reset fused into the request kernel.
// (A two-stream variant -- request stage of frame k+1 beside the ray cast of frame k, range image
// beside integration -- was measured SLOWER on MI355X, 170 vs 157 us/frame: the four cross-stream
// event dependencies cost more than the ~25 us of overlap they buy.  Not kept.)
int itm_process_frame(itm_scene* s, const itm_view* v, itm_render_state* rs, float* points, float* normals, itm_stream stream) {
  return itm_process_frame_ahead(s, v, nullptr, rs, points, normals, stream);
}

int itm_process_frame_ahead(itm_scene* s, const itm_view* v, const itm_view* next, itm_render_state* rs, float* points, float* normals, itm_stream stream) {
  if (!s || !v || !rs || !points || !normals) return set_error(ITM_ERR_INVALID, "null argument");
  if (next && (!next->depth || next->w != rs->w || next->h != rs->h)) return set_error(ITM_ERR_INVALID, "next view / render state mismatch");
  if (!v->depth) return set_error(ITM_ERR_INVALID, "null depth image");
  if (rs->scene != s || v->w != rs->w || v->h != rs->h) return set_error(ITM_ERR_INVALID, "view / render state mismatch");
  hipStream_t st = as_stream(stream);
  int rc;
  if ((rc = enter_scene(s, rs))) return rc;
  const bool hashScene = s->cfg.indexType == ITM_INDEX_HASH;
  if (hashScene && (rc = launch_allocate(s, v, rs, false, true, st))) return rc;
  const bool fuse = hashScene && can_fuse_projection(s, rs);
  // too large to ride in the integration launch: on the render state's side stream, beside the integration
  int beside = 0;
  if (hashScene && !fuse && (beside = launch_projection_beside(s, v->M_d, v->intr_d, rs, st)) < 0) return beside;
  if ((rc = launch_integrate(s, v, rs, st, fuse))) return rc;
  if (beside) ITM_HIP(hipStreamWaitEvent(st, rs->projectionDone, 0));
  // with the projection done inside the integration launch, the ray-cast workgroups reduce the partial range images of their
  // own cells (raycast_kernel<.., REDUCE>); otherwise CreateExpectedDepths runs as its own launches
  const bool reduceInRaycast = (fuse || beside) && !g_debug_no_fused_range_reduce;
  if (!reduceInRaycast && (rc = launch_expected_depths(s, v->M_d, v->intr_d, rs, hashScene, st, fuse || beside))) return rc;
  const bool ahead = next && hashScene && !s->cfg.useSwapping;
  return launch_icp_maps(s, v, rs, (float4*)points, (float4*)normals, st, reduceInRaycast, ahead ? s : nullptr, ahead ? next : nullptr);
}

}  // extern "C"

#if ITM_EXP_WAVE_TIMING
extern "C" int itm_debug_read_wave_stats(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(itm::g_waveStats), (size_t)n * 8);
}
extern "C" int itm_debug_read_wave_trace(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(itm::g_waveTrace), (size_t)n * 8);
}
#endif
