// visualise_aux.hip -- the approximate-raycast and colour-tracker entry points of the
// visualisation engine.  Off by default in the reference (useApproximateRaycast = false,
// TRACKER_COLOR unused; Utils/ITMLibSettings.cpp:38-44) but part of the engine interface.
//
// Reference behaviour:
//   ForwardRender_common      DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:289-354
//   forwardProjectPixel       DeviceAgnostic/ITMVisualisationEngine.h:160-173
//   processPixelForwardRender DeviceAgnostic/ITMVisualisationEngine.h:351-366
//   CreatePointCloud_common   ITMVisualisationEngine_CPU.cpp:242-264, RenderPointCloud :424-462
//
// The reference loops are sequential (last writer in raster order wins the forward projection;
// lists are filled in raster order).  Here the winner is chosen with atomicMax over the source
// pixel index and lists are produced by ordered (scan-based) compaction, which yields the same
// results without serialising.
#include "itm_internal.h"
#include "shading_device.h"
#include "wave_utils.h"

namespace itm {

struct FwdParams {
  Mat4 M;
  float fx, fy, cx, cy;
  float voxelSize;
  int W, H;
};

__global__ void __launch_bounds__(256) fwd_clear_kernel(int32_t* __restrict__ winner, int32_t* __restrict__ chunkCnt, int n, int nChunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) winner[i] = -1;
  if (i < nChunks) chunkCnt[i] = 0;
}

// forwardProjectPixel: the target pixel of every previous ray hit; highest source index wins
__global__ void __launch_bounds__(256) fwd_scatter_kernel(const float4* __restrict__ rays, int32_t* __restrict__ winner, FwdParams p) {
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= p.W * p.H) return;
  const float4 px = rays[loc];
  const Vec3 q = transform_point(p.M, px.x * p.voxelSize, px.y * p.voxelSize, px.z * p.voxelSize);
  const float u = p.fx * q.x / q.z + p.cx;
  const float v = p.fy * q.y / q.z + p.cy;
  if ((u < 0) || (u > p.W - 1) || (v < 0) || (v > p.H - 1)) return;
  if (u != u || v != v) return;  // 0/0: the reference converts NaN to int here (undefined); skipped
  const int t = (int)(u + 0.5f) + (int)(v + 0.5f) * p.W;
  if (t >= 0) atomicMax(&winner[t], loc);
}

// gathers the winners into forwardProjection and flags the holes that need a fresh ray
__global__ void __launch_bounds__(256) fwd_gather_flag_kernel(const float4* __restrict__ rays, int32_t* __restrict__ winnerThenFlag,
                                                              float4* __restrict__ fwd, const float2* __restrict__ range,
                                                              const float* __restrict__ depth, int32_t* __restrict__ chunkCnt, int W, int H) {
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= W * H) return;
  const int s = winnerThenFlag[loc];
  const float4 fp = (s >= 0) ? rays[s] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  fwd[loc] = fp;
  const int y = loc / W, x = loc - y * W;
  const float2 mm = range[(x >> 3) + (y >> 3) * W];
  const float d = depth[loc];
  const int flag = ((fp.w <= 0) && ((fp.x == 0 && fp.y == 0 && fp.z == 0) || (d >= 0)) && (mm.x < mm.y)) ? 1 : 0;
  winnerThenFlag[loc] = flag;
  if (flag) atomicAdd(&chunkCnt[loc / kSweepChunk], 1);
}

// ordered compaction of flagged pixel indices (raster order)
__global__ void __launch_bounds__(256) compact_pixels_kernel(const int32_t* __restrict__ flags, const int32_t* __restrict__ chunkCnt,
                                                             int nChunks, int n, int32_t* __restrict__ outIdx, int32_t* __restrict__ totalOut) {
  __shared__ int lds[8];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  int b = 0, all = 0;
  for (int j = tid; j < nChunks; j += 256) { const int c = chunkCnt[j]; all += c; if (j < chunk) b += c; }
  const int base = block_reduce_sum<4>(b, lds);
  if (chunk == 0) {
    const int total = block_reduce_sum<4>(all, lds + 4);
    if (tid == 0) *totalOut = total;
  }
  const int i0 = chunk * kSweepChunk + tid * 8;
  int f[8], cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { f[k] = (i0 + k < n) ? flags[i0 + k] : 0; cnt += f[k]; }
  int tot;
  int pos = base + block_exclusive_scan<4>(cnt, lds, &tot);
#pragma unroll
  for (int k = 0; k < 8; ++k) if (f[k]) outIdx[pos++] = i0 + k;
}

template <class VX, bool DENSE>
__global__ void __launch_bounds__(256) fwd_raycast_missing_kernel(VolumeView vol, const int32_t* __restrict__ missing,
                                                                  const RenderCounters* __restrict__ rc, const float2* __restrict__ range,
                                                                  float4* __restrict__ fwd, RayParams p) {
  const int n = rc->noFwdProjMissingPoints;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int loc = missing[i];
    const int y = loc / p.W, x = loc - y * p.W;
    fwd[loc] = cast_ray<VX, DENSE>(x, y, vol, p, range[(x >> 3) + (y >> 3) * p.W]);
  }
}

__global__ void __launch_bounds__(256) fwd_shade_kernel(const float4* __restrict__ fwd, uchar4* __restrict__ image, RayParams p) {
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= p.W * p.H) return;
  const int y = loc / p.W, x = loc - y * p.W;
  bool found = fwd[loc].w > 0.0f;
  float angle = 0;
  if (found) { float nx, ny, nz; found = normal_from_hits(fwd, x, y, p.W, p.H, p.voxelSize, p.lx, p.ly, p.lz, nx, ny, nz, angle); }
  image[loc] = found ? grey_pixel(angle) : make_uchar4(0, 0, 0, 0);
}

int launch_forward_render(const itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st) {
  const int W = rs->w, H = rs->h, P = W * H;
  const int nChunks = (P + kSweepChunk - 1) / kSweepChunk;
  float invM[16];
  if (!invert4(v->M_d, invM)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  FwdParams fp;
  memcpy(fp.M.m, v->M_d, 64);
  fp.fx = v->intr_d[0]; fp.fy = v->intr_d[1]; fp.cx = v->intr_d[2]; fp.cy = v->intr_d[3];
  fp.voxelSize = s->prm.voxelSize; fp.W = W; fp.H = H;
  RayParams rp; make_ray_params(s, invM, v->intr_d, W, H, rp);
  const VolumeView vol = make_volume(s);
  const int blocks = (P + 255) / 256;
  fwd_clear_kernel<<<blocks, 256, 0, st>>>(rs->pixScratch, rs->pixChunk, P, nChunks);
  fwd_scatter_kernel<<<blocks, 256, 0, st>>>(rs->raycast, rs->pixScratch, fp);
  fwd_gather_flag_kernel<<<blocks, 256, 0, st>>>(rs->raycast, rs->pixScratch, rs->fwdProj, rs->range, v->depth, rs->pixChunk, W, H);
  compact_pixels_kernel<<<nChunks, 256, 0, st>>>(rs->pixScratch, rs->pixChunk, nChunks, P, rs->missing, &rs->counters->noFwdProjMissingPoints);
  const bool dense = s->cfg.indexType == ITM_INDEX_DENSE;
  int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    if (dense) fwd_raycast_missing_kernel<VX, true><<<1024, 256, 0, st>>>(vol, rs->missing, rs->counters, rs->range, rs->fwdProj, rp);
    else fwd_raycast_missing_kernel<VX, false><<<1024, 256, 0, st>>>(vol, rs->missing, rs->counters, rs->range, rs->fwdProj, rp);
    return ITM_OK;
  });
  if (rc) return rc;
  fwd_shade_kernel<<<blocks, 256, 0, st>>>(rs->fwdProj, rs->image, rp);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

// ---------------------------------------------------------------------------------------------
// point cloud for the colour tracker
// ---------------------------------------------------------------------------------------------
// pass 1: shade every pixel, flag the pixels that enter the cloud
template <class VX, bool DENSE>
__global__ void __launch_bounds__(256) pc_flag_kernel(VolumeView vol, const float4* __restrict__ rays, uchar4* __restrict__ image,
                                                      int32_t* __restrict__ flags, int32_t* __restrict__ chunkCnt, int skipPoints, RayParams p) {
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= p.W * p.H) return;
  const int y = loc / p.W, x = loc - y * p.W;
  const float4 r = rays[loc];
  bool found = r.w > 0;
  float angle = 0;
  if (found) { float nx, ny, nz; found = normal_from_sdf<VX, DENSE>(vol, r.x, r.y, r.z, p, nx, ny, nz, angle); }
  image[loc] = found ? grey_pixel(angle) : make_uchar4(0, 0, 0, 0);
  if (skipPoints && ((x % 2 == 0) || (y % 2 == 0))) found = false;
  flags[loc] = found ? 1 : 0;
  if (found) atomicAdd(&chunkCnt[loc / kSweepChunk], 1);
}

// pass 2: ordered write of (location, colour) for flagged pixels
template <class VX, bool DENSE>
__global__ void __launch_bounds__(256) pc_write_kernel(VolumeView vol, const float4* __restrict__ rays, const int32_t* __restrict__ flags,
                                                       const int32_t* __restrict__ chunkCnt, int nChunks, float4* __restrict__ locations,
                                                       float4* __restrict__ colours, RenderCounters* __restrict__ rc, RayParams p) {
  __shared__ int lds[8];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int n = p.W * p.H;
  int b = 0, all = 0;
  for (int j = tid; j < nChunks; j += 256) { const int c = chunkCnt[j]; all += c; if (j < chunk) b += c; }
  const int base = block_reduce_sum<4>(b, lds);
  if (chunk == 0) {
    const int total = block_reduce_sum<4>(all, lds + 4);
    if (tid == 0) rc->noTotalPoints = total;
  }
  const int i0 = chunk * kSweepChunk + tid * 8;
  int f[8], cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { f[k] = (i0 + k < n) ? flags[i0 + k] : 0; cnt += f[k]; }
  int tot;
  int pos = base + block_exclusive_scan<4>(cnt, lds, &tot);
  for (int k = 0; k < 8; ++k) {
    if (!f[k]) continue;
    const float4 r = rays[i0 + k];
    float4 c = colour_at<VX, DENSE>(vol, r.x, r.y, r.z);
    if (c.w > 0.0f) { c.x /= c.w; c.y /= c.w; c.z /= c.w; c.w = 1.0f; }
    colours[pos] = c;
    locations[pos] = make_float4(r.x * p.voxelSize, r.y * p.voxelSize, r.z * p.voxelSize, 1.0f);
    ++pos;
  }
}

int launch_point_cloud(const itm_scene* s, const itm_view* v, itm_render_state* rs, bool skip, float4* loc, float4* col, hipStream_t st) {
  const int W = rs->w, H = rs->h, P = W * H;
  const int nChunks = (P + kSweepChunk - 1) / kSweepChunk;
  float invMd[16], invM[16];
  if (!invert4(v->M_d, invMd)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  matmul4(invMd, v->rgb_to_depth, invM);  // pose_d->GetInvM() * calib  (_CPU.cpp:247)
  int rc = launch_raycast(s, invM, v->intr_rgb, rs, rs->raycast, st);
  if (rc) return rc;
  RayParams rp; make_ray_params(s, invM, v->intr_rgb, W, H, rp);
  const VolumeView vol = make_volume(s);
  const int blocks = (P + 255) / 256;
  const bool dense = s->cfg.indexType == ITM_INDEX_DENSE;
  fwd_clear_kernel<<<blocks, 256, 0, st>>>(rs->pixScratch, rs->pixChunk, P, nChunks);
  rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    if (dense) {
      pc_flag_kernel<VX, true><<<blocks, 256, 0, st>>>(vol, rs->raycast, rs->image, rs->pixScratch, rs->pixChunk, skip ? 1 : 0, rp);
      pc_write_kernel<VX, true><<<nChunks, 256, 0, st>>>(vol, rs->raycast, rs->pixScratch, rs->pixChunk, nChunks, loc, col, rs->counters, rp);
    } else {
      pc_flag_kernel<VX, false><<<blocks, 256, 0, st>>>(vol, rs->raycast, rs->image, rs->pixScratch, rs->pixChunk, skip ? 1 : 0, rp);
      pc_write_kernel<VX, false><<<nChunks, 256, 0, st>>>(vol, rs->raycast, rs->pixScratch, rs->pixChunk, nChunks, loc, col, rs->counters, rp);
    }
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_forward_render(const itm_scene* s, const itm_view* v, itm_render_state* rs, itm_stream stream) {
  if (!s || !v || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (!v->depth) return set_error(ITM_ERR_INVALID, "null depth image");
  if (rs->scene != s || v->w != rs->w || v->h != rs->h) return set_error(ITM_ERR_INVALID, "view / render state mismatch");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  return launch_forward_render(s, v, rs, as_stream(stream));
}

int itm_create_point_cloud(const itm_scene* s, const itm_view* v, itm_render_state* rs, int skipPoints, float* locations, float* colours, itm_stream stream) {
  if (!s || !v || !rs || !locations || !colours) return set_error(ITM_ERR_INVALID, "null argument");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  return launch_point_cloud(s, v, rs, skipPoints != 0, (float4*)locations, (float4*)colours, as_stream(stream));
}

}  // extern "C"
