// shading_device.h -- per-pixel shading helpers shared by the visualisation kernels.
//
// Reference behaviour restated:
//   drawPixelGrey / computeNormalAndAngle (both variants)  DeviceAgnostic/ITMVisualisationEngine.h:175-260
//   computeSingleNormalFromSDF / readFromSDF_color4u_interpolated  DeviceAgnostic/ITMRepresentationAccess.h:187-337
#pragma once

#include <cstring>

#include "itm_internal.h"
#include "raycast_device.h"

namespace itm {

static inline VolumeView make_volume(const itm_scene* s) {
  VolumeView v;
  v.hash = s->hash; v.vba = s->vba; v.headBits = s->headBits;
  v.dirPtr = g_debug_no_directory ? nullptr : s->dirPtr;
  v.sdfMirror = (g_debug_no_directory || g_debug_no_sdf_mirror) ? nullptr : s->sdfMirror;
  v.pageTable = s->org.mTable;
  v.org = s->org;
  v.mask = (uint32_t)s->cfg.bucketNum - 1u; v.bucketNum = s->cfg.bucketNum;
  v.sx = s->cfg.denseSize[0]; v.sy = s->cfg.denseSize[1]; v.sz = s->cfg.denseSize[2];
  v.ox = s->cfg.denseOffset[0]; v.oy = s->cfg.denseOffset[1]; v.oz = s->cfg.denseOffset[2];
  return v;
}

static inline void make_ray_params(const itm_scene* s, const float* invM, const float* intr, int W, int H, RayParams& p) {
  memcpy(p.invM.m, invM, 64);
  p.ifx = 1.0f / intr[0]; p.ify = 1.0f / intr[1]; p.cx = intr[2]; p.cy = intr[3];
  p.oneOverVoxel = 1.0f / s->prm.voxelSize;
  p.mu = s->prm.mu; p.voxelSize = s->prm.voxelSize;
  p.lx = -invM[8]; p.ly = -invM[9]; p.lz = -invM[10];
  p.W = W; p.H = H;
}

__device__ inline uchar4 grey_pixel(float angle) {  // drawPixelGrey
  const float o = (0.8f * angle + 0.2f) * 255.0f;
  const unsigned char g = (unsigned char)o;
  return make_uchar4(g, g, g, g);
}

// computeNormalAndAngle<useSmoothing = true> (DeviceAgnostic/ITMVisualisationEngine.h:191-254); `at(x, y)` returns the ray-cast result of a pixel
template <class AT>
__device__ inline bool normal_from_hits_at(AT&& at, int x, int y, int W, int H, float voxelSize,
                                           float lx, float ly, float lz, float& nx, float& ny, float& nz, float& angle) {
  if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) return false;
  float4 xp = at(x + 2, y), yp = at(x, y + 2);
  float4 xm = at(x - 2, y), ym = at(x, y - 2);
  float dxx = 0, dxy = 0, dxz = 0, dyx = 0, dyy = 0, dyz = 0;
  bool plus1 = false;
  if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) plus1 = true;
  else {
    dxx = xp.x - xm.x; dxy = xp.y - xm.y; dxz = xp.z - xm.z;
    dyx = yp.x - ym.x; dyy = yp.y - ym.y; dyz = yp.z - ym.z;
    const float a = dxx * dxx + dxy * dxy + dxz * dxz, b = dyx * dyx + dyy * dyy + dyz * dyz;
    const float l = (a < b) ? b : a;
    if (l * voxelSize * voxelSize > (0.15f * 0.15f)) plus1 = true;
  }
  if (plus1) {
    xp = at(x + 1, y); yp = at(x, y + 1);
    xm = at(x - 1, y); ym = at(x, y - 1);
    dxx = xp.x - xm.x; dxy = xp.y - xm.y; dxz = xp.z - xm.z;
    dyx = yp.x - ym.x; dyy = yp.y - ym.y; dyz = yp.z - ym.z;
    if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) return false;
  }
  nx = -(dxy * dyz - dxz * dyy);
  ny = -(dxz * dyx - dxx * dyz);
  nz = -(dxx * dyy - dxy * dyx);
  const float sc = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
  nx *= sc; ny *= sc; nz *= sc;
  angle = nx * lx + ny * ly + nz * lz;
  return angle > 0.0f;
}
__device__ inline bool normal_from_hits(const float4* __restrict__ rays, int x, int y, int W, int H, float voxelSize,
                                        float lx, float ly, float lz, float& nx, float& ny, float& nz, float& angle) {
  return normal_from_hits_at([&](int qx, int qy) { return rays[qx + qy * W]; }, x, y, W, H, voxelSize, lx, ly, lz, nx, ny, nz, angle);
}

template <class VX, bool DENSE>
__device__ inline float raw_at(const VolumeView& vol, int x, int y, int z) {
  BlockCache c; bool f;   // computeSingleNormalFromSDF uses the uncached readVoxel overload
  return read_raw_sdf<VX, DENSE>(vol, x, y, z, f, c);
}

// One component of the sdf gradient at a fractional position (computeSingleNormalFromSDF, DeviceAgnostic/ITMRepresentationAccess.h:
// 187-252): along AXIS the volume is sampled on the four integer planes k = -1, 0, 1, 2 around the position, each plane blended
// bilinearly over the other two axes (u, v: the remaining axes in x < y < z order); the component is the difference between the
// linear blends of planes (1, 2) and of planes (0, -1).  The products and sums are the reference's, term for term:
//   plane = s00 (1-fu)(1-fv) + s10 fu (1-fv) + s01 (1-fu) fv + s11 fu fv,   lower = plane0 fa + plane-1 (1-fa),   upper = plane1 (1-fa) + plane2 fa
template <class VX, bool DENSE, int AXIS>
__device__ inline float gradient_component(const VolumeView& vol, int ix, int iy, int iz, float fa, float fu, float fv) {
  auto sample = [&](int k, int u, int v) {
    const int dx = (AXIS == 0) ? k : u;
    const int dy = (AXIS == 1) ? k : (AXIS == 0 ? u : v);
    const int dz = (AXIS == 2) ? k : v;
    return raw_at<VX, DENSE>(vol, ix + dx, iy + dy, iz + dz);
  };
  const float gu = 1.0f - fu, gv = 1.0f - fv, ga = 1.0f - fa;
  float plane[4];
#pragma unroll
  for (int k = -1; k <= 2; ++k)
    plane[k + 1] = sample(k, 0, 0) * gu * gv + sample(k, 1, 0) * fu * gv + sample(k, 0, 1) * gu * fv + sample(k, 1, 1) * fu * fv;
  const float lower = plane[1] * fa + plane[0] * ga;
  return VX::to_float(plane[2] * ga + plane[3] * fa - lower);
}

template <class VX, bool DENSE>
__device__ inline void sdf_gradient(const VolumeView& vol, float px, float py, float pz, float& gx, float& gy, float& gz) {
  const float bx = floorf(px), by = floorf(py), bz = floorf(pz);
  const float fx = px - bx, fy = py - by, fz = pz - bz;
  const int ix = (int)bx, iy = (int)by, iz = (int)bz;
  gx = gradient_component<VX, DENSE, 0>(vol, ix, iy, iz, fx, fy, fz);
  gy = gradient_component<VX, DENSE, 1>(vol, ix, iy, iz, fy, fx, fz);
  gz = gradient_component<VX, DENSE, 2>(vol, ix, iy, iz, fz, fx, fy);
}

template <class VX, bool DENSE>
__device__ inline bool normal_from_sdf(const VolumeView& vol, float px, float py, float pz, const RayParams& p,
                                       float& nx, float& ny, float& nz, float& angle) {
  sdf_gradient<VX, DENSE>(vol, px, py, pz, nx, ny, nz);
  const float sc = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
  nx *= sc; ny *= sc; nz *= sc;
  angle = nx * p.lx + ny * p.ly + nz * p.lz;
  return angle > 0.0f;
}

// readFromSDF_color4u_interpolated; returns (r,g,b)/255 and w = 1
template <class VX, bool DENSE>
__device__ inline float4 colour_at(const VolumeView& vol, float px, float py, float pz) {
  if constexpr (!VX::kColor) {
    return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  } else {
    BlockCache cache;
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    const float cx = px - flx, cy = py - fly, cz = pz - flz;
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    float r[3] = {0.0f, 0.0f, 0.0f};
    auto add = [&](int dx, int dy, int dz, float wgt) {
      const long long a = locate_voxel<DENSE>(vol, ix + dx, iy + dy, iz + dz, cache);
      int c[3] = {0, 0, 0}, wc = 0;
      if (a >= 0) VX::get_color(VX::load(vol.vba, (size_t)a), c, wc);
      r[0] += wgt * (float)c[0]; r[1] += wgt * (float)c[1]; r[2] += wgt * (float)c[2];
    };
    add(0, 0, 0, (1.0f - cx) * (1.0f - cy) * (1.0f - cz));
    add(1, 0, 0, (cx) * (1.0f - cy) * (1.0f - cz));
    add(0, 1, 0, (1.0f - cx) * (cy) * (1.0f - cz));
    add(1, 1, 0, (cx) * (cy) * (1.0f - cz));
    add(0, 0, 1, (1.0f - cx) * (1.0f - cy) * cz);
    add(1, 0, 1, (cx) * (1.0f - cy) * cz);
    add(0, 1, 1, (1.0f - cx) * (cy)*cz);
    add(1, 1, 1, (cx) * (cy)*cz);
    return make_float4(r[0] / 255.0f, r[1] / 255.0f, r[2] / 255.0f, 255.0f / 255.0f);
  }
}


}  // namespace itm
