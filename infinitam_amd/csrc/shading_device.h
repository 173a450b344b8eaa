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

// computeNormalAndAngle<useSmoothing = true> (DeviceAgnostic/ITMVisualisationEngine.h:191-254)
__device__ inline bool normal_from_hits(const float4* __restrict__ rays, int x, int y, int W, int H, float voxelSize,
                                        float lx, float ly, float lz, float& nx, float& ny, float& nz, float& angle) {
  if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) return false;
  float4 xp = rays[(x + 2) + y * W], yp = rays[x + (y + 2) * W];
  float4 xm = rays[(x - 2) + y * W], ym = rays[x + (y - 2) * W];
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
    xp = rays[(x + 1) + y * W]; yp = rays[x + (y + 1) * W];
    xm = rays[(x - 1) + y * W]; ym = rays[x + (y - 1) * W];
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

template <class VX, bool DENSE>
__device__ inline float raw_at(const VolumeView& vol, int x, int y, int z) {
  BlockCache c; bool f;   // computeSingleNormalFromSDF uses the uncached readVoxel overload
  return read_raw_sdf<VX, DENSE>(vol, x, y, z, f, c);
}

template <class VX, bool DENSE>
__device__ inline void sdf_gradient(const VolumeView& vol, float px, float py, float pz, float& gx, float& gy, float& gz) {
  const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
  const float cx = px - flx, cy = py - fly, cz = pz - flz;
  const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
  const float nx = 1.0f - cx, ny = 1.0f - cy, nz = 1.0f - cz;
#define R(dx, dy, dz) raw_at<VX, DENSE>(vol, ix + (dx), iy + (dy), iz + (dz))
  const float f000 = R(0, 0, 0), f100 = R(1, 0, 0), f010 = R(0, 1, 0), f110 = R(1, 1, 0);
  const float f001 = R(0, 0, 1), f101 = R(1, 0, 1), f011 = R(0, 1, 1), f111 = R(1, 1, 1);
  float p1, p2, v1, a, b, c, d;
  p1 = f000 * ny * nz + f010 * cy * nz + f001 * ny * cz + f011 * cy * cz;
  a = R(-1, 0, 0); b = R(-1, 1, 0); c = R(-1, 0, 1); d = R(-1, 1, 1);
  p2 = a * ny * nz + b * cy * nz + c * ny * cz + d * cy * cz;
  v1 = p1 * cx + p2 * nx;
  p1 = f100 * ny * nz + f110 * cy * nz + f101 * ny * cz + f111 * cy * cz;
  a = R(2, 0, 0); b = R(2, 1, 0); c = R(2, 0, 1); d = R(2, 1, 1);
  p2 = a * ny * nz + b * cy * nz + c * ny * cz + d * cy * cz;
  gx = VX::to_float(p1 * nx + p2 * cx - v1);
  p1 = f000 * nx * nz + f100 * cx * nz + f001 * nx * cz + f101 * cx * cz;
  a = R(0, -1, 0); b = R(1, -1, 0); c = R(0, -1, 1); d = R(1, -1, 1);
  p2 = a * nx * nz + b * cx * nz + c * nx * cz + d * cx * cz;
  v1 = p1 * cy + p2 * ny;
  p1 = f010 * nx * nz + f110 * cx * nz + f011 * nx * cz + f111 * cx * cz;
  a = R(0, 2, 0); b = R(1, 2, 0); c = R(0, 2, 1); d = R(1, 2, 1);
  p2 = a * nx * nz + b * cx * nz + c * nx * cz + d * cx * cz;
  gy = VX::to_float(p1 * ny + p2 * cy - v1);
  p1 = f000 * nx * ny + f100 * cx * ny + f010 * nx * cy + f110 * cx * cy;
  a = R(0, 0, -1); b = R(1, 0, -1); c = R(0, 1, -1); d = R(1, 1, -1);
  p2 = a * nx * ny + b * cx * ny + c * nx * cy + d * cx * cy;
  v1 = p1 * cz + p2 * nz;
  p1 = f001 * nx * ny + f101 * cx * ny + f011 * nx * cy + f111 * cx * cy;
  a = R(0, 0, 2); b = R(1, 0, 2); c = R(0, 1, 2); d = R(1, 1, 2);
  p2 = a * nx * ny + b * cx * ny + c * nx * cy + d * cx * cy;
  gz = VX::to_float(p1 * nz + p2 * cz - v1);
#undef R
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
