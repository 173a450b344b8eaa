// raycast_device.h -- device-side voxel lookup and ray marching shared by the visualisation kernels.
//
// Reference behaviour restated (operation order: SURVEY.md Appendix A.6-A.8):
//   pointToVoxelBlockPos / readVoxel (hash, dense)  DeviceAgnostic/ITMRepresentationAccess.h:12-20, :85-142
//   readFromSDF_float_uninterpolated / _interpolated  :144-185
//   castRay                                           DeviceAgnostic/ITMVisualisationEngine.h:92-158
#pragma once

#include "itm_types.h"

namespace itm {

struct VolumeView {
  const uint4* hash;   // hash entries (hash index only)
  const void* vba;     // voxel storage
  uint32_t mask;       // bucketNum - 1
  int bucketNum;
  int sx, sy, sz;      // dense size
  int ox, oy, oz;      // dense offset
};

// per-ray block cache: ITMVoxelBlockHash::IndexCache (Objects/ITMVoxelBlockHash.h:27-33)
struct BlockCache {
  int bx, by, bz;
  int base;
  __device__ BlockCache() : bx(0x7fffffff), by(0x7fffffff), bz(0x7fffffff), base(-1) {}
};

__device__ inline int floor_div8(int p) { return ((p < 0) ? p - 7 : p) / 8; }
__device__ inline float round_ref(float x) { return (x < 0) ? (x - 0.5f) : (x + 0.5f); }

// Linear voxel index of integer point (px,py,pz), or -1 when no voxel is stored there.
template <bool DENSE>
__device__ inline long long locate_voxel(const VolumeView& vol, int px, int py, int pz, BlockCache& cache) {
  if (DENSE) {
    const int qx = px - vol.ox, qy = py - vol.oy, qz = pz - vol.oz;
    if (qx < 0 || qx >= vol.sx || qy < 0 || qy >= vol.sy || qz < 0 || qz >= vol.sz) return -1;
    return (long long)(qx + qy * vol.sx + qz * vol.sx * vol.sy);
  } else {
    const int bx = floor_div8(px), by = floor_div8(py), bz = floor_div8(pz);
    const int lin = (px - bx * 8) + (py - by * 8) * 8 + (pz - bz * 8) * 64;
    if (bx == cache.bx && by == cache.by && bz == cache.bz) return (long long)cache.base + lin;
    int idx = hash_index(bx, by, bz, vol.mask);
    for (;;) {
      const HashEntry e = unpack_entry(vol.hash[idx]);
      if (e.px == bx && e.py == by && e.pz == bz && e.ptr >= 0) {
        cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = e.ptr * kBlockVoxels;
        return (long long)cache.base + lin;
      }
      if (e.offset < 1) break;
      idx = vol.bucketNum + e.offset - 1;
    }
    return -1;
  }
}

// raw (unconverted) sdf of the voxel at an integer point; the default voxel when absent
template <class VX, bool DENSE>
__device__ inline float read_raw_sdf(const VolumeView& vol, int px, int py, int pz, bool& found, BlockCache& cache) {
  const long long a = locate_voxel<DENSE>(vol, px, py, pz, cache);
  found = a >= 0;
  if (!found) return VX::kShort ? 32767.0f : 1.0f;
  return VX::load_raw_sdf(vol.vba, (size_t)a);
}

template <class VX, bool DENSE>
__device__ inline float sdf_nearest(const VolumeView& vol, float x, float y, float z, bool& found, BlockCache& cache) {
  return VX::to_float(read_raw_sdf<VX, DENSE>(vol, (int)round_ref(x), (int)round_ref(y), (int)round_ref(z), found, cache));
}

template <class VX, bool DENSE>
__device__ inline float sdf_trilinear(const VolumeView& vol, float x, float y, float z, bool& found, BlockCache& cache) {
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const float cx = x - fx, cy = y - fy, cz = z - fz;
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float v1, v2, r1, r2;
  v1 = read_raw_sdf<VX, DENSE>(vol, ix, iy, iz, found, cache);
  v2 = read_raw_sdf<VX, DENSE>(vol, ix + 1, iy, iz, found, cache);
  r1 = (1.0f - cx) * v1 + cx * v2;
  v1 = read_raw_sdf<VX, DENSE>(vol, ix, iy + 1, iz, found, cache);
  v2 = read_raw_sdf<VX, DENSE>(vol, ix + 1, iy + 1, iz, found, cache);
  r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v1 + cx * v2);
  v1 = read_raw_sdf<VX, DENSE>(vol, ix, iy, iz + 1, found, cache);
  v2 = read_raw_sdf<VX, DENSE>(vol, ix + 1, iy, iz + 1, found, cache);
  r2 = (1.0f - cx) * v1 + cx * v2;
  v1 = read_raw_sdf<VX, DENSE>(vol, ix, iy + 1, iz + 1, found, cache);
  v2 = read_raw_sdf<VX, DENSE>(vol, ix + 1, iy + 1, iz + 1, found, cache);
  r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v1 + cx * v2);
  found = true;
  return VX::to_float((1.0f - cz) * r1 + cz * r2);
}

struct RayParams {
  Mat4 invM;
  float ifx, ify, cx, cy;   // (1/fx, 1/fy, cx, cy)
  float oneOverVoxel, mu, voxelSize;
  float lx, ly, lz;         // light source = -(invM column 2)
  int W, H;
};

template <class VX, bool DENSE>
__device__ inline float4 cast_ray(int x, int y, const VolumeView& vol, const RayParams& p, float2 mm) {
  float sdf = 1.0f;
  const float stepScale = p.mu * p.oneOverVoxel;
  float pcz = mm.x;
  float pcx = pcz * (((float)x - p.cx) * p.ifx);
  float pcy = pcz * (((float)y - p.cy) * p.ify);
  float acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  float total = sqrtf(acc) * p.oneOverVoxel;
  Vec3 t = transform_point(p.invM, pcx, pcy, pcz);
  const float sx = t.x * p.oneOverVoxel, sy = t.y * p.oneOverVoxel, sz = t.z * p.oneOverVoxel;
  pcz = mm.y;
  pcx = pcz * (((float)x - p.cx) * p.ifx);
  pcy = pcz * (((float)y - p.cy) * p.ify);
  acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  const float totalMax = sqrtf(acc) * p.oneOverVoxel;
  t = transform_point(p.invM, pcx, pcy, pcz);
  float dx = t.x * p.oneOverVoxel - sx, dy = t.y * p.oneOverVoxel - sy, dz = t.z * p.oneOverVoxel - sz;
  const float dn = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= dn; dy *= dn; dz *= dn;
  float px = sx, py = sy, pz = sz;
  BlockCache cache;
  bool found;
  float step;
  while (total < totalMax) {
    sdf = sdf_nearest<VX, DENSE>(vol, px, py, pz, found, cache);
    if (!found) {
      step = (float)kBlockSide;
    } else {
      if ((sdf <= 0.1f) && (sdf >= -0.5f)) sdf = sdf_trilinear<VX, DENSE>(vol, px, py, pz, found, cache);
      if (sdf <= 0.0f) break;
      const float s = sdf * stepScale;
      step = (s < 1.0f) ? 1.0f : s;
    }
    px += step * dx; py += step * dy; pz += step * dz;
    total += step;
  }
  float w = 0.0f;
  if (sdf <= 0.0f) {
    step = sdf * stepScale;
    px += step * dx; py += step * dy; pz += step * dz;
    sdf = sdf_trilinear<VX, DENSE>(vol, px, py, pz, found, cache);
    step = sdf * stepScale;
    px += step * dx; py += step * dy; pz += step * dz;
    w = 1.0f;
  }
  return make_float4(px, py, pz, w);
}

}  // namespace itm
