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
  const uint32_t* headBits;  // occupancy bitmap of the ordered buckets (hash index only)
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
    // both loads are issued together; the 16-byte entry is only consumed when the bucket is occupied
    const uint32_t word = vol.headBits[idx >> 5];
    HashEntry e = unpack_entry(vol.hash[idx]);
    if (!((word >> (idx & 31)) & 1u)) return -1;
    for (;;) {
      if (e.px == bx && e.py == by && e.pz == bz && e.ptr >= 0) {
        cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = e.ptr * kBlockVoxels;
        return (long long)cache.base + lin;
      }
      if (e.offset < 1) break;
      e = unpack_entry(vol.hash[vol.bucketNum + e.offset - 1]);
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

// Walks the excess chain starting from an already loaded head entry; block base or -1.
__device__ inline int resolve_block(const VolumeView& vol, HashEntry e, int bx, int by, int bz) {
  for (;;) {
    if (e.px == bx && e.py == by && e.pz == bz && e.ptr >= 0) return e.ptr * kBlockVoxels;
    if (e.offset < 1) return -1;
    e = unpack_entry(vol.hash[vol.bucketNum + e.offset - 1]);
  }
}

// Trilinear read (reference: eight readVoxel calls in the order 000 100 | 010 110 | 001 101 | 011 111,
// blended on the raw values).  The voxel values do not depend on the order in which blocks are
// looked up (the reference's IndexCache only short-cuts the probe), so for the hash index the
// lookups are restructured for memory-level parallelism: the up-to-8 distinct blocks touched by
// the 2x2x2 neighbourhood are probed with independent loads issued back to back, then the eight
// voxel loads are issued back to back.  The arithmetic on the values is unchanged.
template <class VX, bool DENSE>
__device__ inline float sdf_trilinear(const VolumeView& vol, float x, float y, float z, bool& found, BlockCache& cache) {
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const float cx = x - fx, cy = y - fy, cz = z - fz;
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float v[8];
  if (DENSE) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = read_raw_sdf<VX, DENSE>(vol, ix + (c & 1), iy + ((c >> 1) & 1), iz + (c >> 2), found, cache);
  } else {
    const int bx = floor_div8(ix), by = floor_div8(iy), bz = floor_div8(iz);
    const int lx = ix - bx * 8, ly = iy - by * 8, lz = iz - bz * 8;
    // bit k set: the +1 neighbour along axis k lies in the next block
    const int cross = (lx == 7 ? 1 : 0) | (ly == 7 ? 2 : 0) | (lz == 7 ? 4 : 0);
    const bool cached = (bx == cache.bx && by == cache.by && bz == cache.bz);
    HashEntry head[8];
    bool need[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      need[s] = ((s & ~cross) == 0) && !(s == 0 && cached);
      if (need[s]) head[s] = unpack_entry(vol.hash[hash_index(bx + (s & 1), by + ((s >> 1) & 1), bz + (s >> 2), vol.mask)]);
    }
    int base[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      base[s] = -1;
      if (need[s]) base[s] = resolve_block(vol, head[s], bx + (s & 1), by + ((s >> 1) & 1), bz + (s >> 2));
    }
    if (cached) base[0] = cache.base;
    else if (base[0] >= 0) { cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = base[0]; }
    const float dflt = VX::kShort ? 32767.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int b = base[c & cross];
      const int off = ((lx + (c & 1)) & 7) + ((ly + ((c >> 1) & 1)) & 7) * 8 + ((lz + (c >> 2)) & 7) * 64;
      v[c] = (b >= 0) ? VX::load_raw_sdf(vol.vba, (size_t)(b + off)) : dflt;
    }
  }
  float r1, r2;
  r1 = (1.0f - cx) * v[0] + cx * v[1];
  r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
  r2 = (1.0f - cx) * v[4] + cx * v[5];
  r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
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
      if (!DENSE) {
        // Empty-space skipping.  While blocks are missing the march is pure arithmetic
        // (pt += 8*dir, total += 8, exactly as the reference computes it), so the next kLook
        // positions are known in advance: their occupancy bits are fetched with independent loads
        // and every leading position whose bucket is provably empty is stepped over at once.
        px += step * dx; py += step * dy; pz += step * dz;
        total += step;
        constexpr int kLook = 4;
        float qx[kLook + 1], qy[kLook + 1], qz[kLook + 1], qt[kLook + 1];
        bool empty[kLook];
        qx[0] = px; qy[0] = py; qz[0] = pz; qt[0] = total;
#pragma unroll
        for (int j = 0; j < kLook; ++j) {
          const int vx = (int)round_ref(qx[j]), vy = (int)round_ref(qy[j]), vz = (int)round_ref(qz[j]);
          const int h = hash_index(floor_div8(vx), floor_div8(vy), floor_div8(vz), vol.mask);
          empty[j] = !((vol.headBits[h >> 5] >> (h & 31)) & 1u);
          qx[j + 1] = qx[j] + step * dx; qy[j + 1] = qy[j] + step * dy; qz[j + 1] = qz[j] + step * dz;
          qt[j + 1] = qt[j] + step;
        }
        int adv = 0;
#pragma unroll
        for (int j = 0; j < kLook; ++j) {
          if (adv == j && qt[j] < totalMax && empty[j]) adv = j + 1;
        }
#pragma unroll
        for (int j = 1; j <= kLook; ++j) if (adv == j) { px = qx[j]; py = qy[j]; pz = qz[j]; total = qt[j]; }
        continue;
      }
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
