// raycast_device.h -- device-side voxel lookup and ray marching shared by the visualisation kernels.
//
// Reference behaviour restated (operation order: SURVEY.md Appendix A.6-A.8):
//   pointToVoxelBlockPos / readVoxel (hash, dense)  DeviceAgnostic/ITMRepresentationAccess.h:12-20, :85-142
//   readFromSDF_float_uninterpolated / _interpolated  :144-185
//   castRay                                           DeviceAgnostic/ITMVisualisationEngine.h:92-158
#pragma once

#include "itm_types.h"

#ifndef ITM_RAY_SAMEBLOCK_FAST
#define ITM_RAY_SAMEBLOCK_FAST 0
#endif
#ifndef ITM_RAY_PREDICT_BAND
#define ITM_RAY_PREDICT_BAND 0
#endif
#ifndef ITM_EXP_SIMPLE_TRILINEAR
#define ITM_EXP_SIMPLE_TRILINEAR 0
#endif
#ifndef ITM_EXP_MAXITER
#define ITM_EXP_MAXITER 0
#endif
#ifndef ITM_RAY_FUSED_FETCH
#define ITM_RAY_FUSED_FETCH 0
#endif
#ifndef ITM_RAY_BITMAP_GUARD
#define ITM_RAY_BITMAP_GUARD 1  // consult the occupancy bitmap before fetching a hash entry (single-voxel lookups)
#endif

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
// ROUND(x) = x<0 ? x-0.5 : x+0.5 (ORUtils/MathUtils.h:21-23); after the (int) truncation this equals
// x + copysign(0.5, x) for every x (they only differ for x = -0.0: -0.5 vs +0.5, both truncate to 0).
__device__ inline float round_ref(float x) { return x + __builtin_copysignf(0.5f, x); }

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
#if ITM_RAY_BITMAP_GUARD
    // the 16-byte entry is only fetched when the occupancy bit says the bucket is in use
    if (!((vol.headBits[idx >> 5] >> (idx & 31)) & 1u)) return -1;
#endif
    HashEntry e = unpack_entry(vol.hash[idx]);
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

// The 2x2x2 voxel neighbourhood of floor(p): raw sdf values in the reference's read order
// 000 100 | 010 110 | 001 101 | 011 111 (default value where no block is allocated).
//
// The values of a trilinear read do not depend on the order in which blocks are looked up (the
// reference's IndexCache only short-cuts the probe), so for the hash index the lookups are
// restructured for memory-level parallelism: the up-to-8 distinct blocks the neighbourhood touches
// are probed with independent loads issued back to back, then the eight voxel loads are issued
// back to back.  The nearest voxel of p (ROUND per axis) is always one of these eight corners, so
// one fetch serves both the nearest-neighbour read and the trilinear read of a ray step.
template <class VX, bool DENSE>
struct Corners {
  float v[8];      // raw sdf per corner
  bool present[8]; // block allocated / inside the dense volume
  float cx, cy, cz;  // fractional position
  int ix, iy, iz;    // floor(p)

  __device__ inline void fetch(const VolumeView& vol, float x, float y, float z, BlockCache& cache) {
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    cx = x - fx; cy = y - fy; cz = z - fz;
    ix = (int)fx; iy = (int)fy; iz = (int)fz;
    const float dflt = VX::kShort ? 32767.0f : 1.0f;
    if (DENSE) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const long long a = locate_voxel<true>(vol, ix + (c & 1), iy + ((c >> 1) & 1), iz + (c >> 2), cache);
        present[c] = a >= 0;
        v[c] = present[c] ? VX::load_raw_sdf(vol.vba, (size_t)a) : dflt;
      }
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
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int b = base[c & cross];
        const int off = ((lx + (c & 1)) & 7) + ((ly + ((c >> 1) & 1)) & 7) * 8 + ((lz + (c >> 2)) & 7) * 64;
        present[c] = b >= 0;
        v[c] = present[c] ? VX::load_raw_sdf(vol.vba, (size_t)(b + off)) : dflt;
      }
    }
  }

  // readFromSDF_float_interpolated on the fetched values (blend on raw values, then convert)
  __device__ inline float trilinear() const {
    float r1, r2;
    r1 = (1.0f - cx) * v[0] + cx * v[1];
    r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
    r2 = (1.0f - cx) * v[4] + cx * v[5];
    r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
    return VX::to_float((1.0f - cz) * r1 + cz * r2);
  }

  // readFromSDF_float_uninterpolated at the same point: the voxel at ROUND(p) is corner
  // (ROUND(p) - floor(p)) in {0,1}^3
  __device__ inline float nearest(float x, float y, float z, bool& found) const {
    const int c = ((int)round_ref(x) - ix) | (((int)round_ref(y) - iy) << 1) | (((int)round_ref(z) - iz) << 2);
    float val = v[0]; bool pr = present[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) if (c == k) { val = v[k]; pr = present[k]; }
    found = pr;
    return VX::to_float(val);
  }
};

template <class VX, bool DENSE>
__device__ inline float sdf_trilinear(const VolumeView& vol, float x, float y, float z, bool& found, BlockCache& cache) {
#if ITM_EXP_SIMPLE_TRILINEAR
  // timing experiment only (wrong at block borders): every corner is read from the block of floor(p)
  if (!DENSE) {
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    const float cx = x - fx, cy = y - fy, cz = z - fz;
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    const long long a0 = locate_voxel<false>(vol, ix, iy, iz, cache);
    const int lx = ix & 7, ly = iy & 7, lz = iz & 7;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int off = ((lx + (c & 1)) & 7) + ((ly + ((c >> 1) & 1)) & 7) * 8 + ((lz + (c >> 2)) & 7) * 64;
      v[c] = (a0 >= 0) ? VX::load_raw_sdf(vol.vba, (size_t)(cache.base + off)) : 32767.0f;
    }
    float r1 = (1.0f - cx) * v[0] + cx * v[1];
    r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
    float r2 = (1.0f - cx) * v[4] + cx * v[5];
    r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
    found = true;
    return VX::to_float((1.0f - cz) * r1 + cz * r2);
  }
#endif
  Corners<VX, DENSE> cn;
  cn.fetch(vol, x, y, z, cache);
  found = true;
  return cn.trilinear();
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
  int iters = 0; (void)iters;
  bool expectBand = false;  // the previous step was inside the band: expect a trilinear read again
  while (total < totalMax) {
#if ITM_RAY_FUSED_FETCH
    // one fetch of the 2x2x2 neighbourhood serves the nearest read and the trilinear re-read
    // (measured slower on MI355X: 155 us vs 72 us -- the extra probes/ALU of every step outweigh
    // the saved round trip; kept for experiments)
    Corners<VX, DENSE> cn;
    cn.fetch(vol, px, py, pz, cache);
    sdf = cn.nearest(px, py, pz, found);
    if (found && (sdf <= 0.1f) && (sdf >= -0.5f)) sdf = cn.trilinear();
#else
    bool fast = false;
#if ITM_RAY_SAMEBLOCK_FAST
    if (!DENSE) {
      // Fast path: the whole 2x2x2 neighbourhood of floor(p) lies in the block this ray already
      // holds in its cache -> no probe is needed and the nearest voxel is one of the 8 corners, so
      // all 8 values are fetched with independent loads in ONE round trip and serve both the
      // nearest read and (if the value is inside the band) the trilinear re-read.
      const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
      const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
      const int bx = floor_div8(ix), by = floor_div8(iy), bz = floor_div8(iz);
      const int lx = ix - bx * 8, ly = iy - by * 8, lz = iz - bz * 8;
      fast = expectBand && (bx == cache.bx && by == cache.by && bz == cache.bz && lx < 7 && ly < 7 && lz < 7);
      if (fast) {
        const size_t a = (size_t)(cache.base + lx + ly * 8 + lz * 64);
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = VX::load_raw_sdf(vol.vba, a + (size_t)((c & 1) + ((c >> 1) & 1) * 8 + (c >> 2) * 64));
        const int cn = ((int)round_ref(px) - ix) | (((int)round_ref(py) - iy) << 1) | (((int)round_ref(pz) - iz) << 2);
        float vn = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if (cn == k) vn = v[k];
        found = true;
        sdf = VX::to_float(vn);
        expectBand = (sdf <= 0.1f) && (sdf >= -0.5f);
        if (expectBand) {
          const float cx = px - fx, cy = py - fy, cz = pz - fz;
          float r1 = (1.0f - cx) * v[0] + cx * v[1];
          r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
          float r2 = (1.0f - cx) * v[4] + cx * v[5];
          r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
          sdf = VX::to_float((1.0f - cz) * r1 + cz * r2);
        }
      }
    }
#endif
#if ITM_RAY_PREDICT_BAND
    if (!fast) {
      // A step inside the band is usually followed by another one: then fetch the 2x2x2 neighbourhood
      // straight away (the nearest voxel is one of its corners) instead of nearest -> trilinear, which
      // saves one dependent round trip per step for rays that graze the surface (the tail of the kernel).
      bool corners = expectBand;
      bool inBand = false;
      if (!corners) {
        sdf = sdf_nearest<VX, DENSE>(vol, px, py, pz, found, cache);
        inBand = found && (sdf <= 0.1f) && (sdf >= -0.5f);
        corners = inBand;
      }
      if (corners) {
        Corners<VX, DENSE> cn;
        cn.fetch(vol, px, py, pz, cache);
        if (expectBand) {
          sdf = cn.nearest(px, py, pz, found);
          inBand = found && (sdf <= 0.1f) && (sdf >= -0.5f);
        }
        if (inBand) sdf = cn.trilinear();
      }
      expectBand = inBand;
    }
#else
    if (!fast) {
      sdf = sdf_nearest<VX, DENSE>(vol, px, py, pz, found, cache);
      expectBand = found && (sdf <= 0.1f) && (sdf >= -0.5f);
      if (expectBand) sdf = sdf_trilinear<VX, DENSE>(vol, px, py, pz, found, cache);
    }
#endif
#endif
#if ITM_EXP_MAXITER
    if (++iters >= ITM_EXP_MAXITER) break;   // timing experiment only: results are wrong
#endif
    if (!found) {
      step = (float)kBlockSide;
    } else {
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
