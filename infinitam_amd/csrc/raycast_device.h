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
#ifndef ITM_RAY_MISS_LOOKAHEAD
#define ITM_RAY_MISS_LOOKAHEAD 0  // K > 0: after a miss, probe the occupancy bits of the next K 8-voxel steps at once
#endif
#ifndef ITM_EXP_WAVE_TIMING
#define ITM_EXP_WAVE_TIMING 0  // experiment: per-wave cycle accounting of cast_ray into g_waveStats (read with itm_debug_read_wave_stats)
#endif
#ifndef ITM_RAY_PREFETCH_COLUMN
#define ITM_RAY_PREFETCH_COLUMN 0  // on entering a block, touch the ray's (x,y) column in all 8 z-slices of the block (LDS-DMA loads into a junk buffer)
#endif
#ifndef ITM_RAY_BITMAP_GUARD
#define ITM_RAY_BITMAP_GUARD 1  // consult the occupancy bitmap before fetching a hash entry (single-voxel lookups)
#endif

namespace itm {

#if ITM_EXP_WAVE_TIMING
static __device__ unsigned long long g_waveStats[8192 * 12];
static __device__ unsigned long long g_waveTrace[160 * 64 * 4];  // waves with index % 32 == 1: per iteration (t0-start, near end, tri end, iteration end | lanes<<48)
#define ITM_WT(...) __VA_ARGS__
__device__ inline unsigned long long wt_wave_max(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
  return v;
}
__device__ inline unsigned long long wt_clock() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
#else
#define ITM_WT(...)
#endif

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
// `guard`: consult the occupancy bitmap first (one extra dependent round trip when the block exists, but an absent
// block is then proven by a 4-byte L2-resident read instead of a 16-byte entry fetch from a 19 MB table).
template <bool DENSE>
__device__ inline long long locate_voxel(const VolumeView& vol, int px, int py, int pz, BlockCache& cache, bool guard = true) {
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
    if (guard && !((vol.headBits[idx >> 5] >> (idx & 31)) & 1u)) return -1;
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
__device__ inline float read_raw_sdf(const VolumeView& vol, int px, int py, int pz, bool& found, BlockCache& cache, bool guard = true) {
#if ITM_RAY_PREFETCH_COLUMN
  const int prevBase = cache.base;
#endif
  const long long a = locate_voxel<DENSE>(vol, px, py, pz, cache, guard);
  found = a >= 0;
  if (!found) return VX::kShort ? 32767.0f : 1.0f;
  const float v = VX::load_raw_sdf(vol.vba, (size_t)a);
#if ITM_RAY_PREFETCH_COLUMN
  if (!DENSE && cache.base != prevBase) {
    // first read in this block: a z-step is a new 128-byte line every time (x + 8y + 64z layout), cold in this
    // XCD's L2.  Request the ray's column of all 8 z-slices now, so the following steps inside the block hit.
    __shared__ int junk[64];
    const char* col = (const char*)vol.vba + ((size_t)cache.base + (size_t)((int)a & 63)) * VX::kBytes;
#pragma unroll
    for (int zz = 0; zz < 8; ++zz)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(col + zz * 64 * VX::kBytes),
                                       (__attribute__((address_space(3))) void*)junk, 4, 0, 0);
  }
#endif
  return v;
}

template <class VX, bool DENSE>
__device__ inline float sdf_nearest(const VolumeView& vol, float x, float y, float z, bool& found, BlockCache& cache, bool guard = true) {
  return VX::to_float(read_raw_sdf<VX, DENSE>(vol, (int)round_ref(x), (int)round_ref(y), (int)round_ref(z), found, cache, guard));
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
    // a tree of selects on the three offset bits (a chain of `if (c == k)` gets turned into a dynamically indexed
    // array by the optimiser, which then moves v[] to LDS: +30 us on the kernel)
    const bool bx = ((int)round_ref(x) - ix) != 0, by = ((int)round_ref(y) - iy) != 0, bz = ((int)round_ref(z) - iz) != 0;
    const float x0 = bx ? v[1] : v[0], x1 = bx ? v[3] : v[2], x2 = bx ? v[5] : v[4], x3 = bx ? v[7] : v[6];
    const bool p0 = bx ? present[1] : present[0], p1 = bx ? present[3] : present[2], p2 = bx ? present[5] : present[4], p3 = bx ? present[7] : present[6];
    const float y0 = by ? x1 : x0, y1 = by ? x3 : x2;
    const bool q0 = by ? p1 : p0, q1 = by ? p3 : p2;
    found = bz ? q1 : q0;
    return VX::to_float(bz ? y1 : y0);
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
  ITM_WT(const unsigned long long wtEntry = wt_clock();)
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
  ITM_WT(unsigned long long wtStart = wt_clock(); unsigned long long wtNear = 0, wtTri = 0; unsigned long long wtMaxIter = 0; unsigned wtMaxIdx = 0; unsigned wtIters = 0, wtTriIters = 0, wtLanesIter = 0, wtLanesTri = 0;)
  while (total < totalMax) {
    ITM_WT(const unsigned long long wt0 = wt_clock(); ++wtIters; const unsigned wtLanesNow = __popcll(__ballot(1)); wtLanesIter += wtLanesNow; unsigned long long wtT1 = wt0, wtT2 = wt0;)
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
      ITM_WT(const unsigned long long anyBand = __ballot(expectBand); const unsigned long long wt1 = wt_clock(); wtNear += wt1 - wt0; wtT1 = wt1; wtT2 = wt1;)
      if (expectBand) sdf = sdf_trilinear<VX, DENSE>(vol, px, py, pz, found, cache);
      ITM_WT(if (anyBand) { const float keep = sdf; asm volatile("" :: "v"(keep)); wtT2 = wt_clock(); wtTri += wtT2 - wt1; ++wtTriIters; wtLanesTri += __popcll(anyBand); })
    }
#endif
#endif
#if ITM_EXP_MAXITER
    if (++iters >= ITM_EXP_MAXITER) break;   // timing experiment only: results are wrong
#endif
#if ITM_RAY_MISS_LOOKAHEAD > 0
    if (!DENSE && !found) {
      // A miss advances the ray by exactly one block side, and so does every following miss: the positions
      // p + k*(8*dir) (accumulated one add at a time, as the sequential loop does) are known in advance.
      // Their occupancy bits are loaded together (one round trip instead of K dependent ones); the ray
      // jumps over the leading run of positions whose bucket is provably empty.  A set bit (or the end of
      // the range) stops the run and the position is examined by the normal path of the next iteration.
      constexpr int K = ITM_RAY_MISS_LOOKAHEAD;
      const float ex = (float)kBlockSide * dx, ey = (float)kBlockSide * dy, ez = (float)kBlockSide * dz;
      float qx[K + 1], qy[K + 1], qz[K + 1], qt[K + 1];
      uint32_t occ[K];
      float ax = px, ay = py, az = pz, at = total;
#pragma unroll
      for (int k = 0; k <= K; ++k) {
        ax += ex; ay += ey; az += ez; at += (float)kBlockSide;
        qx[k] = ax; qy[k] = ay; qz[k] = az; qt[k] = at;
        if (k < K) {
          const int idx = hash_index(floor_div8((int)round_ref(ax)), floor_div8((int)round_ref(ay)), floor_div8((int)round_ref(az)), vol.mask);
          occ[k] = (vol.headBits[idx >> 5] >> (idx & 31)) & 1u;
        }
      }
      px = qx[0]; py = qy[0]; pz = qz[0]; total = qt[0];
      bool go = true;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        go = go && (qt[k] < totalMax) && (occ[k] == 0u);
        if (go) { px = qx[k + 1]; py = qy[k + 1]; pz = qz[k + 1]; total = qt[k + 1]; }
      }
    } else
#endif
    {
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
    ITM_WT({ const float keep2 = total + px; asm volatile("" :: "v"(keep2)); const unsigned long long te = wt_clock(); const unsigned long long d = te - wt0; if (d > wtMaxIter) { wtMaxIter = d; wtMaxIdx = wtIters; }
      const unsigned wv = blockIdx.x * 4 + (threadIdx.x >> 6);
      if ((wv & 31) == 1 && (wv >> 5) < 160 && wtIters <= 64) { unsigned long long* tr = g_waveTrace + ((size_t)(wv >> 5) * 64 + (wtIters - 1)) * 4; tr[0] = wt0 - wtStart; tr[1] = wtT1 - wtStart; tr[2] = wtT2 - wtStart; tr[3] = (te - wtStart) | ((unsigned long long)wtLanesNow << 48); } })
  }
  ITM_WT(const unsigned long long wtLoopEnd = wt_clock();)
  float w = 0.0f;
  if (sdf <= 0.0f) {
    step = sdf * stepScale;
    px += step * dx; py += step * dy; pz += step * dz;
    sdf = sdf_trilinear<VX, DENSE>(vol, px, py, pz, found, cache);
    step = sdf * stepScale;
    px += step * dx; py += step * dy; pz += step * dz;
    w = 1.0f;
  }
#if ITM_EXP_WAVE_TIMING
  {
    const float keepw = px + w; asm volatile("" :: "v"(keepw));
    const unsigned long long wtEnd = wt_clock();
    const unsigned wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    // the accumulators are per lane (a lane only counts while it is active): the lane that stayed longest has the wave's totals
    const unsigned long long mIters = wt_wave_max(wtIters), mTriIters = wt_wave_max(wtTriIters), mNear = wt_wave_max(wtNear), mTri = wt_wave_max(wtTri),
                             mLanesIter = wt_wave_max(wtLanesIter), mLanesTri = wt_wave_max(wtLanesTri);
    wtMaxIter = wt_wave_max(wtMaxIter);
    if ((threadIdx.x & 63) == 0 && wv < 8192) {
      unsigned long long* o = g_waveStats + (size_t)wv * 12;
      o[8] = wtStart - wtEntry; o[9] = wtLoopEnd - wtStart; o[10] = wtEnd - wtLoopEnd; o[11] = wtMaxIter; o[7] = wtMaxIdx;
      o[0] = wtEnd - wtStart; o[1] = mIters; o[2] = mTriIters; o[3] = mNear; o[4] = mTri; o[5] = mLanesIter; o[6] = mLanesTri;
    }
  }
#endif
  return make_float4(px, py, pz, w);
}

#ifndef ITM_RAY_WHILE_WHILE
#define ITM_RAY_WHILE_WHILE 1
#endif
#ifndef ITM_RAY_ADAPTIVE_GUARD
#define ITM_RAY_ADAPTIVE_GUARD 0  // skip the occupancy-bitmap round trip while the ray is in allocated territory; measured 63.0 vs 61.5 us: off
#endif
#ifndef ITM_RAY_PREDICT_TRI
#define ITM_RAY_PREDICT_TRI 0     // after a band step, fetch the 2x2x2 neighbourhood at the next position directly; measured 72 vs 61.5 us: off
#endif
#ifndef ITM_RAY_MARCH_BURST
#define ITM_RAY_MARCH_BURST 4   // cheap steps a lane may take before the wave serves the lanes waiting for a trilinear read
#endif

// castRay restructured as two nested loops ("while-while"): the inner loop only does the cheap part of a
// step (nearest-voxel read, empty-space / far-field advance) and a lane leaves it as soon as its value lies
// inside the truncation band; the expensive trilinear read (band steps and the post-hit refinement alike)
// sits after the inner loop.  SIMT reconvergence then does the scheduling: lanes that need a trilinear
// read wait while the other lanes of the wave take up to ITM_RAY_MARCH_BURST cheap steps, then the
// trilinear code runs once for all of them -- instead of once per iteration in which ANY lane happens to
// be in the band.  Per ray the sequence of positions, reads and float operations is exactly that of
// cast_ray (and of the reference), so results are bit-identical.
//
// Measured on MI355X (config 2, tools/wave_stats.py per-wave cycle traces): the kernel lasts as long as its
// slowest wave (rays that pass the sphere and run ~45 empty-space steps to the wall); a cold voxel/hash
// line costs ~2 000 cycles per dependent round trip, a trilinear step ~4 500.  Burst 1 (= the plain loop)
// 69 us, 2: 65, 3: 64, 4: 62, 8: 66, unbounded: 95 (lanes then serialise each other's empty-space runs).
// Neither a leaner trilinear (upper bound tried with ITM_EXP_SIMPLE_TRILINEAR: no change), nor the
// empty-space look-ahead (ITM_RAY_MISS_LOOKAHEAD, desynchronises the lanes' arrival at the surface: +5 us),
// nor a z-column prefetch at block entry (ITM_RAY_PREFETCH_COLUMN: +12 us) help on top of it.
template <class VX, bool DENSE>
__device__ inline float4 cast_ray_ww(int x, int y, const VolumeView& vol, const RayParams& p, float2 mm) {
  enum : int { MARCH = 0, TRI = 1, REFINE = 2, DONE = 3, TRIP = 4 };
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
  float w = 0.0f;
  int st = (total < totalMax) ? MARCH : DONE;
  bool inAllocated = true;   // rays start at the near face of the visible blocks' bounding range
  ITM_WT(const unsigned long long wtStart = wt_clock(); unsigned wtOuter = 0;)
  while (st != DONE) {
    ITM_WT(const unsigned long long wt0 = wt_clock(); unsigned wtInner = 0; const unsigned wtLanes0 = __popcll(__ballot(1)); const unsigned wtMarch0 = __popcll(__ballot(st == MARCH));)
    // ---- cheap phase: at most ITM_RAY_MARCH_BURST steps, so that waiting lanes are served regularly ----
    int budget = ITM_RAY_MARCH_BURST;
    while (st == MARCH && budget > 0) {
      --budget;
      ITM_WT(++wtInner;)
      const float sdf = sdf_nearest<VX, DENSE>(vol, px, py, pz, found, cache, !(ITM_RAY_ADAPTIVE_GUARD && inAllocated));
      inAllocated = found;
      if (!found) {
        const float ex = (float)kBlockSide * dx, ey = (float)kBlockSide * dy, ez = (float)kBlockSide * dz;
        px += ex; py += ey; pz += ez; total += (float)kBlockSide;
#if ITM_RAY_MISS_LOOKAHEAD > 0
        if (!DENSE) {
          // A miss advances the ray by exactly one block side, and so does every following miss: the next
          // positions (accumulated one add at a time, as the sequential loop does) are known in advance.
          // Their occupancy bits are loaded together (one round trip instead of K dependent ones) and the
          // ray jumps over the leading run of provably empty buckets.  A set bit or the end of the range
          // stops the run; that position is examined by the normal path.
          constexpr int K = ITM_RAY_MISS_LOOKAHEAD;
          float qx[K], qy[K], qz[K], qt[K];
          uint32_t occ[K];
          float ax = px, ay = py, az = pz, at = total;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int idx = hash_index((int)round_ref(ax) >> 3, (int)round_ref(ay) >> 3, (int)round_ref(az) >> 3, vol.mask);
            occ[k] = (vol.headBits[idx >> 5] >> (idx & 31)) & 1u;
            if (!(at < totalMax)) occ[k] = 1u;
            ax += ex; ay += ey; az += ez; at += (float)kBlockSide;
            qx[k] = ax; qy[k] = ay; qz[k] = az; qt[k] = at;
          }
          bool go = true;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            go = go && (occ[k] == 0u);
            if (go) { px = qx[k]; py = qy[k]; pz = qz[k]; total = qt[k]; }
          }
        }
#endif
        if (!(total < totalMax)) st = DONE;
      } else if ((sdf <= 0.1f) && (sdf >= -0.5f)) {
        st = TRI;                        // the position is kept for the trilinear read
      } else if (sdf <= 0.0f) {
        // surface crossed on the nearest value (below the band): first refinement move
        const float step = sdf * stepScale;
        px += step * dx; py += step * dy; pz += step * dz;
        st = REFINE;
      } else {
        const float s = sdf * stepScale;
        const float step = (s < 1.0f) ? 1.0f : s;
        px += step * dx; py += step * dy; pz += step * dz;
        total += step;
        if (!(total < totalMax)) st = DONE;
      }
    }
    // ---- expensive phase: one trilinear read for every lane that waits for one ------------------------
    ITM_WT(const unsigned long long wtA = wt_clock(); const unsigned wtTriLanes = __popcll(__ballot(st == TRI || st == REFINE || st == TRIP)); wtInner = (unsigned)wt_wave_max(wtInner);)
    if (st == TRI || st == REFINE || st == TRIP) {
      Corners<VX, DENSE> cn;
      cn.fetch(vol, px, py, pz, cache);
      float sdf;
      bool band = true;
      if (st == TRIP) {
        // predicted band step: the nearest voxel is one of the fetched corners; redo the decision of the cheap
        // phase on it (same value, same tests) and fall back to its outcome when the ray has left the band
        sdf = cn.nearest(px, py, pz, found);
        inAllocated = found;
        band = found && (sdf <= 0.1f) && (sdf >= -0.5f);
        if (!band) {
          float step;
          if (!found) { step = (float)kBlockSide; st = MARCH; }
          else if (sdf <= 0.0f) { step = sdf * stepScale; st = REFINE; }
          else { const float s = sdf * stepScale; step = (s < 1.0f) ? 1.0f : s; st = MARCH; }
          px += step * dx; py += step * dy; pz += step * dz;
          if (st == MARCH) { total += step; if (!(total < totalMax)) st = DONE; }
        }
      }
      if (band) {
        sdf = cn.trilinear();
        if (st == REFINE) {
          const float step = sdf * stepScale;
          px += step * dx; py += step * dy; pz += step * dz;
          w = 1.0f; st = DONE;
        } else if (sdf <= 0.0f) {
          const float step = sdf * stepScale;
          px += step * dx; py += step * dy; pz += step * dz;
          st = REFINE;
        } else {
          const float s = sdf * stepScale;
          const float step = (s < 1.0f) ? 1.0f : s;
          px += step * dx; py += step * dy; pz += step * dz;
          total += step;
          st = (total < totalMax) ? (ITM_RAY_PREDICT_TRI ? TRIP : MARCH) : DONE;
        }
      }
    }
    ITM_WT({ const float keep2 = total + px; asm volatile("" :: "v"(keep2)); const unsigned long long tB = wt_clock();
      const unsigned wv = blockIdx.x * 4 + (threadIdx.x >> 6);
      if ((wv & 31) == 1 && (wv >> 5) < 160 && wtOuter < 64) { unsigned long long* tr = g_waveTrace + ((size_t)(wv >> 5) * 64 + wtOuter) * 4; tr[0] = wt0 - wtStart; tr[1] = wtA - wtStart; tr[2] = tB - wtStart; tr[3] = wtLanes0 | (wtMarch0 << 8) | (wtTriLanes << 16) | (wtInner << 24); }
      ++wtOuter; })
  }
#if ITM_EXP_WAVE_TIMING
  {
    const unsigned long long wtEnd = wt_clock();
    const unsigned wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long mOuter = wt_wave_max(wtOuter);
    if ((threadIdx.x & 63) == 0 && wv < 8192) { unsigned long long* o = g_waveStats + (size_t)wv * 12; o[0] = wtEnd - wtStart; o[1] = mOuter; }
  }
#endif
  return make_float4(px, py, pz, w);
}

}  // namespace itm
