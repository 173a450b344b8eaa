// raycast_device.h -- device-side voxel lookup and ray marching shared by the visualisation kernels.
//
// Reference behaviour restated (operation order: SURVEY.md Appendix A.6-A.8):
//   pointToVoxelBlockPos / readVoxel (hash, dense)  DeviceAgnostic/ITMRepresentationAccess.h:12-20, :85-142
//   readFromSDF_float_uninterpolated / _interpolated  :144-185
//   castRay                                           DeviceAgnostic/ITMVisualisationEngine.h:92-158
#pragma once

#include "itm_types.h"
#include <type_traits>

#ifndef ITM_EXP_WAVE_TIMING
#define ITM_EXP_WAVE_TIMING 0  // measurement build: per-wave cycle accounting of cast_ray (tools/wave_stats.py)
#endif

namespace itm {

// ---- launch constants of the march (each swept on the final kernels; the numbers are MI355X ray-cast times in frame) ----------------
// cheap steps a lane may take before the wave serves the lanes waiting for a trilinear read.  Round 4 (config 2 / 3 / 5 ray cast):
// 1: 37.7-38.5 us, 2: 37.1-37.7 / c3 +2.6 % frames/s, 3: 38.1-38.3, 4 (rounds 2-3): 38.0-38.8, 6: 40.3
constexpr int kMarchBurst = 2;
// a ray that has just taken this many "block not found" steps in a row is crossing empty space (a silhouette ray on its way from
// the sphere to the wall): phase 1 of the ray-cast workgroup parks it, phase 2 marches the parked rays in waves of their own.
// Config 2 / config 5, one-phase kernel 60.4 / 132.5 us: streak 4: 76 / 152, 6: 62, 8: 54 / 129, 10: 53 / 123, 12: 53 / 117,
// 16: 55 / 118, 24: 58 / 120 -- parking too early also catches rays that only skip a few blocks.
constexpr int kParkStreak = 12;
// directory cells fetched together per round trip by a parked ray's empty-space run (round 2: 4: 55 us, 6: 53, 8: 53; round 4 with a march
// burst of 2, config 2 / config 5 ray cast: 6: 37.2-37.7 / 116-118, 8: 35.8-36.9 / 114.7-114.9, 12: 35.5-36.7 / 112.8-113.1, 16: 40.2-40.6 / 119.6,
// 20: 41.8 / 130, 24: 40.4-42.1 / 131)
constexpr int kParkedLookahead = 12;
// phase 1: "not found" steps in a row after which a ray counts as inside a run for the wave-wide look-ahead, and the cells it fetches
constexpr int kProbeAt = 2;
constexpr int kProbeCells = 10;
// Dense volumes: voxels fetched together by a ray that is crossing free space.  BASELINE configs[2] starts every ray 0.2 m in front
// of the camera and the surface is 1.3-2.3 m away: ~65-115 steps of mu / voxelSize voxels through voxels that read exactly 1 (free
// or never seen), each a dependent round trip.  Measured (config 3 ray cast, event timers): none 90.6 us, 4 voxels 86.6, 8: 92.5,
// 12: 94.0, 16: 103.7 -- the run is NOT what bounds the dense ray cast (its neighbouring rays read the same lines from L2); kept at 4.
// Round 5, with all loads of a look-ahead really in flight together (far_run): 4: 70.7-71.3 us, 6: > 75, 8: 75.3-75.7.
constexpr int kDenseLookahead = 4;

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
  const int32_t* dirPtr;    // block directory (itm_types.h); nullptr = walk the table (hash index only)
  const void* sdfMirror;    // sdf by position (itm_types.h); nullptr = none (hash index only)
  const int32_t* pageTable; // the mirror's page table as THIS kernel reads it: the scene's (memory) or the workgroup's copy in LDS (raycast_kernel)
  AccelOrigin org;          // where the directory / mirror cubes lie
  int sx, sy, sz;      // dense size
  int ox, oy, oz;      // dense offset
  // which form of the sdf mirror the code reading through this view is compiled for: -1 = decided at run time from org.mMaxPages
  // (both forms compiled in: the free-view and helper kernels), see VolumeViewM
  static constexpr int kMirror = -1;
};
// The same view for a kernel that is compiled for ONE form of the mirror (0 none, 1 dense cube, 2 paged: itm_types.h): the ray-cast
// kernel needs every tile of the image resident at once -- 4 800 waves, five per SIMD, at most 96 vector registers -- and with both
// forms' address arithmetic alive it takes 104.  Every function below takes the view as a template parameter and folds the other form away.
template <int M> struct VolumeViewM : VolumeView {
  static constexpr int kMirror = M;
  __device__ VolumeViewM(const VolumeView& v) : VolumeView(v) {}
};
template <class VOL> __device__ inline bool mirror_is_dense(const VOL& vol) { return VOL::kMirror == 1 || (VOL::kMirror < 0 && vol.org.mMaxPages < 0); }
template <class VOL> __device__ inline bool mirror_is_paged(const VOL& vol) { return VOL::kMirror == 2 || (VOL::kMirror < 0 && vol.org.mMaxPages > 0); }

// per-ray block cache: ITMVoxelBlockHash::IndexCache (Objects/ITMVoxelBlockHash.h:27-33)
struct BlockCache {
  int bx, by, bz;
  int base;
  // the mirror page the ray was last in (itm_types.h): table index and entry -- a page is 32 voxels wide, a step at most 8, so most
  // steps find their page here and go straight to the one load that follows from the position
  uint32_t pageIdx;
  int page;
  __device__ BlockCache() : bx(0x7fffffff), by(0x7fffffff), bz(0x7fffffff), base(-1), pageIdx(0xffffffffu), page(kPageNone) {}
};

// The mirror page with table index tIdx, through the per-lane cache.  The table is read by a wave only when one of its lanes has left
// its page (uniform branch; a lane that keeps its page reads entry 0 and drops it).
__device__ inline int mirror_page_of(const VolumeView& vol, bool inCube, uint32_t tIdx, BlockCache& cache) {
  const bool need = inCube && tIdx != cache.pageIdx;
  if (__any(need)) {
    const int v = vol.pageTable[need ? tIdx : 0u];
    if (need) { cache.pageIdx = tIdx; cache.page = v; }
  }
  return cache.page;
}

__device__ inline int floor_div8(int p) { return ((p < 0) ? p - 7 : p) / 8; }
// ROUND(x) = x<0 ? x-0.5 : x+0.5 (ORUtils/MathUtils.h:21-23); after the (int) truncation this equals
// x + copysign(0.5, x) for every x (they only differ for x = -0.0: -0.5 vs +0.5, both truncate to 0).
__device__ inline float round_ref(float x) { return x + __builtin_copysignf(0.5f, x); }

// Linear voxel index of integer point (px,py,pz), or -1 when no voxel is stored there.
template <bool DENSE>
__device__ inline long long locate_voxel(const VolumeView& vol, int px, int py, int pz, BlockCache& cache) {
  if (DENSE) {
    // (unsigned: 0 <= q < size in one comparison per axis; `&`, not `&&`: three compares and two ANDs instead of a ladder of
    // exec-masked branches)
    const uint32_t qx = (uint32_t)(px - vol.ox), qy = (uint32_t)(py - vol.oy), qz = (uint32_t)(pz - vol.oz);
    const bool in = (qx < (uint32_t)vol.sx) & (qy < (uint32_t)vol.sy) & (qz < (uint32_t)vol.sz);
    return in ? (long long)(qx + qy * (uint32_t)vol.sx + qz * (uint32_t)(vol.sx * vol.sy)) : -1ll;
  } else {
    const int bx = px >> 3, by = py >> 3, bz = pz >> 3;                       // floor(p / 8): arithmetic shift
    const int lin = (px & 7) + ((py & 7) << 3) + ((pz & 7) << 6);             // (p - 8 b) per axis
    if (bx == cache.bx && by == cache.by && bz == cache.bz) return (long long)cache.base + lin;
    {
      // covered by the block directory: one load instead of the table walk (same answer: the directory holds exactly
      // the entries with ptr >= 0)
      const uint32_t ux = (uint32_t)(bx - vol.org.dx), uy = (uint32_t)(by - vol.org.dy), uz = (uint32_t)(bz - vol.org.dz);
      if (vol.dirPtr && dir_covers(ux, uy, uz)) {
        const int ptr = vol.dirPtr[dir_cell(ux, uy, uz)];
        if (ptr < 0) return -1;
        cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = ptr * kBlockVoxels;
        return (long long)cache.base + lin;
      }
    }
    int idx = hash_index(bx, by, bz, vol.mask);
    // the 16-byte entry is only fetched when the occupancy bit says the bucket is in use
    if (!((vol.headBits[idx >> 5] >> (idx & 31)) & 1u)) return -1;
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

// Dense volumes on the ray-march path: the same index as locate_voxel<true> in 32 bits, 0xffffffff = no voxel there (a dense scene holds
// fewer than 2^32 - 1 voxels: checked at creation, scene.hip).  The 64-bit form cost the issue-bound dense ray cast two quarter-rate 64-bit
// multiply-adds in an exec-masked branch, two 64-bit comparisons and a pair of selects per voxel read (found in the ISA, round 6); here
// the products are 24-bit multiply-adds when the slice fits (sx * sy < 2^24 and every side < 2^24, a wave-uniform test).
// (qx, qy, qz): the point relative to the volume's first voxel, as unsigned (0 <= q < size in one comparison per axis).  Returns the
// linear index, 0 for a point without a voxel (always a valid address); `in` says which.
__device__ inline uint32_t dense_lin(const VolumeView& vol, uint32_t qx, uint32_t qy, uint32_t qz, bool& in, bool use = true) {
  in = use & (qx < (uint32_t)vol.sx) & (qy < (uint32_t)vol.sy) & (qz < (uint32_t)vol.sz);
  const uint32_t sx = (uint32_t)vol.sx, sxy = (uint32_t)(vol.sx * vol.sy);
  uint32_t lin;
  if ((((uint32_t)vol.sx | (uint32_t)vol.sy | (uint32_t)vol.sz | sxy) >> 24) == 0u) {      // (uniform)
    // two full-rate 24-bit multiply-adds (written out: from __umul24 with a scalar factor the compiler makes a mask and the quarter-rate
    // v_mul_lo_u32).  A lane outside the volume may have q >= 2^24: its product is not used.
    uint32_t t;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(qy), "s"(sx), "v"(qx));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(lin) : "v"(qz), "s"(sxy), "v"(t));
  } else lin = qx + qy * sx + qz * sxy;
  uint32_t r = in ? lin : 0u;
  asm("" : "+v"(r));      // (the select stays in front of the 64-bit address arithmetic: behind it, it is two selects on a shifted pair)
  return r;
}
__device__ inline uint32_t dense_lin(const VolumeView& vol, int px, int py, int pz, bool& in) {
  return dense_lin(vol, (uint32_t)(px - vol.ox), (uint32_t)(py - vol.oy), (uint32_t)(pz - vol.oz), in);
}

// raw (unconverted) sdf of the voxel at an integer point; the default voxel when absent
template <class VX, bool DENSE, class VOL = VolumeView>
__device__ inline float read_raw_sdf(const VOL& vol, int px, int py, int pz, bool& found, BlockCache& cache) {
  if constexpr (!DENSE) {
    // sdf mirror: one load, address from the position alone (same value and same "found" as the walk below: the mirror holds
    // exactly the voxels of the allocated blocks inside its cube)
    using MC = MirrorCodec<VX::kShort>;
    if (vol.sdfMirror && mirror_is_dense(vol)) {
      // DENSE cube (itm_types.h): the load is unconditional (cell 0 for a lane outside the cube) and the general path below is skipped by
      // a UNIFORM branch when every lane was served: no exec-mask bracket around the common case (ray cast 42.1 -> 41.7 us)
      const uint32_t ux = (uint32_t)((px >> 3) - vol.org.mx), uy = (uint32_t)((py >> 3) - vol.org.my), uz = (uint32_t)((pz >> 3) - vol.org.mz);
      const int mbits = mirror_dense_bits(vol.org);
      const bool covered = mirror_dense_covers(ux, uy, uz, mbits);
      const size_t mi = ((size_t)mirror_dense_cell(ux, uy, uz, mbits) << 9) | (size_t)((px & 7) + ((py & 7) << 3) + ((pz & 7) << 6));
      const typename MC::T v = ((const typename MC::T*)vol.sdfMirror)[covered ? mi : (size_t)0];
      const bool present = covered && !MC::absent(v);
      const float value = present ? MC::raw(v) : (VX::kShort ? 32767.0f : 1.0f);
      if (__all(covered)) { found = present; return value; }
      if (covered) { found = present; return value; }
    } else if (vol.sdfMirror && mirror_is_paged(vol)) {
      // PAGED cube: the page's table entry (per-lane cache, else one read), then the one load that follows from the position
      const uint32_t vx = (uint32_t)(px - (vol.org.mx << 3)), vy = (uint32_t)(py - (vol.org.my << 3)), vz = (uint32_t)(pz - (vol.org.mz << 3));      // cube-relative voxel
      const bool inCube = mirror_covers_voxel(vx, vy, vz);
      const int page = mirror_page_of(vol, inCube, mirror_table_index_voxel(vx, vy, vz), cache);
      // the page answers: with a value, or -- no block was ever allocated in it -- with "no block" and no further load
      const bool covered = inCube && page != kPageUnmappable;
      const bool mapped = covered && page >= 0;
      typename MC::T v = VX::kShort ? (typename MC::T)-32768 : (typename MC::T)0xffffffffu;
      if (__any(mapped)) {
        const size_t mi = mapped ? mirror_element(page, mirror_in_page(vx, vy, vz)) : (size_t)0;
        const typename MC::T got = ((const typename MC::T*)vol.sdfMirror)[mi];
        if (mapped) v = got;
      }
      const bool present = mapped && !MC::absent(v);
      const float value = present ? MC::raw(v) : (VX::kShort ? 32767.0f : 1.0f);
      if (__all(covered)) { found = present; return value; }
      if (covered) { found = present; return value; }
    }
  }
  if constexpr (DENSE) {
    // the load is unconditional (voxel 0 for a position outside the volume): inside an exec-masked branch the compiler waits for it there
    const uint32_t a = dense_lin(vol, px, py, pz, found);
    const float v = VX::load_raw_sdf(vol.vba, (size_t)a);
    return found ? v : (VX::kShort ? 32767.0f : 1.0f);
  } else {
    const long long a = locate_voxel<DENSE>(vol, px, py, pz, cache);
    found = a >= 0;
    if (!found) return VX::kShort ? 32767.0f : 1.0f;
    return VX::load_raw_sdf(vol.vba, (size_t)a);
  }
}

template <class VX, bool DENSE, class VOL = VolumeView>
__device__ inline float sdf_nearest(const VOL& vol, float x, float y, float z, bool& found, BlockCache& cache) {
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

  // Every load below is issued unconditionally from an address that is always valid (a lane that has nothing to fetch
  // reads voxel 0 / its own cell again and drops the value): a load inside an exec-masked branch forces the compiler to
  // wait for it inside that branch, which turned the eight voxel reads of a trilinear sample into eight serial round
  // trips (round 1's "4 500 cycles per trilinear step").  Unconditional, they are in flight together.
  template <class VOL>
  __device__ inline void fetch(const VOL& vol, float x, float y, float z, BlockCache& cache) {
    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    cx = x - fx; cy = y - fy; cz = z - fz;
    ix = (int)fx; iy = (int)fy; iz = (int)fz;
    const float dflt = VX::kShort ? 32767.0f : 1.0f;
    size_t addr[8];
    if (DENSE) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        addr[c] = (size_t)dense_lin(vol, ix + (c & 1), iy + ((c >> 1) & 1), iz + (c >> 2), present[c]);
      }
    } else {
      {
        // sdf mirror: eight independent loads.  Taken when every lane of the wave that is here can use it (a wave with a lane
        // outside the mirrored cube takes the general path as a whole: both give the same values)
        using MC = MirrorCodec<VX::kShort>;
        if (vol.sdfMirror && mirror_is_dense(vol)) {
          // DENSE cube.  The eight addresses from ONE: with the blocks in plain x-fastest order the +1 neighbour along an axis is one voxel
          // further inside the block, or -- from the block's last voxel -- the first voxel of the next block, a fixed distance either
          // way.  (Taken when the blocks of (ix, iy, iz) and of (ix, iy, iz) + 8 per axis both lie in the cube: one block more than the
          // neighbourhood needs at the cube's upper faces, where the general path gives the same values.)  ~35 vector instructions for
          // the eight addresses instead of ~140 -- half of what a trilinear read issued.
          const uint32_t mx = (uint32_t)((ix >> 3) - vol.org.mx), my = (uint32_t)((iy >> 3) - vol.org.my), mz = (uint32_t)((iz >> 3) - vol.org.mz);
          const int mbits = mirror_dense_bits(vol.org);
          const bool all = mirror_dense_covers(mx, my, mz, mbits) && mirror_dense_covers(mx + 1u, my + 1u, mz + 1u, mbits);
          const int kx = ix & 7, ky = iy & 7, kz = iz & 7;
          const size_t base = ((size_t)mirror_dense_cell(mx, my, mz, mbits) << 9) + (size_t)(kx + (ky << 3) + (kz << 6));
          const uint32_t ox = (kx == 7) ? 512u - 7u : 1u;
          const uint32_t oy = (ky == 7) ? (512u << mbits) - 56u : 8u;
          const uint32_t oz = (kz == 7) ? (512u << (2 * mbits)) - 448u : 64u;
          if (__all(all)) {
            typename MC::T m[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) m[c] = ((const typename MC::T*)vol.sdfMirror)[base + (size_t)(((c & 1) ? ox : 0u) + ((c & 2) ? oy : 0u) + ((c & 4) ? oz : 0u))];
#pragma unroll
            for (int c = 0; c < 8; ++c) { present[c] = !MC::absent(m[c]); v[c] = present[c] ? MC::raw(m[c]) : dflt; }
            return;
          }
        } else if (vol.sdfMirror && mirror_is_paged(vol)) {
          // PAGED cube.  The eight addresses from ONE: inside a page the blocks lie x-fastest at a kilobyte each, so the +1 neighbour along an axis is one
          // voxel further inside the block or -- from the block's last voxel -- the first voxel of the next block of the page, a fixed
          // distance either way (~35 vector instructions for the eight addresses instead of ~140).  That holds while the neighbourhood
          // stays inside ONE page (127 of 128 positions per axis); a wave with a lane whose neighbourhood straddles pages looks up the page
          // of each of the eight voxels instead (eight independent table reads, then the eight values).
          const uint32_t vx = (uint32_t)(ix - (vol.org.mx << 3)), vy = (uint32_t)(iy - (vol.org.my << 3)), vz = (uint32_t)(iz - (vol.org.mz << 3));
          const bool inCube = mirror_covers_voxel(vx, vy, vz) && mirror_covers_voxel(vx + 1u, vy + 1u, vz + 1u);
          const bool onePage = ((vx & kPageVoxMask) != kPageVoxMask) && ((vy & kPageVoxMask) != kPageVoxMask) && ((vz & kPageVoxMask) != kPageVoxMask);
          const typename MC::T none = VX::kShort ? (typename MC::T)-32768 : (typename MC::T)0xffffffffu;
          if (__all(inCube && onePage)) {
            const int page = mirror_page_of(vol, inCube, mirror_table_index_voxel(vx, vy, vz), cache);
            if (__all(page != kPageUnmappable)) {
              const bool mapped = page >= 0;
              typename MC::T m[8];
#pragma unroll
              for (int c = 0; c < 8; ++c) m[c] = none;
              if (__any(mapped)) {
                const size_t base = mapped ? mirror_element(page, mirror_in_page(vx, vy, vz)) : (size_t)0;
                const uint32_t ox = ((vx & 7u) == 7u) ? 512u - 7u : 1u;
                const uint32_t oy = ((vy & 7u) == 7u) ? (512u << kPageBits) - 56u : 8u;
                const uint32_t oz = ((vz & 7u) == 7u) ? (512u << (2 * kPageBits)) - 448u : 64u;
                typename MC::T got[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) got[c] = ((const typename MC::T*)vol.sdfMirror)[mapped ? base + (size_t)(((c & 1) ? ox : 0u) + ((c & 2) ? oy : 0u) + ((c & 4) ? oz : 0u)) : (size_t)0];
#pragma unroll
                for (int c = 0; c < 8; ++c) if (mapped) m[c] = got[c];
              }
#pragma unroll
              for (int c = 0; c < 8; ++c) { present[c] = !MC::absent(m[c]); v[c] = present[c] ? MC::raw(m[c]) : dflt; }
              return;
            }
          } else if (__all(inCube)) {
            // a neighbourhood that straddles pages: the page of every corner's voxel
            int pg[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) pg[c] = vol.pageTable[mirror_table_index_voxel(vx + (uint32_t)(c & 1), vy + (uint32_t)((c >> 1) & 1), vz + (uint32_t)(c >> 2))];
            bool usable = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) usable &= pg[c] != kPageUnmappable;
            if (__all(usable)) {
              typename MC::T got[8];
#pragma unroll
              for (int c = 0; c < 8; ++c)
                got[c] = ((const typename MC::T*)vol.sdfMirror)[pg[c] >= 0 ? mirror_element(pg[c], mirror_in_page(vx + (uint32_t)(c & 1), vy + (uint32_t)((c >> 1) & 1), vz + (uint32_t)(c >> 2))) : (size_t)0];
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const typename MC::T mv = pg[c] >= 0 ? got[c] : none;
                present[c] = !MC::absent(mv); v[c] = present[c] ? MC::raw(mv) : dflt;
              }
              return;
            }
          }
        }
      }
      const int bx = ix >> 3, by = iy >> 3, bz = iz >> 3;            // floor division by 8
      const int lx = ix & 7, ly = iy & 7, lz = iz & 7;
      // bit k set: the +1 neighbour along axis k lies in the next block
      const int cross = (lx == 7 ? 1 : 0) | (ly == 7 ? 2 : 0) | (lz == 7 ? 4 : 0);
      const bool cached = (bx == cache.bx && by == cache.by && bz == cache.bz);
      int base[8];
      const uint32_t ux = (uint32_t)(bx - vol.org.dx), uy = (uint32_t)(by - vol.org.dy), uz = (uint32_t)(bz - vol.org.dz);
      const bool viaDir = vol.dirPtr && dir_covers(ux, uy, uz) && dir_covers(ux + 1u, uy + 1u, uz + 1u);
      // block directory: the blocks of the 2x2x2 neighbourhood are eight 4-byte cells, mostly of one 256-byte brick.  A wave
      // whose lanes all sit inside their cached block and away from its upper faces skips the round altogether.
      if (__any(viaDir && !(cached && cross == 0))) {
        int ptr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int t = s & cross;                                     // block of corner pattern s (s itself when it crosses)
          const uint32_t cell = viaDir ? dir_cell(ux + (uint32_t)(t & 1), uy + (uint32_t)((t >> 1) & 1), uz + (uint32_t)(t >> 2)) : 0u;
          ptr[s] = vol.dirPtr[cell];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) base[s] = (ptr[s] < 0) ? -1 : ptr[s] * kBlockVoxels;
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) base[s] = cache.base;             // every lane: cached && cross == 0 (or no lane uses the directory)
      }
      if (!viaDir) {
        // outside the directory (or directory disabled): table walk; entries are fetched for the blocks actually needed, the
        // other loads re-read the first entry
        const int idx0 = hash_index(bx, by, bz, vol.mask);
        HashEntry head[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bool need = ((s & ~cross) == 0) && !(s == 0 && cached);
          head[s] = unpack_entry(vol.hash[need ? hash_index(bx + (s & 1), by + ((s >> 1) & 1), bz + (s >> 2), vol.mask) : idx0]);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bool need = ((s & ~cross) == 0) && !(s == 0 && cached);
          base[s] = need ? resolve_block(vol, head[s], bx + (s & 1), by + ((s >> 1) & 1), bz + (s >> 2)) : -1;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) base[s] = base[s & cross];
      }
      if (cached) {
#pragma unroll
        for (int s = 0; s < 8; ++s) if ((s & cross) == 0) base[s] = cache.base;
      } else if (base[0] >= 0) { cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = base[0]; }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int b = base[c];
        const int off = ((lx + (c & 1)) & 7) + ((ly + ((c >> 1) & 1)) & 7) * 8 + ((lz + (c >> 2)) & 7) * 64;
        present[c] = b >= 0;
        addr[c] = present[c] ? (size_t)(b + off) : (size_t)0;
      }
    }
    float raw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) raw[c] = VX::load_raw_sdf(vol.vba, addr[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = present[c] ? raw[c] : dflt;
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

template <class VX, bool DENSE, class VOL = VolumeView>
__device__ inline float sdf_trilinear(const VOL& vol, float x, float y, float z, bool& found, BlockCache& cache) {
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

// Ray set-up shared by both loops: start point, direction and the [total, totalMax) range in voxel units.
struct RaySetup { float px, py, pz, dx, dy, dz, total, totalMax; };

__device__ inline RaySetup ray_setup(int x, int y, const RayParams& p, float2 mm) {
  RaySetup r;
  float pcz = mm.x;
  float pcx = pcz * (((float)x - p.cx) * p.ifx);
  float pcy = pcz * (((float)y - p.cy) * p.ify);
  float acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  r.total = sqrtf(acc) * p.oneOverVoxel;
  Vec3 t = transform_point(p.invM, pcx, pcy, pcz);
  const float sx = t.x * p.oneOverVoxel, sy = t.y * p.oneOverVoxel, sz = t.z * p.oneOverVoxel;
  pcz = mm.y;
  pcx = pcz * (((float)x - p.cx) * p.ifx);
  pcy = pcz * (((float)y - p.cy) * p.ify);
  acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  r.totalMax = sqrtf(acc) * p.oneOverVoxel;
  t = transform_point(p.invM, pcx, pcy, pcz);
  float dx = t.x * p.oneOverVoxel - sx, dy = t.y * p.oneOverVoxel - sy, dz = t.z * p.oneOverVoxel - sz;
  const float dn = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  r.dx = dx * dn; r.dy = dy * dn; r.dz = dz * dn;
  r.px = sx; r.py = sy; r.pz = sz;
  return r;
}

// castRay restructured as two nested loops ("while-while"): the inner loop only does the cheap part of a
// step (nearest-voxel read, empty-space / far-field advance) and a lane leaves it as soon as its value lies
// inside the truncation band; the expensive trilinear read (band steps and the post-hit refinement alike)
// sits after the inner loop.  SIMT reconvergence then does the scheduling: lanes that need a trilinear
// read wait while the other lanes of the wave take up to kMarchBurst cheap steps, then the
// trilinear code runs once for all of them -- instead of once per iteration in which ANY lane happens to
// be in the band.  Per ray the sequence of positions, reads and float operations is exactly that of
// the reference's loop (DeviceAgnostic/ITMVisualisationEngine.h:92-158), so results are bit-identical.
//
// Measured on MI355X (config 2, tools/wave_stats.py per-wave cycle traces): the kernel lasts as long as its
// slowest wave (rays that pass the sphere and run ~45 empty-space steps to the wall); a cold voxel/hash
// line costs ~2 000 cycles per dependent round trip, a trilinear step ~4 500.  Burst 1 (= the plain loop)
// 69 us, 2: 65, 3: 64, 4: 62, 8: 66, unbounded: 95 (lanes then serialise each other's empty-space runs).
// The other restructurings that were tried and dropped are listed with their numbers in DESIGN.md section 5.
// One look-ahead round of a far-field run in a DENSE volume.  A ray whose single-voxel read returned exactly 1 steps by
// max(1 * stepScale, 1) voxels and, in free space, reads 1 again: the voxels of the next K positions -- q0 = pt, q1 = pt + step dir,
// ..., each computed with the reference's own operations -- are fetched together and the ray advances over as many of them as
// read exactly 1 (the reference's step for such a value, with its length update and range test); it stops in front of the first
// other value (or position outside the volume), which the regular loop then reads again.  K dependent round trips become one.
// Returns the number of steps taken.
template <class VX, int K>
__device__ inline int far_run(const VolumeView& vol, bool runner, float& px, float& py, float& pz, float& total, float dx, float dy, float dz,
                              float stepScale, float totalMax, bool& ended) {
  const float one = 1.0f * stepScale;                       // sdf * stepScale with sdf == 1
  const float step = (one < 1.0f) ? 1.0f : one;
  const float sx = step * dx, sy = step * dy, sz = step * dz;
  const float farRaw = VX::kShort ? 32767.0f : 1.0f;
  // bit j set: q_j holds something else than exactly 1 (or lies outside the volume, or the lane is no runner): the run stops in front of it.
  // EVERY value goes into the mask, so all K loads are in flight together: with the values consumed one by one in an accept / break
  // chain the compiler sinks the last load into the chain (it is only needed when all others were accepted), i.e. one more dependent
  // round trip in exactly the common case of a run through free space (found in the ISA, round 5).
  uint32_t stop = 0;
  {
    float qx = px, qy = py, qz = pz;
    uint32_t at[K];
    bool in[K];
    float raw[K];
    // (all addresses first, then all loads: interleaved, the register allocator reused a load's destination inside the next address's
    // 64-bit multiply-add and the wave waited for the load there)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      at[j] = dense_lin(vol, (uint32_t)((int)round_ref(qx) - vol.ox), (uint32_t)((int)round_ref(qy) - vol.oy), (uint32_t)((int)round_ref(qz) - vol.oz), in[j], runner);
      qx += sx; qy += sy; qz += sz;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < K; ++j) raw[j] = VX::load_raw_sdf(vol.vba, (size_t)at[j]);   // voxel 0 / no use for the others
#pragma unroll
    for (int j = 0; j < K; ++j) stop |= (((raw[j] != farRaw) | !in[j]) ? 1u : 0u) << j;      // outside the volume / no runner: stops the run
  }
  const int n = runner ? __builtin_ctz(stop | (1u << K)) : 0;      // positions that read exactly 1, counted from the first
  int taken = 0;
  while (taken < n) {
    px += sx; py += sy; pz += sz; total += step;               // the reference's step for a value of exactly 1, with its length update ...
    ++taken;
    if (!(total < totalMax)) { ended = true; break; }         // ... and its range test
  }
  return taken;
}

// a parked ray: where it stands and how far it has come; direction, end of range etc. are recomputed from the pixel
struct RayResume { float px, py, pz, total; };

// castRay restructured as two nested loops ("while-while"), resumable.
//   LOOKAHEAD > 0 (hash index with the block directory): after a "block not found" step the directory cells of the next K
//     positions of the ray -- computed with the reference's own additions, pt += 8 dir -- are fetched together and the ray
//     advances over as many of them as are empty: K dependent round trips become one.  In a wave where only SOME lanes cross
//     empty space this loses (every lane pays the K classifications and loads: 61 -> 64 / 69 / 77 us for K = 3 / 6 / 10, gated or
//     not); in a wave of nothing but such rays it is what makes their ~45-step runs cheap.
//   PARK: the ray stops (parked = true, returns its position and length in xyz / w) once it has taken kParkStreak
//     "not found" steps in a row; the caller appends it to the queue of the second pass.
// Per ray the sequence of positions, reads and float operations is that of the reference, whatever the pass structure.
template <class VX, bool DENSE, int LOOKAHEAD, bool PARK, class VOL = VolumeView>
__device__ inline float4 march_ray(int x, int y, const VOL& vol, const RayParams& p, float2 mm, const RayResume* resume, bool& parked) {
  // MARCH: next read is a single voxel; TRI: a single-voxel read found the band, the trilinear read of the same position is
  // due; REFINE: the surface was crossed
  enum : int { MARCH = 0, TRI = 1, REFINE = 2, DONE = 4 };
  const float stepScale = p.mu * p.oneOverVoxel;
  const RaySetup r = ray_setup(x, y, p, mm);
  float px = r.px, py = r.py, pz = r.pz, total = r.total;
  const float dx = r.dx, dy = r.dy, dz = r.dz, totalMax = r.totalMax;
  int missStreak = 0;
  if (resume) { px = resume->px; py = resume->py; pz = resume->pz; total = resume->total; missStreak = kParkStreak; }
  BlockCache cache;
  bool found;
  float w = 0.0f;
  parked = false;
  int st = (total < totalMax) ? MARCH : DONE;
  // one forward step of the march loop for a value that is not in the band (or a trilinear value): returns the next state
  auto advance = [&](bool fnd, float sdf) -> int {
    float step;
    int next = MARCH;
    if (!fnd) step = (float)kBlockSide;
    else if (sdf <= 0.0f) { step = sdf * stepScale; next = REFINE; }          // surface crossed: first refinement move, no length update
    else { const float s = sdf * stepScale; step = (s < 1.0f) ? 1.0f : s; }
    px += step * dx; py += step * dy; pz += step * dz;
    if (next != REFINE) { total += step; if (!(total < totalMax)) next = DONE; }
    return next;
  };
  // One look-ahead round of an empty-space run (hash index with the block directory): the directory cells of the positions q0 = pt,
  // q1 = pt + 8 dir, ... -- computed with the reference's own additions -- are fetched together (cell 0 / no use for lanes that are not
  // runners) and the ray advances over as many of them as are empty, each the reference's step for a position without a block, with
  // its length update and range test; it stops in front of the first position that holds a block (or lies outside the directory),
  // which the regular loop then reads.  Returns the number of steps taken.
  auto miss_run = [&](auto kc, bool runner) -> int {
    constexpr int K = decltype(kc)::value;
    const float sx = (float)kBlockSide * dx, sy = (float)kBlockSide * dy, sz = (float)kBlockSide * dz;   // exact products
    // (Round 5: with every cell folded into one mask -- all K loads awaited together, the count taken with a find-first-bit -- the hash
    // ray cast got SLOWER, configs[1] 36.5 -> 37.3 us, configs[4] 111.5 -> 119 us: the chain below consumes the cells as they arrive and
    // leaves at the first block, and the compiler's habit of issuing the last cell's load only when all others were empty costs less
    // than waiting for all of them every time.  profiles/r5_raycast_notes.md)
    int ahead[K];
    {
      float qx = px, qy = py, qz = pz;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const uint32_t ux = (uint32_t)(((int)round_ref(qx) >> 3) - vol.org.dx), uy = (uint32_t)(((int)round_ref(qy) >> 3) - vol.org.dy),
                       uz = (uint32_t)(((int)round_ref(qz) >> 3) - vol.org.dz);
        const bool use = runner && dir_covers(ux, uy, uz);
        const int v = vol.dirPtr[use ? dir_cell(ux, uy, uz) : 0u];
        ahead[j] = use ? v : 0;                               // 0 = "cannot tell / a block": stops the run
        qx += sx; qy += sy; qz += sz;
      }
    }
    int taken = 0;
    if (runner) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (ahead[j] >= 0) break;                             // q_j holds a block (or lies outside the directory): regular read next
        px += sx; py += sy; pz += sz; total += (float)kBlockSide;   // the reference's step for a position without a block
        ++taken;
        if (!(total < totalMax)) { st = DONE; break; }
      }
    }
    return taken;
  };
  ITM_WT(const unsigned long long wtStart = wt_clock(); unsigned wtOuter = 0;)
  bool missed = false;      // DENSE: the last single-voxel read found no voxel
  while (st != DONE) {
    if constexpr (DENSE) {
      // A ray that has LEFT the volume -- outside on an axis along which it moves away -- never comes back (its coordinate on that
      // axis only grows in magnitude, additions of one sign are monotone in float arithmetic too, and so is the rounding of the
      // look-up): every further read finds nothing and the reference takes its 8-voxel step until the range ends.  Once every lane
      // of the wave is done or gone, those steps are taken here without the look-ups: the same additions in the same order, ~6
      // instructions instead of ~35 per step (BASELINE configs[2]: the wall lies behind the volume, its rays take ~30 such steps).
      if (__any(missed)) {
        bool gone = false;
        if (st == MARCH && missed) {
          const int ix = (int)round_ref(px) - vol.ox, iy = (int)round_ref(py) - vol.oy, iz = (int)round_ref(pz) - vol.oz;
          gone = (ix >= vol.sx && dx >= 0.0f) || (ix < 0 && dx <= 0.0f) || (iy >= vol.sy && dy >= 0.0f) || (iy < 0 && dy <= 0.0f) ||
                 (iz >= vol.sz && dz >= 0.0f) || (iz < 0 && dz <= 0.0f);
        }
        if (__all(st == DONE || gone)) {
          const float step = (float)kBlockSide;
          const float sx = step * dx, sy = step * dy, sz = step * dz;        // the products of advance(false, .)
          while (st == MARCH) {
            px += sx; py += sy; pz += sz;
            total += step;
            if (!(total < totalMax)) st = DONE;
          }
          break;
        }
      }
    }
    ITM_WT(const unsigned long long wt0 = wt_clock(); unsigned wtInner = 0; const unsigned wtLanes0 = __popcll(__ballot(1)); const unsigned wtMarch0 = __popcll(__ballot(st == MARCH));)
    // ---- cheap phase: at most kMarchBurst single-voxel steps, so that waiting lanes are served regularly ----
    int budget = kMarchBurst;
    while (st == MARCH && budget > 0) {
      --budget;
      ITM_WT(++wtInner;)
      const float sdf = sdf_nearest<VX, DENSE>(vol, px, py, pz, found, cache);
      {
        // the step as selects rather than branches (the same operations on the same values as advance(); a lone wave pays every
        // taken or skipped branch of a step with issue bubbles, profiles/r3_raycast_notes.md)
        const bool band = found && (sdf <= 0.1f) && (sdf >= -0.5f);       // the position is kept for the trilinear read
        const bool crossed = found && (sdf <= 0.0f);                      // (below the band) first refinement move, no length update
        const float s = sdf * stepScale;
        const float fwd = (s < 1.0f) ? 1.0f : s;
        const float step = found ? (crossed ? s : fwd) : (float)kBlockSide;
        const float nx = px + step * dx, ny = py + step * dy, nz = pz + step * dz, nt = total + step;
        px = band ? px : nx; py = band ? py : ny; pz = band ? pz : nz;
        total = (band || crossed) ? total : nt;
        st = band ? TRI : crossed ? REFINE : (total < totalMax) ? MARCH : DONE;
      }
      if constexpr (DENSE) missed = !found;
      if constexpr (DENSE && LOOKAHEAD > 0) {
        // (Several look-aheads back to back while all of their positions read exactly 1 -- no single-voxel read in between -- were
        // measured in round 5: config 3's ray cast 71 us with one, 79 with two, 95 with four, 120 with eight: the lanes of the wave that
        // are near the surface wait through the others' runs.  profiles/r5_raycast_notes.md)
        const bool far = st == MARCH && found && sdf == 1.0f;      // SDF_valueToFloat(32767) is exactly 1
        if (__any(far)) {
          bool ended = false;
          (void)far_run<VX, LOOKAHEAD>(vol, far, px, py, pz, total, dx, dy, dz, stepScale, totalMax, ended);
          if (ended) st = DONE;
        }
      }
      if constexpr (!DENSE && (PARK || LOOKAHEAD > 0)) missStreak = found ? 0 : missStreak + 1;
      if constexpr (!DENSE && PARK && kProbeCells > 0) {
        // phase 1, once the wave has nothing left to march but rays inside "not found" runs (the other lanes are done or wait for their
        // trilinear read): those rays look kProbeCells positions ahead together, one round trip for the whole wave
        const bool runner = st == MARCH && missStreak >= kProbeAt;
        if (vol.dirPtr && __all(st != MARCH || runner) && __any(runner)) {
          const int adv = miss_run(std::integral_constant<int, kProbeCells>(), runner);
          if (runner) missStreak += adv;
        }
      }
      if constexpr (!DENSE && PARK) {
        if (st == MARCH && missStreak >= kParkStreak) { parked = true; st = DONE; }
      }
      if constexpr (!DENSE && LOOKAHEAD > 0) {
        if (vol.dirPtr && __any(missStreak >= kParkStreak && st == MARCH))
          (void)miss_run(std::integral_constant<int, (LOOKAHEAD > 0 ? LOOKAHEAD : 1)>(), missStreak >= kParkStreak && st == MARCH);
      }
    }
    ITM_WT(const unsigned long long wtA = wt_clock(); const unsigned wtTriLanes = __popcll(__ballot(st == TRI || st == REFINE)); wtInner = (unsigned)wt_wave_max(wtInner);)
    // ---- expensive phase: one 2x2x2 fetch for every lane that waits for one ---------------------------------------------
    if (st == TRI || st == REFINE) {
      Corners<VX, DENSE> cn;
      cn.fetch(vol, px, py, pz, cache);
      if (st == REFINE) {
        const float step = cn.trilinear() * stepScale;
        px += step * dx; py += step * dy; pz += step * dz;
        w = 1.0f; st = DONE;
      } else {
        st = advance(true, cn.trilinear());
      }
    }
    ITM_WT({ const float keep2 = total + px; asm volatile("" :: "v"(keep2)); const unsigned long long tB = wt_clock();
      const unsigned wv = blockIdx.x * 4 + (threadIdx.x >> 6);
#ifdef ITM_EXP_TRACE_TILE0
      // trace the four waves of 40 consecutive tiles starting at ITM_EXP_TRACE_TILE0, first pass only
      const bool wtPick = !resume && wv >= 4u * ITM_EXP_TRACE_TILE0 && wv < 4u * ITM_EXP_TRACE_TILE0 + 160u;
      const unsigned wtSlot = wv - 4u * ITM_EXP_TRACE_TILE0;
#else
      const bool wtPick = (wv & 31) == 1 && (wv >> 5) < 160;
      const unsigned wtSlot = wv >> 5;
#endif
      if (wtPick && wtOuter < 64) { unsigned long long* tr = g_waveTrace + ((size_t)wtSlot * 64 + wtOuter) * 4; tr[0] = wt0 - wtStart; tr[1] = wtA - wtStart; tr[2] = tB - wtStart; tr[3] = wtLanes0 | (wtMarch0 << 8) | (wtTriLanes << 16) | (wtInner << 24); }
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
  if (PARK && parked) return make_float4(px, py, pz, total);
  return make_float4(px, py, pz, w);
}

// castRay for callers that want one ray start to finish
template <class VX, bool DENSE, class VOL = VolumeView>
__device__ inline float4 cast_ray(int x, int y, const VOL& vol, const RayParams& p, float2 mm) {
  bool parked;
  return march_ray<VX, DENSE, 0, false>(x, y, vol, p, mm, nullptr, parked);
}

}  // namespace itm
