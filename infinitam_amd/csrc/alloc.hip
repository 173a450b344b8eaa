// alloc.hip -- voxel-block allocation and visible-list construction for the hash index.
//
// Reference behaviour (sequential CPU engine = the parity target):
//   AllocateSceneFromDepth            DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:116-291
//   buildHashAllocAndVisibleTypePP    DeviceAgnostic/ITMSceneReconstructionEngine.h:141-241
//   checkBlockVisibility<false>       DeviceAgnostic/ITMSceneReconstructionEngine.h:243-342
//   FindVisibleBlocks                 DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:39-77
//
// MI355X design (not the reference's, whose CUDA twin hands out blocks through atomicSub and is
// therefore non-deterministic):
//   1. request_kernel        : one lane per depth pixel (16x4 pixels per wave, coalesced depth
//                              reads) walks its ray segment.  A block that exists is found through
//                              the slot directory (one coherent load) or by probing the table; a
//                              missing block is requested by atomicMax of the key (pixel, step)+1 on
//                              the target slot; the largest key is exactly the "last writer in raster
//                              order, then step order" of the sequential loop.  The first requester
//                              of a slot also bumps the per-chunk request counters.  "Last frame's
//                              visible list -> type 3" is folded into the type encoding (a touched
//                              bit); mark_previous_kernel remains for callers that rewrote the list.
//   2. visible_list_kernel   : ONE launch per 2048-slot chunk and workgroup:
//      a. sweep_chunk        : ranks in ascending slot order come from the per-chunk counters + a
//                              workgroup scan, so pointers are handed out exactly as the sequential
//                              sweep does (vbaIdx = lastFree - rank).  Only the winning key's ray is
//                              recomputed to get the block coords.  Also fills the block directory,
//                              the slot directory and the sdf mirror for the new block.
//      b. visible list       : frustum re-test of type-3 slots and an ORDERED compaction (ascending
//                              slot ids); counts travel between workgroups as 8-byte granules.
//      (allocate_sweep_kernel / visible_count_kernel / visible_compact_kernel: the same steps as
//      separate launches, kept for FindVisibleBlocks and behind test hooks.)
// No host synchronisation: all counts stay in HBM.
#include <cstdlib>
#include <cstring>

#include "itm_internal.h"
#include "alloc_device.h"
#include "wave_utils.h"

namespace itm {

int g_debug_explicit_mark = 0;   // test hook: always run the explicit mark-previous launch
int g_debug_two_pass_visible_list = 0;   // test hook: count and compact as two launches
int g_debug_separate_sweep = 0;          // test hook (key 13): the allocation sweep as its own launch
int g_debug_force_list_stuck = 0;        // test hook (key 20): chunk n - 1 of the one-launch list behaves as if its bounded wait had expired

// (When the last list build found more visible slots than the list holds -- the reference then writes past its list, SURVEY section 7
// trap 6; list and count are clamped here -- every slot that was visible counts as "visible in the previous frame", listed or not:
// the rule the folded form of this pass, request_kernel<.., LAZY>, applies by construction, and the oracle's.)
__global__ void __launch_bounds__(256) mark_previous_kernel(const int32_t* __restrict__ ids, RenderCounters* __restrict__ rc,
                                                            uint8_t* __restrict__ visT, int noTotalEntries, int capIds) {
  if (rc->rawVisibleCount > capIds) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < noTotalEntries; i += gridDim.x * blockDim.x) if (visT[i]) visT[i] = 3;
    return;
  }
  const int nv = rc->noVisibleEntries;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) visT[ids[i]] = 3;
}

// One workgroup = 16x16 pixels, one wave = 16x4 pixels (the body: alloc_device.h).
template <bool ONLY_VISIBLE, bool FUSE_RANGE_INIT, bool LAZY>
__global__ void __launch_bounds__(256) request_kernel(RequestArgs a, AllocParams p) {
  request_tile<ONLY_VISIBLE, FUSE_RANGE_INIT, LAZY>(blockIdx.x, blockIdx.y, a, p);
}

// Block coordinates requested by (pixel, step): replays the winner's ray with identical arithmetic.
// (`d` = the depth pixel of the key, depth[(key - 1) >> stepBits], fetched ahead by the caller)
__device__ inline void replay_block_pos(uint32_t key, float d, const AllocParams& p, int& bx, int& by, int& bz) {
  const uint32_t k = key - 1u;
  const int loc = (int)(k >> p.stepBits);
  const int step = (int)(k & ((1u << p.stepBits) - 1u));
  const int y = loc / p.W, x = loc - y * p.W;
  BlockRay r;
  make_block_ray(d, x, y, p, r);
  for (int i = 0; i < step; ++i) { r.px += r.dx; r.py += r.dy; r.pz += r.dz; }
  bx = (int)(int16_t)(int)floorf(r.px); by = (int)(int16_t)(int)floorf(r.py); bz = (int)(int16_t)(int)floorf(r.pz);
}

constexpr int kSlotsPerThread = kSweepChunk / 256;  // 8

// Ascending-slot allocation sweep (_CPU.cpp:175-227).  chunkReq holds, per 2048-slot chunk, the
// number of requested slots and of excess-list requests; next-frame counters are zeroed here.
// The sweep of ONE chunk by its workgroup.  ACROSS: the visible type of a new excess entry lies in another chunk; when the visible
// list is built by the same launch (visible_list_kernel<.., SWEEP>) the workgroup of that chunk reads it there, so it is written
// with a device-scope store that bypasses the non-coherent L2s.
template <bool ACROSS>
__device__ inline void sweep_chunk(const int chunk, int* lds, uint32_t* __restrict__ allocKey, const int2* __restrict__ chunkReq,
                                   int2* __restrict__ chunkReqNext, int numChunks, uint4* __restrict__ hash,
                                   const int32_t* __restrict__ excessList, const int32_t* __restrict__ allocList,
                                   uint8_t* __restrict__ visT, const SceneCounters* __restrict__ counters,
                                   uint32_t* __restrict__ headBits, int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, void* __restrict__ sdfMirror,
                                   const float* __restrict__ depth, int lazy, const AllocParams& p) {

  const int tid = threadIdx.x;
  if (tid == 0) chunkReqNext[chunk] = make_int2(0, 0);
  const int2 mine = chunkReq[chunk];
  if (mine.x == 0) return;  // nothing requested in this chunk (uniform per workgroup)

  // The sweep of a chunk with requests is a chain of dependent loads (keys -> entry / depth pixel -> free-list slot), and the
  // whole visible-list launch waits for the slowest such chunk (per-workgroup timeline, tools/list_timeline.py: 8.9 us for it, 1 us
  // for the others).  The loads are therefore issued in three rounds of independent ones, unconditionally from addresses that are
  // always valid, instead of one by one inside the per-request branches.
  // ---- round 1: the keys of this thread's slots, the pool counters, the request counts of the earlier chunks ----
  const int slot0 = chunk * kSweepChunk + tid * kSlotsPerThread;
  const bool inTable = slot0 < p.noTotalEntries;              // noTotalEntries is a multiple of 8 (checked on the host)
  const uint4 ka = *(const uint4*)(allocKey + (inTable ? slot0 : 0)), kb = *(const uint4*)(allocKey + (inTable ? slot0 + 4 : 0));
  const int lastFreeVBA = counters->lastFreeBlockId;
  const int lastFreeExc = counters->lastFreeExcessListId;
  int b1 = 0, b2 = 0;
  for (int j = tid; j < chunk; j += 256) { int2 c = chunkReq[j]; b1 += c.x; b2 += c.y; }
  const uint32_t keep = inTable ? 0xffffffffu : 0u;
  const uint32_t keys[kSlotsPerThread] = {ka.x & keep, ka.y & keep, ka.z & keep, ka.w & keep, kb.x & keep, kb.y & keep, kb.z & keep, kb.w & keep};
  // ---- round 2: for every request its target entry (ordered or excess request?) and the depth pixel of the winning ray ----
  int ptrOfTarget[kSlotsPerThread];
  float depthOfKey[kSlotsPerThread];
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    ptrOfTarget[k] = (int)((const uint32_t*)&hash[keys[k] ? slot0 + k : 0])[3];
    depthOfKey[k] = depth[keys[k] ? (int)((keys[k] - 1u) >> p.stepBits) : 0];
  }
  const int baseReq = block_reduce_sum<4>(b1, lds);
  const int baseExc = block_reduce_sum<4>(b2, lds + 4);
  uint32_t isExcessBits = 0;
  int n1 = 0, n2 = 0;
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    if (keys[k]) {
      ++n1;
      // target of an excess request is an occupied chain tail; of an ordered request an empty head
      if (ptrOfTarget[k] >= -1) { isExcessBits |= (1u << k); ++n2; }
    }
  }
  int tot;
  int r1 = baseReq + block_exclusive_scan<4>(n1, lds, &tot);
  int r2 = baseExc + block_exclusive_scan<4>(n2, lds + 4, &tot);
  if (n1 == 0) return;
  // ---- round 3: the free-list entries the ranks select (ranks advance as in the sequential sweep, also past an empty pool) ----
  int vbaIdx[kSlotsPerThread], exlIdx[kSlotsPerThread], ptrNew[kSlotsPerThread], offNew[kSlotsPerThread];
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    vbaIdx[k] = -1; exlIdx[k] = -1;
    if (keys[k]) {
      vbaIdx[k] = lastFreeVBA - r1; ++r1;
      if (isExcessBits & (1u << k)) { exlIdx[k] = lastFreeExc - r2; ++r2; }
    }
  }
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    ptrNew[k] = allocList[vbaIdx[k] >= 0 ? vbaIdx[k] : 0];
    offNew[k] = excessList[exlIdx[k] >= 0 ? exlIdx[k] : 0];
  }
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    if (!keys[k]) continue;
    const int slot = slot0 + k;
    if (isExcessBits & (1u << k)) {
      if (vbaIdx[k] >= 0 && exlIdx[k] >= 0) {
        int bx, by, bz;
        replay_block_pos(keys[k], depthOfKey[k], p, bx, by, bz);
        const int off = offNew[k];
        ((uint32_t*)&hash[slot])[2] = (uint32_t)(off + 1);                 // connect the chain tail to the child
        const int ptr = ptrNew[k];
        hash[p.bucketNum + off] = pack_entry(bx, by, bz, 0, ptr);
        if (ACROSS) __hip_atomic_store(&visT[p.bucketNum + off], (uint8_t)(lazy ? 0x81 : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else visT[p.bucketNum + off] = lazy ? 0x81 : 1;
        directory_insert(dirPtr, dirSlot, p.org, bx, by, bz, ptr, p.bucketNum + off);
        mirror_init_block(sdfMirror, p.mirrorFloat != 0, p.org, bx, by, bz);
      }
    } else if (vbaIdx[k] >= 0) {
      int bx, by, bz;
      replay_block_pos(keys[k], depthOfKey[k], p, bx, by, bz);
      const int ptr = ptrNew[k];
      hash[slot] = pack_entry(bx, by, bz, 0, ptr);
      atomicOr(&headBits[slot >> 5], 1u << (slot & 31));
      directory_insert(dirPtr, dirSlot, p.org, bx, by, bz, ptr, slot);
      mirror_init_block(sdfMirror, p.mirrorFloat != 0, p.org, bx, by, bz);
    }
    allocKey[slot] = 0u;
  }
}


// checkBlockVisibility<false>: corners are reached by incremental +-f updates in a fixed order.
__device__ inline bool corner_in_image(const Mat4& M, float fx, float fy, float cx, float cy, int W, int H, float x, float y, float z) {
  Vec3 q = transform_point(M, x, y, z);
  if (q.z < 1e-10f) return false;
  float u = fx * q.x / q.z + cx;
  float v = fy * q.y / q.z + cy;
  return (u >= 0 && u < W && v >= 0 && v < H);
}
__device__ inline bool block_in_frustum(int bx, int by, int bz, const Mat4& M, float fx, float fy, float cx, float cy, float voxelSize, int W, int H) {
  const float f = (float)kBlockSide * voxelSize;
  float x = (float)bx * f, y = (float)by * f, z = (float)bz * f;
  if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  z += f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  y += f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  x += f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  z -= f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  y -= f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  x -= f; y += f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  x += f; y -= f; z += f; if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  return false;
}

// The same walk as a LOOP, for the visible-list launch: that kernel re-tests up to sixteen slots per lane, and with the walk unrolled at
// every one of them it was 16 000 instructions (~100 KB of code for a 64 KB instruction cache shared by two compute units); rolled it
// holds the corner arithmetic twice.  Steps are 6-bit codes (x+ x- y+ y- z+ z-), uniform: scalar branches.
__device__ inline bool block_in_frustum_rolled(int bx, int by, int bz, const Mat4& M, float fx, float fy, float cx, float cy, float voxelSize, int W, int H) {
  const float f = (float)kBlockSide * voxelSize;
  float x = (float)bx * f, y = (float)by * f, z = (float)bz * f;
  constexpr unsigned long long kSteps = (16ull << 6) | (4ull << 12) | (1ull << 18) | (32ull << 24) | (8ull << 30) | (6ull << 36) | (25ull << 42);
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const uint32_t op = (uint32_t)(kSteps >> (6 * i)) & 63u;
    if (op & 1u) x += f;
    if (op & 2u) x -= f;
    if (op & 4u) y += f;
    if (op & 8u) y -= f;
    if (op & 16u) z += f;
    if (op & 32u) z -= f;
    if (corner_in_image(M, fx, fy, cx, cy, W, H, x, y, z)) return true;
  }
  return false;
}

// The frustum re-tests of the excess region, dealt out to the region's workgroups (visible_list_kernel explains why): group G of 64
// consecutive slots is re-tested by wave (G / E) % 4 of workgroup G % E in its round G / (4 E), one block per lane.  The verdicts go
// to the owners of the slots as tagged granules -- two 8-byte stores per group.  A verdict is a function of what the slot held when
// the launch began; the OWNER decides whether it applies: a slot that a sweep of this launch has filled meanwhile carries the touched
// mark by the time the owner reads it (stale types of a render state that outlived a ResetScene sit on empty slots), and its verdict
// -- possibly computed from a half-written entry -- is ignored.  Lazy form of the types only; called by whole workgroups.
__device__ inline void share_excess_retests(int chunk, int numChunks, uint8_t* __restrict__ visT, const uint4* __restrict__ hash, const AllocParams& p,
                                            unsigned long long* __restrict__ keptGran, uint32_t epoch) {
  const int tid = threadIdx.x;
  const int firstEx = p.bucketNum / kSweepChunk, E = numChunks - firstEx, e = chunk - firstEx;
  const int regionSlots = p.noTotalEntries - p.bucketNum;
  int sl[kSlotsPerThread];
  uint32_t cand = 0;
#pragma unroll
  for (int r = 0; r < kSlotsPerThread; ++r) {
    const int rs = ((r * 4 + (tid >> 6)) * E + e) * kWave + (tid & (kWave - 1));
    sl[r] = p.bucketNum + (rs < regionSlots ? rs : 0);
    // (plain loads: a type byte that another workgroup rewrites meanwhile is one of a slot that was empty when the launch began)
    const uint32_t t = visT[sl[r]];
    if (rs < regionSlots && t != 0u && !(t & 0x80u)) cand |= 1u << r;
  }
  // (the entries are asked for with the types, not behind them: one memory round trip instead of two; most go unused)
  uint4 ent[kSlotsPerThread];
#pragma unroll
  for (int r = 0; r < kSlotsPerThread; ++r) ent[r] = hash[sl[r]];
#pragma unroll 1
  for (int r = 0; r < kSlotsPerThread; ++r) {
    bool kept = false;
    if (__ballot((cand >> r) & 1u) != 0ull) {                    // (uniform; most rounds of most waves have nothing to test)
      uint4 er = ent[0];
#pragma unroll
      for (int j = 1; j < kSlotsPerThread; ++j) if (r == j) er = ent[j];      // r is uniform: selects, no indexed registers
      if ((cand >> r) & 1u) {
        const HashEntry he = unpack_entry(er);
        kept = block_in_frustum_rolled(he.px, he.py, he.pz, p.M, p.fx, p.fy, p.cx, p.cy, p.voxelSize, p.W, p.H);
      }
    }
    const unsigned long long mask = __ballot(kept);
    if ((tid & (kWave - 1)) < 2) {
      const int G = (r * 4 + (tid >> 6)) * E + e;
      const uint32_t half = (tid & 1) ? (uint32_t)(mask >> 32) : (uint32_t)mask;
      __hip_atomic_store(&keptGran[2 * G + (tid & 1)], ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void __launch_bounds__(256) allocate_sweep_kernel(uint32_t* __restrict__ allocKey, const int2* __restrict__ chunkReq,
                                                             int2* __restrict__ chunkReqNext, int numChunks, uint4* __restrict__ hash,
                                                             const int32_t* __restrict__ excessList, const int32_t* __restrict__ allocList,
                                                             uint8_t* __restrict__ visT, const SceneCounters* __restrict__ counters,
                                                             uint32_t* __restrict__ headBits, int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, void* __restrict__ sdfMirror,
                                                             const float* __restrict__ depth, int lazy, AllocParams p) {
  __shared__ int lds[8];
  // (The excess region's re-tests are NOT shared here although the idle workgroups are the same: with the count in a later launch the
  // verdicts would have to travel in the type bytes, and after ResetScene the render state's stale types sit on EMPTY slots -- slots
  // that this very launch may fill and mark "new", racing with the verdict's store.  Built, measured at +1 % for two or three scenes
  // per process, and caught by the second-life test one run in twelve; removed.)
  sweep_chunk<false>((int)blockIdx.x, lds, allocKey, chunkReq, chunkReqNext, numChunks, hash, excessList, allocList, visT, counters, headBits, dirPtr, dirSlot, sdfMirror, depth, lazy, p);
}

// checkBlockVisibility<true> (DeviceAgnostic/ITMSceneReconstructionEngine.h:243-342): a corner outside the image but inside the image
// enlarged by an eighth on every side makes the block "visible enlarged"; the walk ends at the first corner inside the image proper
__device__ inline bool block_in_frustum_enlarged(int bx, int by, int bz, const Mat4& M, float fx, float fy, float cx, float cy, float voxelSize, int W, int H) {
  const float f = (float)kBlockSide * voxelSize;
  float x = (float)bx * f, y = (float)by * f, z = (float)bz * f;
  bool enlarged = false;
  auto corner = [&]() -> bool {
    Vec3 q = transform_point(M, x, y, z);
    if (q.z < 1e-10f) return false;
    const float u = fx * q.x / q.z + cx, v = fy * q.y / q.z + cy;
    if (u >= 0 && u < W && v >= 0 && v < H) { enlarged = true; return true; }
    const int lx = -W / 8, hx = W + W / 8, ly = -H / 8, hy = H + H / 8;
    if (u >= lx && u < hx && v >= ly && v < hy) enlarged = true;
    return false;
  };
  if (corner()) return true;
  z += f; if (corner()) return true;
  y += f; if (corner()) return true;
  x += f; if (corner()) return true;
  z -= f; if (corner()) return true;
  y -= f; if (corner()) return true;
  x -= f; y += f; if (corner()) return true;
  x += f; y -= f; z += f; if (corner()) return true;
  return enlarged;
}

// Pass 1 of the visible list (_CPU.cpp:229-269): re-test type-3 slots, count visible slots per
// chunk.  Workgroup 0 also commits the pool counters of the allocation sweep (:287-290).
template <bool COMMIT_ALLOC, bool LAZY>
__global__ void __launch_bounds__(256) visible_count_kernel(uint8_t* __restrict__ visT, const uint4* __restrict__ hash,
                                                            int32_t* __restrict__ chunkVis, const int2* __restrict__ chunkReq,
                                                            int numChunks, SceneCounters* __restrict__ counters, AllocParams p) {
  __shared__ int lds[8];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int slot0 = chunk * kSweepChunk + tid * kSlotsPerThread;
  int n = 0;
  if (slot0 < p.noTotalEntries) {  // noTotalEntries is a multiple of 8 (checked on the host)
    uint2 raw = *(const uint2*)(visT + slot0);
    if (raw.x | raw.y) {
      uint32_t w[2] = {raw.x, raw.y};
      bool changed = false;
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        uint32_t t = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        const uint32_t t0 = t;
        if (LAZY && (t & 0x80u)) {
          t &= 0x7fu;                         // touched this frame: type 1 / 2
        } else if (LAZY ? (t != 0u) : (t == 3u)) {
          // visible in the previous frame and not seen again: keep only if still in the frustum
          HashEntry he = unpack_entry(hash[slot0 + k]);
          const bool keep = p.useSwapping ? block_in_frustum_enlarged(he.px, he.py, he.pz, p.M, p.fx, p.fy, p.cx, p.cy, p.voxelSize, p.W, p.H)
                                          : block_in_frustum(he.px, he.py, he.pz, p.M, p.fx, p.fy, p.cx, p.cy, p.voxelSize, p.W, p.H);
          t = keep ? 3u : 0u;
        }
        if (t != t0) { w[k >> 2] = (w[k >> 2] & ~(0xffu << ((k & 3) * 8))) | (t << ((k & 3) * 8)); changed = true; }
        n += (t > 0u);
      }
      if (changed) *(uint2*)(visT + slot0) = make_uint2(w[0], w[1]);
    }
  }
  const int sum = block_reduce_sum<4>(n, lds);
  if (tid == 0) chunkVis[chunk] = sum;
  if (COMMIT_ALLOC && chunk == 0) {
    int a = 0, b = 0;
    for (int j = tid; j < numChunks; j += 256) { int2 c = chunkReq[j]; a += c.x; b += c.y; }
    a = block_reduce_sum<4>(a, lds);
    b = block_reduce_sum<4>(b, lds + 4);
    if (tid == 0) {
      counters->lastFreeBlockId -= a;
      counters->lastFreeExcessListId -= b;
      counters->noAllocRequests = a;
    }
  }
}

// Pass 2: ordered compaction of slots flagged in `flags` (non-zero byte) into ascending ids.
__global__ void __launch_bounds__(256) visible_compact_kernel(const uint8_t* __restrict__ flags, const int32_t* __restrict__ chunkVis,
                                                              int numChunks, int noTotalEntries, int32_t* __restrict__ ids, int capIds,
                                                              RenderCounters* __restrict__ rc) {
  __shared__ int lds[8];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int mine = chunkVis[chunk];
  if (chunk != 0 && mine == 0) return;
  int b = 0, all = 0;
  for (int j = tid; j < numChunks; j += 256) { int c = chunkVis[j]; all += c; if (j < chunk) b += c; }
  const int base = block_reduce_sum<4>(b, lds);
  if (chunk == 0) {
    const int total = block_reduce_sum<4>(all, lds + 4);
    if (tid == 0) { rc->rawVisibleCount = total; rc->noVisibleEntries = total < capIds ? total : capIds; }
    if (mine == 0) return;
  }
  const int slot0 = chunk * kSweepChunk + tid * kSlotsPerThread;
  uint32_t w[2] = {0u, 0u};
  if (slot0 < noTotalEntries) { uint2 raw = *(const uint2*)(flags + slot0); w[0] = raw.x; w[1] = raw.y; }
  int n = 0;
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) n += (((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0u);
  int tot;
  int pos = base + block_exclusive_scan<4>(n, lds, &tot);
  if (n == 0) return;
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    if (((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0u) {
      if (pos < capIds) ids[pos] = slot0 + k;
      ++pos;
    }
  }
}

// Passes 1 and 2 in ONE launch (AllocateSceneFromDepth): every workgroup counts its chunk as visible_count_kernel does, publishes
// the count as an 8-byte {count, epoch} granule with a single device-scope store, then reads the granules of the chunks before
// it (device-scope loads, polling until the epoch matches) for its base and compacts from the types it still holds in
// registers.  A granule is one naturally aligned 8-byte store, so value and tag arrive together and no fence is needed; only
// counts cross workgroups, the types and ids a workgroup writes are read by later launches.  All chunks' workgroups are
// resident at once (numChunks * 4 waves << the chip's wave slots), and a workgroup only waits for lower-numbered ones; the poll
// is bounded all the same and raises statusFlags bit 1 instead of hanging.
//
// SWEEP: the allocation sweep of the chunk runs first, in the same workgroup (one launch less per frame).  Ordered allocations only
// touch the chunk's own slots.  An EXCESS allocation writes the visible type of an entry in the excess region, i.e. in another
// chunk: a chunk with excess requests stamps sweepDone[chunk] with the launch's epoch once its stores have completed, and the
// workgroup of an excess-region chunk waits for the stamps of all such chunks before it counts, then reads its types with a
// device-scope load (the sweep wrote them with one).  No sweep waits for anything, so the waiting cannot cycle; in a frame without
// excess requests -- almost all of them -- nobody waits.  The pool counters, which every sweep reads, are committed by the LAST
// chunk after its look-back (all granules in = all sweeps done) instead of by chunk 0.
struct SweepArgs {
  uint32_t* allocKey; int2* chunkReqNext; const int32_t* excessList; const int32_t* allocList; uint32_t* headBits;
  int32_t* dirPtr; int32_t* dirSlot; void* sdfMirror; const float* depth; int lazy;
  uint32_t* sweepDone;     // per chunk: the epoch of the launch whose sweep has placed the chunk's excess allocations
  unsigned long long* keptGran;      // per 32 slots of the excess region: {epoch, "kept" bits} of the shared re-tests
  int32_t* fatalDev;       // the scene's host-visible status word (alloc_device.h: raise_fatal)
  int forceStuck;          // test hook (debug key 20): chunk whose wait is treated as expired, or -1
};

constexpr int kListSpinSleep = 1;    // s_sleep argument between two polls of a predecessor's granule in the look-back (x 64 cycles)
#ifndef ITM_EXP_LIST_STAMPS
#define ITM_EXP_LIST_STAMPS 0     // measurement build: per-workgroup timeline of the visible-list launch (100 MHz clock)
#endif
#if ITM_EXP_LIST_STAMPS
__device__ unsigned long long g_listStamps[1024 * 6];
#define ITM_LS(k) if (threadIdx.x == 0 && blockIdx.x < 1024) g_listStamps[6 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime();
extern "C" int itm_debug_read_list_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_listStamps), (size_t)n * 8); }
#else
#define ITM_LS(k)
#endif

// WHO WAITS FOR WHOM, AND WHEN THAT IS SAFE.  The look-back waits for lower-numbered chunks only; workgroups are dispatched in index order
// and run to their end, so that wait cannot be stuck behind the waiter.  The stamps are different: an excess-region chunk waits for the
// sweep of EVERY chunk with excess requests, and such a chunk may lie behind it in the excess region (a request whose chain tail is an
// excess entry) -- safe only while all workgroups of the launch are resident TOGETHER.  That is a property of the device, and it is
// checked, not assumed: one_pass_list_is_safe() asks the runtime how many workgroups of this kernel a compute unit holds
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) and takes this launch only if units x that covers the grid -- a partitioned or masked
// device gets the launches that never wait (separate sweep, count, compaction), like a device shared by several scenes.  The waits
// stay bounded all the same; one that does expire (another PROCESS holding the compute units) is fatal for the scene: statusFlags
// bit 1, ITM_ERR_DEVICE at the next call, the frame's list marked invalid and nothing fused through it -- never a silently dropped frame.
// (Round 4 also built the two shapes that make every wait target running work by construction -- 64 extra workgroups at the head of the
// grid that sweep the excess region, and sweeps of the excess region CLAIMED by whoever needs them first -- and measured them at
// +1.5 us and +5.5 us per frame on BASELINE configs[1]: the loop around the sweep that the claim needs makes the compiler drain the
// early loads before it, 13.2 -> 19.3 us.  profiles/r4_integrate_notes.md.)
template <bool COMMIT_ALLOC, bool LAZY, bool SWEEP>
__global__ void __launch_bounds__(256) visible_list_kernel(uint8_t* __restrict__ visT, uint4* __restrict__ hash,
                                                           unsigned long long* __restrict__ chunkGran, uint32_t epoch,
                                                           const int2* __restrict__ chunkReq, int numChunks, SceneCounters* __restrict__ counters,
                                                           int32_t* __restrict__ ids, int capIds, RenderCounters* __restrict__ rc, AllocParams p, SweepArgs sw) {
  __shared__ int lds[12];
  __shared__ int sweepLds[8];
  const int tid = threadIdx.x;
  // the chunk's stores must have COMPLETED before its stamp may follow: on gfx950 a workgroup-scope release fence is only
  // s_waitcnt lgkmcnt(0) -- it does not wait for vector stores -- so the wait is spelled out (loads and stores share vmcnt);
  // the type bytes and entries were stored with device scope (write-through), nothing is left to write back
  auto stamp_sweep = [&](int c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();                                         // ... all of the chunk's have: say that its excess allocations are in place
    if (tid == 0) __hip_atomic_store(&sw.sweepDone[c], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const int chunk = (int)blockIdx.x;
  const int slot0 = chunk * kSweepChunk + tid * kSlotsPerThread;
  const bool excessRegion = SWEEP && slot0 - tid * kSlotsPerThread >= p.bucketNum;      // uniform: bucketNum is a multiple of the chunk size (host)
  ITM_LS(0)
  int before = 0;
  bool stuck = SWEEP && chunk == sw.forceStuck;
  if constexpr (SWEEP && LAZY) {
    if (excessRegion) {
      // THE EXCESS REGION'S RE-TESTS, SHARED.  Excess entries are handed out from the top of the region downwards, so the ones in use
      // lie side by side in its LAST chunk (BASELINE configs[1]: 1 732 entries, 1 518 of them in the visible list), and whenever the
      // camera turns away from the pixels that request them that one workgroup walked the corners of eight blocks per lane, one after
      // the other -- 16 us of a launch whose other 575 workgroups are through after 7 (tools/list_timeline.py; the 20-24 us launches
      // of profiles/r5_counters.md; BASELINE configs[4]: 36 us every frame).  The candidates of a frame are known before the launch
      // starts (previously visible, no request this frame: nothing in this launch changes that -- a sweep only fills empty slots), so
      // the excess-region workgroups -- 63 of 64 idle -- share them before anything else, and the 64 verdicts of a group travel to the
      // chunk that owns the slots as two tagged granules {epoch, 32 "kept" bits}: one indivisible 8-byte store each, nothing to wait
      // for behind them (the look-back's idiom).  The owner's count picks its lanes' bytes out of them instead of walking (below).
      share_excess_retests(chunk, numChunks, visT, hash, p, sw.keptGran, epoch);
    }
  }
  // sums the granules of the chunks before this one, waiting for each to carry this launch's epoch
  auto look_back = [&]() {
    for (int j = tid; j < chunk; j += 256) {
      unsigned long long g = __hip_atomic_load(&chunkGran[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spin = 0; (uint32_t)(g >> 32) != epoch; ++spin) {
        if (spin > (1 << 22)) { stuck = true; break; }
        __builtin_amdgcn_s_sleep(kListSpinSleep);
        g = __hip_atomic_load(&chunkGran[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      before += (int)(uint32_t)g;
    }
  };
  // An ordered-region chunk's visible types and the entries its count re-tests do not depend on its own sweep (the sweep fills EMPTY
  // slots whose types the request stage has set, the re-tested entries are those of the previous frame's list): they are requested
  // here, ahead of the sweep's chain of dependent rounds, and have arrived when the count needs them -- for the chunk with the most
  // requests, which every later chunk's look-back waits for, two round trips less behind its sweep.
  const bool early = SWEEP && !excessRegion && slot0 < p.noTotalEntries;
  uint2 rawEarly = make_uint2(0u, 0u);
  uint4 entryEarly[kSlotsPerThread];
  if (early) {
    rawEarly = *(const uint2*)(visT + slot0);
    if (rawEarly.x | rawEarly.y) {
      uint32_t retest = 0;
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        const uint32_t t = ((k < 4 ? rawEarly.x : rawEarly.y) >> ((k & 3) * 8)) & 0xffu;
        if (!(LAZY && (t & 0x80u)) && (LAZY ? (t != 0u) : (t == 3u))) retest |= 1u << k;
      }
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) entryEarly[k] = hash[slot0 + ((retest >> k) & 1u ? k : 0)];
    }
  }
  if constexpr (SWEEP) {
    sweep_chunk<true>(chunk, sweepLds, sw.allocKey, chunkReq, sw.chunkReqNext, numChunks, hash, sw.excessList, sw.allocList, visT, counters, sw.headBits,
                      sw.dirPtr, sw.dirSlot, sw.sdfMirror, sw.depth, sw.lazy, p);
    if (chunkReq[chunk].y > 0) stamp_sweep(chunk);           // (uniform) only excess allocations are read by other workgroups of this launch
    if (excessRegion) {
      // every chunk that had excess requests (any index: no sweep waits for anything, so this cannot cycle) must be through
      for (int j = tid; j < numChunks; j += 256) {
        if (chunkReq[j].y <= 0 || j == chunk) continue;
        for (int spin = 0; __hip_atomic_load(&sw.sweepDone[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin) {
          if (spin > (1 << 22)) { stuck = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    }
  }
  ITM_LS(1)
  int n = 0;
  uint32_t w[2] = {0u, 0u};
  if (slot0 < p.noTotalEntries) {  // noTotalEntries is a multiple of 8 (checked on the host)
    uint2 raw;
    unsigned long long keptEarly = 0;
    if (excessRegion) {
      const unsigned long long q = __hip_atomic_load((const unsigned long long*)(visT + slot0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the verdicts on this lane's slots, asked for beside the types: by now they have usually been published)
      if constexpr (SWEEP && LAZY) keptEarly = __hip_atomic_load(&sw.keptGran[(slot0 - p.bucketNum) >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      raw = make_uint2((uint32_t)q, (uint32_t)(q >> 32));
    } else if (early) {
      raw = rawEarly;
    } else {
      raw = *(const uint2*)(visT + slot0);
    }
    w[0] = raw.x; w[1] = raw.y;
    if (raw.x | raw.y) {
      bool changed = false;
      // the entries of the slots that need the frustum re-test, fetched together (a thread's eight entries are 128 contiguous
      // bytes) rather than one dependent load per slot inside the branch below
      uint32_t retest = 0;
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        const uint32_t t = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        if (!(LAZY && (t & 0x80u)) && (LAZY ? (t != 0u) : (t == 3u))) retest |= 1u << k;
      }
      constexpr bool kShared = SWEEP && LAZY;      // the excess region's re-tests were shared out at the top of the kernel
      uint4 entry[kSlotsPerThread];
      uint32_t kept8 = 0;
      if (kShared && excessRegion) {
        if (retest) {
          // this lane's eight verdicts: one byte of the granule pair of the 64-slot group its slots lie in
          const int rel = slot0 - p.bucketNum;
          const unsigned long long* gp = &sw.keptGran[rel >> 5];
          unsigned long long g = keptEarly;
          for (int spin = 0; (uint32_t)(g >> 32) != epoch; ++spin) {
            if (spin > (1 << 22)) { stuck = true; break; }
            __builtin_amdgcn_s_sleep(kListSpinSleep);
            g = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          kept8 = ((uint32_t)g >> (rel & 31)) & 0xffu;
        }
      } else {
        if (early) {
#pragma unroll
          for (int k = 0; k < kSlotsPerThread; ++k) entry[k] = entryEarly[k];
        } else {
#pragma unroll
          for (int k = 0; k < kSlotsPerThread; ++k) entry[k] = hash[slot0 + ((retest >> k) & 1u ? k : 0)];
        }
        // visible in the previous frame and not seen again: kept only if still in the frustum (one copy of the walk: a loop over the
        // lane's slots around a loop over the corners)
        if (retest) {
#pragma unroll 1
          for (int k = 0; k < kSlotsPerThread; ++k) {
            if (!((retest >> k) & 1u)) continue;
            uint4 ek = entry[0];
#pragma unroll
            for (int j = 1; j < kSlotsPerThread; ++j) if (k == j) ek = entry[j];
            const HashEntry he = unpack_entry(ek);
            if (block_in_frustum_rolled(he.px, he.py, he.pz, p.M, p.fx, p.fy, p.cx, p.cy, p.voxelSize, p.W, p.H)) kept8 |= 1u << k;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        uint32_t t = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        const uint32_t t0 = t;
        if (LAZY && (t & 0x80u)) {
          t &= 0x7fu;                         // touched this frame: type 1 / 2
        } else if (LAZY ? (t != 0u) : (t == 3u)) {
          t = ((kept8 >> k) & 1u) ? 3u : 0u;
        }
        if (t != t0) { w[k >> 2] = (w[k >> 2] & ~(0xffu << ((k & 3) * 8))) | (t << ((k & 3) * 8)); changed = true; }
        n += (t > 0u);
      }
      if (changed) *(uint2*)(visT + slot0) = make_uint2(w[0], w[1]);
    }
  }
  int mine;
  const int pos0 = block_exclusive_scan<4>(n, lds, &mine);
  if (tid == 0)
    __hip_atomic_store(&chunkGran[chunk], ((unsigned long long)epoch << 32) | (unsigned long long)(uint32_t)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ITM_LS(2)
  // base = visible slots in all earlier chunks
  look_back();
  ITM_LS(3)
  if (stuck) { raise_fatal(counters, sw.fatalDev, 2); rc->listInvalid = 1; }    // fatal for the scene (see above); the frame is not fused (integrate.hip)
  const int base = block_reduce_sum<4>(before, lds + 4);
  if (COMMIT_ALLOC && chunk == (SWEEP ? numChunks - 1 : 0)) {
    // (with SWEEP every sweep has read the pool counters by now: all granules are in)
    int a = 0, b = 0;
    for (int j = tid; j < numChunks; j += 256) { int2 c = chunkReq[j]; a += c.x; b += c.y; }
    __syncthreads();
    a = block_reduce_sum<4>(a, lds + 4);
    b = block_reduce_sum<4>(b, lds + 8);
    if (tid == 0) {
      counters->lastFreeBlockId -= a;
      counters->lastFreeExcessListId -= b;
      counters->noAllocRequests = a;
    }
  }
  if (chunk == numChunks - 1 && tid == 0) {
    const int total = base + mine;
    rc->rawVisibleCount = total;
    rc->noVisibleEntries = total < capIds ? total : capIds;
  }
  ITM_LS(4)
  ITM_LS(5)
  if (n == 0) return;
  int pos = base + pos0;
#pragma unroll
  for (int k = 0; k < kSlotsPerThread; ++k) {
    if (((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0u) {
      if (pos < capIds) ids[pos] = slot0 + k;
      ++pos;
    }
  }
}

// FindVisibleBlocks pass 1: flag every allocated slot whose block passes the frustum test.
__global__ void __launch_bounds__(256) freeview_flag_kernel(const uint4* __restrict__ hash, uint8_t* __restrict__ flags,
                                                            int32_t* __restrict__ chunkVis, AllocParams p) {
  __shared__ int lds[4];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int slot0 = chunk * kSweepChunk + tid * kSlotsPerThread;
  int n = 0;
  if (slot0 < p.noTotalEntries) {
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
      HashEntry he = unpack_entry(hash[slot0 + k]);
      bool vis = false;
      if (he.ptr >= 0) vis = block_in_frustum(he.px, he.py, he.pz, p.M, p.fx, p.fy, p.cx, p.cy, p.voxelSize, p.W, p.H);
      if (vis) { w[k >> 2] |= 1u << ((k & 3) * 8); ++n; }
    }
    *(uint2*)(flags + slot0) = make_uint2(w[0], w[1]);
  }
  const int sum = block_reduce_sum<4>(n, lds);
  if (tid == 0) chunkVis[chunk] = sum;
}

// Is this scene alone on its device?  (scene.hip keeps the count of live hash scenes.)
// ... and can all workgroups of the one-launch list be resident on it together?  (Asked of the runtime once per device and grid size:
// workgroups a compute unit holds of visible_list_kernel<true, true, true> x compute units >= the grid.)
static bool list_grid_is_resident(const itm_scene* s) {
  static int answer[64] = {};                  // per device: 0 unknown, chunks + 1 that fit
  const int dev = (s->device >= 0 && s->device < 64) ? s->device : 0;
  if (!answer[dev]) {
    int perCu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void*)visible_list_kernel<true, true, true>, 256, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); perCu = 0; cus = 0; }
    answer[dev] = perCu * cus + 1;
  }
  return answer[dev] - 1 >= s->numChunks;
}
static bool one_pass_list_is_safe(const itm_scene* s) {
  static const int forced = [] { const char* e = getenv("ITM_ONE_PASS_LIST"); return e ? atoi(e) : -1; }();      // A/B: 1 = always, 0 = never
  if (s->cfg.useSwapping) return false;        // the swapping hooks live in the separate launches
  if (forced >= 0) return forced != 0;
  return live_hash_scenes(s->device) <= 1 && list_grid_is_resident(s);
}

static int fill_params(const itm_scene* s, const float* M, const float* intr, int W, int H, int capIds, AllocParams& p) {
  memcpy(p.M.m, M, 64);
  if (!invert4(M, p.invM.m)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  p.fx = intr[0]; p.fy = intr[1]; p.cx = intr[2]; p.cy = intr[3];
  p.ifx = 1.0f / intr[0]; p.ify = 1.0f / intr[1];
  p.mu = s->prm.mu;
  p.voxelSize = s->prm.voxelSize;
  p.oneOverBlock = 1.0f / (s->prm.voxelSize * kBlockSide);
  p.vfmin = s->prm.viewFrustum_min; p.vfmax = s->prm.viewFrustum_max;
  p.W = W; p.H = H;
  p.mask = (uint32_t)s->cfg.bucketNum - 1u;
  p.bucketNum = s->cfg.bucketNum;
  p.noTotalEntries = s->noTotalEntries;
  p.capIds = capIds;
  p.mirrorFloat = (s->cfg.voxelType == ITM_VOXEL_F || s->cfg.voxelType == ITM_VOXEL_F_RGB) ? 1 : 0;
  p.org = s->org;
  p.useSwapping = s->cfg.useSwapping;
  int pixBits = 1;
  while ((1ll << pixBits) < (long long)W * H) ++pixBits;
  p.stepBits = 31 - pixBits;
  if (p.stepBits > 12) p.stepBits = 12;
  return ITM_OK;
}

// Stage 1 (reads the table, writes the request keys / visible types): can overlap the previous
// frame's integration and ray casting, which do not touch these buffers.
// Stage 1: per-pixel block requests.  When the visible list and the visible types are known to be
// coherent (always, unless the caller rewrote one of them) the "mark previous list as type 3" launch is
// skipped and folded into the type encoding (LAZY, see request_kernel).
// Everything of the request stage up to the launch: parameters, the acceleration cubes placed for this view, the request counters of
// this frame's parity.  `lazy`: the previous list need not be marked (see request_kernel).
int prepare_request_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st, AllocParams& p, RequestArgs& ra, bool& lazy) {
  int rc = fill_params(s, v->M_d, v->intr_d, v->w, v->h, rs->capIds, p);
  if (rc) return rc;
  if (p.stepBits < 4) return set_error(ITM_ERR_INVALID, "depth image too large for the allocation key");
  // the acceleration cubes follow the camera: placed by the first frame, moved (emptied and refilled from the table, on this stream)
  // when the view leaves them
  rc = accel_place(s, p.invM.m, st);
  if (rc) return rc;
  p.org = s->org;
  int2* reqCur = (int2*)s->chunkReq + (size_t)(s->frameParity & 1u) * s->numChunks;
  lazy = rs->listCoherent && !g_debug_explicit_mark;
  ra = RequestArgs{v->depth, s->hash, rs->visibleType, s->allocKey, reqCur, s->counters, rs->range, rs->counters, g_debug_no_directory ? nullptr : s->dirSlot, s->fatalDev};
  return ITM_OK;
}

static bool same_view(const itm_render_state* rs, const itm_view* v) {
  return rs->ahead.depth == v->depth && rs->ahead.w == v->w && rs->ahead.h == v->h && memcmp(rs->ahead.M_d, v->M_d, 64) == 0 && memcmp(rs->ahead.intr_d, v->intr_d, 16) == 0;
}

// Stage 1 (reads the table, writes the request keys / visible types): can overlap the previous
// frame's integration and ray casting, which do not touch these buffers.
// Stage 1: per-pixel block requests.  When the visible list and the visible types are known to be
// coherent (always, unless the caller rewrote one of them) the "mark previous list as type 3" launch is
// skipped and folded into the type encoding (LAZY, see request_kernel).
// What an allocation for view `v` would be refused for, checked without side effects (the entry point records the call, pending.hip,
// and must report these at once): a pose without inverse, an image too large for the request key, block requests of ANOTHER view
// issued ahead on this render state -- or of any view on another render state of the scene (the request keys belong to the scene).
int validate_allocate(const itm_scene* s, const itm_view* v, const itm_render_state* rs, bool onlyVisible) {
  float inv[16];
  if (!invert4(v->M_d, inv)) return set_error(ITM_ERR_INVALID, "pose matrix is singular");
  if ((long long)v->w * v->h > (1ll << 27)) return set_error(ITM_ERR_INVALID, "depth image too large for the allocation key");
  if (s->aheadRs && s->aheadRs != rs)
    return set_error(ITM_ERR_INVALID, "the scene holds the block requests of a frame issued ahead on another render state (itm_process_frame_ahead): fuse that frame or itm_cancel_ahead first");
  if (rs->ahead.valid) {
    if (onlyVisible || !same_view(rs, v))
      return set_error(ITM_ERR_INVALID, "the block requests of another view were issued ahead (itm_process_frame_ahead): the next allocation must be for that view");
    if (rs->ahead.tableEpoch != s->tableEpoch)
      return set_error(ITM_ERR_INVALID, "the scene was reset or its table replaced while the block requests of the next view were pending (itm_process_frame_ahead)");
  }
  return ITM_OK;
}

int launch_request_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, bool fuseRangeInit, hipStream_t st) {
  { const int rc = validate_allocate(s, v, rs, onlyVisible); if (rc) return rc; }
  if (rs->ahead.valid) {
    // the requests of this frame rode in the previous frame's last launch (itm_process_frame_ahead)
    rs->ahead.valid = false;
    if (s->aheadRs == rs) s->aheadRs = nullptr;
    rs->lazyThisFrame = rs->ahead.lazy;
    return ITM_OK;
  }
  AllocParams p; RequestArgs ra; bool lazy;
  int rc = prepare_request_stage(s, v, rs, st, p, ra, lazy);
  if (rc) return rc;
  rs->lazyThisFrame = lazy;
  if (!lazy) mark_previous_kernel<<<64, 256, 0, st>>>(rs->visibleIds, rs->counters, rs->visibleType, s->noTotalEntries, rs->capIds);
  dim3 grid((v->w + 15) / 16, (v->h + 15) / 16);
  KernelTimer tq(s, ITM_TK_REQUEST, st);
#define ITM_REQ(OV, FU, LZ) request_kernel<OV, FU, LZ><<<grid, 256, 0, st>>>(ra, p)
  if (onlyVisible) {
    if (fuseRangeInit) { if (lazy) ITM_REQ(true, true, true); else ITM_REQ(true, true, false); }
    else { if (lazy) ITM_REQ(true, false, true); else ITM_REQ(true, false, false); }
  } else {
    if (fuseRangeInit) { if (lazy) ITM_REQ(false, true, true); else ITM_REQ(false, true, false); }
    else { if (lazy) ITM_REQ(false, false, true); else ITM_REQ(false, false, false); }
  }
#undef ITM_REQ
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

// Stage 2 (writes the table and the visible list): ordered allocation sweep + visible list.
int launch_sweep_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, hipStream_t st) {
  AllocParams p;
  int rc = fill_params(s, v->M_d, v->intr_d, v->w, v->h, rs->capIds, p);
  if (rc) return rc;
  const int nChunks = s->numChunks;
  int2* reqCur = (int2*)s->chunkReq + (size_t)(s->frameParity & 1u) * nChunks;
  int2* reqNext = (int2*)s->chunkReq + (size_t)((s->frameParity + 1u) & 1u) * nChunks;
  const bool lazy = rs->lazyThisFrame;
  // the sweep rides in the visible-list launch unless a test hook asks for separate launches (or the ordered part of the table
  // does not end on a chunk boundary, which no configuration of the reference produces)
  // The one-launch list (and the sweep inside it) hand counts from workgroup to workgroup through memory, i.e. workgroups WAIT for
  // others of the same launch.  That is the fastest form while the scene has the device to itself.  With several scenes live on one
  // device the launches that never wait are used instead -- separate sweep, count and compaction: waiting workgroups hold compute
  // units the other scenes' kernels could use (3 scenes: 21.3 k frames/s with the one-launch list, 22.2 k without), and when the
  // process drives more hardware queues than the device runs at once (GPU_MAX_HW_QUEUES=8, 6 scenes) the queues are time-sliced and
  // the waits collapse the frame rate (492 frames/s against 19.2 k).  No frame of a multi-scene process contains a wait between
  // workgroups.
  const bool onePass = !g_debug_two_pass_visible_list && (one_pass_list_is_safe(s) || g_debug_force_list_stuck != 0);      // (the test hook needs the launch that can get stuck)
  const bool fusedSweep = !onlyVisible && onePass && !g_debug_separate_sweep && (s->cfg.bucketNum % kSweepChunk) == 0;
  if (!onlyVisible) {
    if (!fusedSweep) {
      KernelTimer ts(s, ITM_TK_ALLOC_SWEEP, st);
      allocate_sweep_kernel<<<nChunks, 256, 0, st>>>(s->allocKey, reqCur, reqNext, nChunks, s->hash, s->excessList, s->allocList,
                                                     rs->visibleType, s->counters, s->headBits, s->dirPtr, s->dirSlot, s->sdfMirror, v->depth, lazy ? 1 : 0, p);
    }
    s->frameParity++;
  }
  KernelTimer tv(s, ITM_TK_VISIBLE_LIST, st);
  if (onePass) {
    const uint32_t epoch = ++s->listEpoch;
    const SweepArgs sw{s->allocKey, reqNext, s->excessList, s->allocList, s->headBits, s->dirPtr, s->dirSlot, s->sdfMirror, v->depth, lazy ? 1 : 0, s->chunkSweepDone, s->chunkKeptGran,
                       s->fatalDev, fusedSweep ? g_debug_force_list_stuck - 1 : -1};
#define ITM_VL(CM, LZ, SW) visible_list_kernel<CM, LZ, SW><<<nChunks, 256, 0, st>>>(rs->visibleType, s->hash, s->chunkGran, epoch, reqCur, nChunks, s->counters, rs->visibleIds, rs->capIds, rs->counters, p, sw)
    if (onlyVisible) { if (lazy) ITM_VL(false, true, false); else ITM_VL(false, false, false); }
    else if (fusedSweep) { if (lazy) ITM_VL(true, true, true); else ITM_VL(true, false, true); }
    else { if (lazy) ITM_VL(true, true, false); else ITM_VL(true, false, false); }
#undef ITM_VL
  } else {
#define ITM_CNT(CM, LZ) visible_count_kernel<CM, LZ><<<nChunks, 256, 0, st>>>(rs->visibleType, s->hash, s->chunkVis, reqCur, nChunks, s->counters, p)
    if (onlyVisible) { if (lazy) ITM_CNT(false, true); else ITM_CNT(false, false); }
    else { if (lazy) ITM_CNT(true, true); else ITM_CNT(true, false); }
#undef ITM_CNT
    visible_compact_kernel<<<nChunks, 256, 0, st>>>(rs->visibleType, s->chunkVis, nChunks, s->noTotalEntries, rs->visibleIds, rs->capIds, rs->counters);
  }
  ITM_LAUNCH_CHECK();
  rs->listCoherent = true;   // list == non-zero visible types again
  if (s->cfg.useSwapping) return launch_swap_after_allocation(s, rs, st);      // swap states + re-allocation of swapped-out entries (_CPU.cpp:250-253,271-285)
  return ITM_OK;
}

int launch_allocate(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, bool fuseRangeInit, hipStream_t st) {
  int rc = launch_request_stage(s, v, rs, onlyVisible, fuseRangeInit, st);
  if (rc) return rc;
  return launch_sweep_stage(s, v, rs, onlyVisible, st);
}

// ordered compaction of the slots flagged in `flags` (one byte per slot, chunk counts from the flagging pass) into ascending ids
int launch_ordered_compaction(const uint8_t* flags, const int32_t* chunkCount, int nChunks, int nEntries, int32_t* ids, int cap, RenderCounters* rc, hipStream_t st) {
  visible_compact_kernel<<<nChunks, 256, 0, st>>>(flags, chunkCount, nChunks, nEntries, ids, cap, rc);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int launch_find_visible(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, hipStream_t st) {
  AllocParams p;
  int rc = fill_params(s, M, intr, rs->w, rs->h, rs->capIds, p);
  if (rc) return rc;
  const int nChunks = s->numChunks;
  // The flags and per-chunk counts of a free-view query live in the RENDER STATE (allocated on first use): the scene is const
  // here and a free-view render on another stream must not touch the allocation scratch a frame may be using.
  if (!rs->viewFlags) {
    ITM_HIP(hipMalloc((void**)&rs->viewFlags, (size_t)nChunks * kSweepChunk));
    ITM_HIP(hipMalloc((void**)&rs->viewChunkVis, (size_t)nChunks * 4));
  }
  freeview_flag_kernel<<<nChunks, 256, 0, st>>>(s->hash, rs->viewFlags, rs->viewChunkVis, p);
  visible_compact_kernel<<<nChunks, 256, 0, st>>>(rs->viewFlags, rs->viewChunkVis, nChunks, s->noTotalEntries, rs->visibleIds, rs->capIds, rs->counters);
  ITM_LAUNCH_CHECK();
  rs->listCoherent = false;   // the list no longer mirrors entriesVisibleType
  return ITM_OK;
}

// Abandons the block requests issued ahead on `rs` (itm_process_frame_ahead): the request keys and counters return to "no request",
// every visible type the requests marked is cleared and the entries of the visible list read 3 -- the state of the reference's
// AllocateSceneFromDepth right after its "previous list -> 3" loop (_CPU.cpp:160-161), from which any allocation may follow.
__global__ void __launch_bounds__(256) cancel_requests_kernel(uint8_t* __restrict__ visT, uint32_t* __restrict__ allocKey, int n) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;                                           // n is a multiple of 8
  uint32_t t = *(const uint32_t*)(visT + i);
  const uint32_t marked = t & 0x80808080u;
  if (marked) { t &= ~((marked >> 7) * 0xffu); *(uint32_t*)(visT + i) = t; }
  *(uint4*)(allocKey + i) = make_uint4(0u, 0u, 0u, 0u);
}
int cancel_ahead(itm_scene* s, itm_render_state* rs, hipStream_t st) {
  if (!rs->ahead.valid) return ITM_OK;
  const int n = s->noTotalEntries;
  cancel_requests_kernel<<<(n / 4 + 255) / 256, 256, 0, st>>>(rs->visibleType, s->allocKey, n);
  mark_previous_kernel<<<64, 256, 0, st>>>(rs->visibleIds, rs->counters, rs->visibleType, s->noTotalEntries, rs->capIds);
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipMemsetAsync((int2*)s->chunkReq + (size_t)(s->frameParity & 1u) * s->numChunks, 0, (size_t)s->numChunks * sizeof(int2), st));
  rs->ahead.valid = false;
  if (s->aheadRs == rs) s->aheadRs = nullptr;
  rs->listCoherent = true;       // list == the non-zero types
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_find_visible_blocks(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream stream) {
  if (!s || !M || !intr || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (s->cfg.indexType == ITM_INDEX_DENSE) return ITM_OK;  // ITMVisualisationEngine_CPU.cpp:34-37
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  if (refuse_while_ahead(s, rs, "FindVisibleBlocks")) return ITM_ERR_INVALID;
  return launch_find_visible(s, M, intr, rs, as_stream(stream));
}

}  // extern "C"
