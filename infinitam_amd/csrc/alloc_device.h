// alloc_device.h -- the per-pixel block request of AllocateSceneFromDepth as a device function, shared by the request kernel
// (alloc.hip) and by the launch that carries the NEXT frame's requests beside this frame's ICP maps (visualise.hip).
#pragma once

#include "itm_internal.h"

namespace itm {

struct AllocParams {
  Mat4 invM;     // inverse of M_d (host, ORUtils cofactor scheme)
  Mat4 M;        // M_d
  float ifx, ify, cx, cy;   // (1/fx, 1/fy, cx, cy)
  float fx, fy;
  float mu, oneOverBlock, vfmin, vfmax, voxelSize;
  int W, H;
  uint32_t mask;
  int bucketNum;
  int noTotalEntries;
  int stepBits;
  int capIds;
  int mirrorFloat;   // the sdf mirror holds floats (ITMVoxel_f / _f_rgb) rather than shorts
  AccelOrigin org;   // where the block directory / slot directory / sdf mirror cubes lie (itm_types.h)
  int useSwapping;   // scenes with a global cache: enlarged frustum for the re-test of the previous list (checkBlockVisibility<true>)
};

struct BlockRay {
  float px, py, pz;  // current point in block units
  float dx, dy, dz;
  int noSteps;
};

// Ray segment [d-mu, d+mu] of one depth pixel in block coordinates; same operation order as
// DeviceAgnostic/ITMSceneReconstructionEngine.h:155-184.  Returns false for rejected pixels.
__device__ inline bool make_block_ray(float d, int x, int y, const AllocParams& p, BlockRay& r) {
  if (d <= 0 || (d - p.mu) < 0 || (d - p.mu) < p.vfmin || (d + p.mu) > p.vfmax) return false;
  float cz = d;
  float cxp = cz * (((float)x - p.cx) * p.ifx);
  float cyp = cz * (((float)y - p.cy) * p.ify);
  float norm = sqrtf(cxp * cxp + cyp * cyp + cz * cz);
  float sa = 1.0f - p.mu / norm;
  Vec3 a = transform_point(p.invM, cxp * sa, cyp * sa, cz * sa);
  float sx = a.x * p.oneOverBlock, sy = a.y * p.oneOverBlock, sz = a.z * p.oneOverBlock;
  float sb = 1.0f + p.mu / norm;
  Vec3 b = transform_point(p.invM, cxp * sb, cyp * sb, cz * sb);
  float ex = b.x * p.oneOverBlock, ey = b.y * p.oneOverBlock, ez = b.z * p.oneOverBlock;
  float dx = ex - sx, dy = ey - sy, dz = ez - sz;
  norm = sqrtf(dx * dx + dy * dy + dz * dz);
  int noSteps = (int)ceilf(2.0f * norm);
  float div = (float)(noSteps - 1);
  r.px = sx; r.py = sy; r.pz = sz;
  r.dx = dx / div; r.dy = dy / div; r.dz = dz / div;
  r.noSteps = noSteps;
  return true;
}

// What the request stage of a frame reads and writes (buildHashAllocAndVisibleTypePP, DeviceAgnostic/ITMSceneReconstructionEngine.h:141-241)
struct RequestArgs {
  const float* depth; const uint4* hash; uint8_t* visT; uint32_t* allocKey; int2* chunkReq; SceneCounters* counters;
  float2* range; RenderCounters* rcnt; const int32_t* dirSlot;
  int32_t* fatalDev;     // the scene's status word in page-locked host memory (itm_internal.h), or nullptr
};

// A condition after which the scene is not what the reference would hold: recorded in the device-side counters (itm_get_counters) and
// in the host-visible word every entry point checks (a system-scope atomic: it crosses PCIe; never on the path of a healthy frame).
__device__ inline void raise_fatal(SceneCounters* __restrict__ counters, int32_t* __restrict__ fatalDev, int bits) {
  atomicOr(&counters->statusFlags, bits);
  if (fatalDev) __hip_atomic_fetch_or(fatalDev, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The work of one 16x16-pixel tile (tx, ty) by a workgroup of 256 lanes, one wave = 16x4 pixels.
template <bool ONLY_VISIBLE, bool FUSE_RANGE_INIT, bool LAZY>
__device__ inline void request_tile(int tx, int ty, const RequestArgs& a, const AllocParams& p) {
  const float* __restrict__ depth = a.depth; const uint4* __restrict__ hash = a.hash; uint8_t* __restrict__ visT = a.visT;
  uint32_t* __restrict__ allocKey = a.allocKey; int2* __restrict__ chunkReq = a.chunkReq; SceneCounters* __restrict__ counters = a.counters;
  float2* __restrict__ range = a.range; RenderCounters* __restrict__ rcnt = a.rcnt; const int32_t* __restrict__ dirSlot = a.dirSlot;
  // LAZY: instead of first marking last frame's list as type 3 (a separate launch), this frame's
  // touches carry bit 7; visible_count_kernel then reads every other non-zero type as "3".
  constexpr uint8_t kTouched = LAZY ? 0x80 : 0x00;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (FUSE_RANGE_INIT && tx == 0 && ty == 0) {
    if (threadIdx.x == 0) { rcnt->noRenderingBlocks = 0; rcnt->renderingBlocksAccepted = -1; }
  }
  if (tx == 0 && ty == 0 && threadIdx.x == 0) rcnt->listInvalid = 0;
  const int x = tx * 16 + (lane & 15);
  const int y = ty * 16 + wave * 4 + (lane >> 4);
  if (x >= p.W || y >= p.H) return;
  const int loc = x + y * p.W;
  if (FUSE_RANGE_INIT) range[loc] = make_float2(999999.9f, 0.05f);  // CreateExpectedDepths init, fused
  BlockRay r;
  if (!make_block_ray(depth[loc], x, y, p, r)) return;
  if (!ONLY_VISIBLE && r.noSteps > (1 << p.stepBits)) {
    // more steps than the request key can number (a band of thousands of blocks: mu / voxelSize beyond 2 000): the reference would walk
    // them all; the frame is not the reference's any more and the scene says so (statusFlags bit 0 -> ITM_ERR_DEVICE at the next call)
    raise_fatal(counters, a.fatalDev, 1);
    r.noSteps = 1 << p.stepBits;
  }
  // The table slot of a block that exists inside the slot directory's cube comes from ONE load (entries are never swapped out: ptr >= 0,
  // hence type 1); every other case takes the probe below.  A ray's steps are independent of each other -- where step i + 2 lies is known
  // before step i has been looked up -- so the directory is asked two steps ahead: the 2-4 look-ups of a pixel are in flight together
  // instead of one memory round trip after the other (7.4 -> 7.2 us on BASELINE configs[1]: the launch is mostly its ramp and the depth load).
  auto slot_at = [&](float qx, float qy, float qz) -> int {
    if (!dirSlot) return -1;
    const int cx_ = (int)(int16_t)(int)floorf(qx), cy_ = (int)(int16_t)(int)floorf(qy), cz_ = (int)(int16_t)(int)floorf(qz);
    const uint32_t ux = (uint32_t)(cx_ - p.org.dx), uy = (uint32_t)(cy_ - p.org.dy), uz = (uint32_t)(cz_ - p.org.dz);
    return dir_covers(ux, uy, uz) ? dirSlot[dir_cell(ux, uy, uz)] : -1;
  };
  // (positions advance by repeated addition, as the reference's loop does: the look-ahead replays the same sums)
  const float p1x = r.px + r.dx, p1y = r.py + r.dy, p1z = r.pz + r.dz;
  int slot0 = slot_at(r.px, r.py, r.pz);
  int slot1 = r.noSteps > 1 ? slot_at(p1x, p1y, p1z) : -1;
  for (int i = 0; i < r.noSteps; ++i) {
    const float p2x = (r.px + r.dx) + r.dx, p2y = (r.py + r.dy) + r.dy, p2z = (r.pz + r.dz) + r.dz;
    const int slot2 = i + 2 < r.noSteps ? slot_at(p2x, p2y, p2z) : -1;
    const int slot = slot0;
    slot0 = slot1; slot1 = slot2;
    if (slot >= 0) {
      visT[slot] = 1 | kTouched;
      r.px += r.dx; r.py += r.dy; r.pz += r.dz;
      continue;
    }
    const int bx = (int)(int16_t)(int)floorf(r.px), by = (int)(int16_t)(int)floorf(r.py), bz = (int)(int16_t)(int)floorf(r.pz);
    int idx = hash_index(bx, by, bz, p.mask);
    HashEntry he = unpack_entry(hash[idx]);
    bool found = false;
    if (he.px == bx && he.py == by && he.pz == bz && he.ptr >= -1) {
      visT[idx] = ((he.ptr == -1) ? 2 : 1) | kTouched;
      found = true;
    }
    if (!found) {
      bool isExcess = false;
      if (he.ptr >= -1) {
        while (he.offset >= 1) {
          idx = p.bucketNum + he.offset - 1;
          he = unpack_entry(hash[idx]);
          if (he.px == bx && he.py == by && he.pz == bz && he.ptr >= -1) {
            visT[idx] = ((he.ptr == -1) ? 2 : 1) | kTouched;
            found = true;
            break;
          }
        }
        isExcess = true;
      }
      if (!found) {
        if (!isExcess) visT[idx] = 1 | kTouched;
        if (!ONLY_VISIBLE) {
          const uint32_t key = (((uint32_t)loc << p.stepBits) | (uint32_t)i) + 1u;
          const uint32_t old = atomicMax(&allocKey[idx], key);
          if (old == 0u) {
            atomicAdd(&chunkReq[idx / kSweepChunk].x, 1);
            if (isExcess) atomicAdd(&chunkReq[idx / kSweepChunk].y, 1);
          }
        }
      }
    }
    r.px += r.dx; r.py += r.dy; r.pz += r.dz;
  }
}


}  // namespace itm
