// integrate.hip -- per-voxel TSDF / weight / colour fusion.
//
// Reference behaviour:
//   IntegrateIntoScene (hash)   DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:47-114
//   IntegrateIntoScene (dense)  DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:319-369
//   computeUpdatedVoxelDepthInfo / ColorInfo / ComputeUpdatedVoxelInfo
//                               DeviceAgnostic/ITMSceneReconstructionEngine.h:9-139
//   interpolateBilinear         DeviceAgnostic/ITMPixelUtils.h:11-39
//
// MI355X design: the hash kernel gives (block, 4 z-slices) to ONE WAVE -- four contiguous 64-voxel runs in flight together, then
// four depth gathers together -- with 2 048 workgroups of 8 waves striding over the device-resident visible list, so no count is
// read back; voxels of the short types are mirrored into the position-addressed sdf copy the ray caster reads (itm_types.h).
// The dense kernel streams the volume with 16-byte accesses (4 ITMVoxel_s per lane), culls whole columns of groups against the
// frustum with two integer comparisons and writes only 128-bit groups that changed.  The depth map is small (1.2 MB) and stays in L2.
#include <cmath>
#include <cstring>
#include <memory>

#include "itm_internal.h"
#include "range_device.h"

namespace itm {

#ifndef ITM_EXP_FUSED_STAMPS
#define ITM_EXP_FUSED_STAMPS 0
#endif
#ifndef ITM_MIRROR_FLOAT_TYPES
#define ITM_MIRROR_FLOAT_TYPES 0     // (scene.hip decides whether a float scene gets a mirror; the same switch must be given to both files)
#endif
int g_debug_integrate_wgs = 0;
int g_debug_no_fused_projection = 0;
int g_debug_dense_group_cull = 0;   // debug key 9: per-group frustum test instead of the per-column row interval
int g_debug_dense_no_strips = 0;    // debug key 17: the launch shape of rounds 1-2 (four groups per lane) instead of the strip kernel
int g_debug_dense_classify = 0;     // debug key 16: 0 = classify groups before the fetch, 1 = no classification, 2 = after the fetch, 3 = check mode

struct FuseParams {
  Mat4 M_d, M_rgb;
  float fx, fy, cx, cy;
  float fxc, fyc, cxc, cyc;
  float mu, voxelSize;
  float rcpMu;      // RN(1/mu), computed on the host
  int muFast;       // 1: eta/mu may use the 3-instruction reciprocal division (host-checked)
  int maxW;
  int W, H, Wc, Hc;
  int stopAtMax;
  AccelOrigin org;  // where the sdf mirror cube lies (hash scenes)
  int idsCap;       // capacity of the visible list (ids), for the speculative first fetch of integrate_hash_body
};

// Depth part, stage 1: project the voxel; returns the index of the depth pixel it falls on, or -1 when the voxel
// cannot be touched (behind the camera / outside the image).  Operation order as SURVEY.md Appendix A.5.
// EARLY_OUT: the conservative frustum test in front of the divisions.  It pays where most voxels leave through it (a dense volume: ~70 %
// of the voxels are outside the image) and costs four products and four comparisons per voxel where hardly any does (the blocks of a
// visible list: the hash kernels run without it -- round 6, ~10 % of their vector instructions).
template <bool EARLY_OUT = true>
__device__ inline int fuse_depth_project(float mx, float my, float mz, const FuseParams& p, float& pcz) {
  Vec3 pc = transform_point(p.M_d, mx, my, mz);
  pcz = pc.z;
  if (pc.z <= 0) return -1;
  const float tx = p.fx * pc.x, ty = p.fy * pc.y;
#if ITM_FAST_DIVISIONS
  // Conservative early-out before the two divisions: the voxel is rejected below when
  // u = tx/z + cx is outside [1, W-2]; half a pixel of margin dwarfs every rounding error, so this
  // never rejects a voxel the exact test would keep (most voxels of a dense volume leave here).
  if constexpr (EARLY_OUT) {
    if (tx < (0.5f - p.cx) * pc.z || tx > ((float)(p.W - 2) + 0.5f - p.cx) * pc.z ||
        ty < (0.5f - p.cy) * pc.z || ty > ((float)(p.H - 2) + 0.5f - p.cy) * pc.z) return -1;
  }
  float u, v;
  if (pc.z >= 1e-4f && pc.z <= 1e4f) {       // normal range: shared refined reciprocal, exact quotients
    const float rz = refined_rcp(pc.z);
    u = div_by_rcp(tx, pc.z, rz) + p.cx;
    v = div_by_rcp(ty, pc.z, rz) + p.cy;
  } else {
    u = tx / pc.z + p.cx;
    v = ty / pc.z + p.cy;
  }
#else
  const float u = tx / pc.z + p.cx;
  const float v = ty / pc.z + p.cy;
#endif
  if ((u < 1) || (u > p.W - 2) || (v < 1) || (v > p.H - 2)) return -1;
  // (both factors are below 2^23 -- images have fewer than 2^24 pixels, scene.hip -- so the 24-bit multiply-add is exact and one
  // full-rate instruction where the 32-bit form is a quarter-rate 64-bit mad)
  return __mul24((int)(v + 0.5f), p.W) + (int)(u + 0.5f);
}

// The running average of computeUpdatedVoxelDepthInfo (DeviceAgnostic/ITMSceneReconstructionEngine.h:45-55) with the clamped
// observation newF = MIN(1, eta / mu).
template <class VX>
__device__ inline void fuse_average(typename VX::Reg& r, float newF, const FuseParams& p) {
  const float oldF = VX::to_float(VX::raw_sdf(r));
  const int oldW = VX::w_depth(r);
  int newW = 1;
  newF = (float)oldW * oldF + (float)newW * newF;
  newW = oldW + newW;
#if ITM_FAST_DIVISIONS
  {
    // newW is an integer in [1, 256]: the refined reciprocal equals RN(1/newW) for all of them (tested),
    // its significand is never all ones, so the 3-instruction quotient is the correctly rounded one
    const float w = (float)newW;
    newF = div_markstein(newF, w, refined_rcp(w));
  }
#else
  newF /= (float)newW;
#endif
  newW = (newW < p.maxW) ? newW : p.maxW;
  r = VX::with_depth(r, newF, newW);
}

// Stage 2: the running average with the measured depth dm of that pixel.  Returns eta (or -1 when the voxel is not
// touched); `touched` tells whether the register image changed.
template <class VX>
__device__ inline float fuse_depth_update(typename VX::Reg& r, float dm, float pcz, const FuseParams& p, bool& touched) {
  if (dm <= 0.0f) return -1;
  const float eta = dm - pcz;
  if (eta < -p.mu) return eta;
#if ITM_FAST_DIVISIONS
  float newF = p.muFast ? div_markstein(eta, p.mu, p.rcpMu) : eta / p.mu;
#else
  float newF = eta / p.mu;
#endif
  newF = (1.0f < newF) ? 1.0f : newF;
  fuse_average<VX>(r, newF, p);
  touched = true;
  return eta;
}

template <class VX>
__device__ inline float fuse_depth(typename VX::Reg& r, float mx, float my, float mz, const float* __restrict__ depth,
                                   const FuseParams& p, bool& touched) {
  float pcz;
  const int pix = fuse_depth_project(mx, my, mz, p, pcz);
  if (pix < 0) return -1;
  return fuse_depth_update<VX>(r, depth[pix], pcz, p, touched);
}

// The colour update's condition, !(eta > mu || fabs(eta / mu) > 0.25) (DeviceAgnostic/ITMSceneReconstructionEngine.h:127-131), without the
// IEEE division macro on the common path: for eta < -mu the correctly rounded quotient is <= -1 (division is monotone), so the voxel
// is outside the band whatever eta is (-inf included); for eta in [-mu, mu] the Markstein quotient is the correctly rounded one
// whenever it is a normal number, and both are far below 0.25 when it is not; a NaN eta passes every comparison as it does in the
// reference.  (Round 6: ~10 vector instructions less per voxel of the colour kernels.)
__device__ inline bool in_colour_band(float eta, const FuseParams& p) {
  if (eta < -p.mu || eta > p.mu) return false;
#if ITM_FAST_DIVISIONS
  const float q = p.muFast ? div_markstein(eta, p.mu, p.rcpMu) : eta / p.mu;
#else
  const float q = eta / p.mu;
#endif
  return !(fabsf(q) > 0.25f);
}

__device__ inline float round_half_away(float x) { return (x < 0) ? (x - 0.5f) : (x + 0.5f); }

template <class VX>
__device__ inline void fuse_colour(typename VX::Reg& r, float mx, float my, float mz, const uchar4* __restrict__ rgb, const FuseParams& p) {
  int oc[3], owc;
  VX::get_color(r, oc, owc);
  const float oldW = (float)owc;
  Vec3 pc = transform_point(p.M_rgb, mx, my, mz);
  float u, v;
#if ITM_FAST_DIVISIONS
  if (pc.z >= 1e-4f && pc.z <= 1e4f) {       // normal range: shared refined reciprocal, exact quotients (as in fuse_depth)
    const float rz = refined_rcp(pc.z);
    u = div_by_rcp(p.fxc * pc.x, pc.z, rz) + p.cxc;
    v = div_by_rcp(p.fyc * pc.y, pc.z, rz) + p.cyc;
  } else
#endif
  {
    u = p.fxc * pc.x / pc.z + p.cxc;
    v = p.fyc * pc.y / pc.z + p.cyc;
  }
  if ((u < 1) || (u > p.Wc - 2) || (v < 1) || (v > p.Hc - 2)) return;
  const int px = (int)floorf(u), py = (int)floorf(v);
  const float dx = u - (float)px, dy = v - (float)py;
  // interpolateBilinear (DeviceAgnostic/ITMPixelUtils.h:11-39) fetches b, c, d only when their weight is not zero and uses 0 otherwise.
  // All four taps are inside the image here (1 <= u <= Wc - 2, 1 <= v <= Hc - 2), and a finite channel value times a zero weight is
  // the same +0 the reference adds: fetching the four unconditionally is bit-identical and ONE round trip instead of three
  // dependent ones (until round 6 each conditional tap sat in its own branch behind a wait of its own).
  const uchar4* __restrict__ row = rgb + (size_t)py * p.Wc + px;
  const uchar4 A = row[0], B = row[1], C = row[p.Wc], D = row[p.Wc + 1];
  const float a4[3] = {(float)A.x, (float)A.y, (float)A.z}, b4[3] = {(float)B.x, (float)B.y, (float)B.z};
  const float c4[3] = {(float)C.x, (float)C.y, (float)C.z}, d4[3] = {(float)D.x, (float)D.y, (float)D.z};
  float newW = oldW + 1.0f;
  int nc[3];
#if ITM_FAST_DIVISIONS
  // x / 255 and c / newW (newW an integer in [1, 256]) as Markstein quotients: RN(1/255) is a compile-time constant,
  // the refined reciprocal equals RN(1/newW) for every such weight, neither significand is all ones
  // (tests/test_hip_parity.py::test_fast_divisions_are_ieee, cases 5 and 6)
  const float r255 = 1.0f / 255.0f;
  const float rW = refined_rcp(newW);
#endif
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float m = (a4[k] * (1.0f - dx) * (1.0f - dy) + b4[k] * dx * (1.0f - dy) + c4[k] * (1.0f - dx) * dy + d4[k] * dx * dy);
#if ITM_FAST_DIVISIONS
    m = div_markstein(m, 255.0f, r255);
    float c = div_markstein((float)oc[k], 255.0f, r255) * oldW + m * 1.0f;
    c = div_markstein(c, newW, rW);
#else
    m = m / 255.0f;
    float c = ((float)oc[k] / 255.0f) * oldW + m * 1.0f;
    c /= newW;
#endif
    int vi = (int)round_half_away(c * 255.0f);
    vi = (vi < 255) ? vi : 255;
    nc[k] = (0 < vi) ? vi : 0;
  }
  const float maxWf = (float)(p.maxW & 0xff);
  newW = (newW < maxWf) ? newW : maxWf;
  r = VX::with_color(r, nc, (int)newW);
}

template <class VX>
__device__ inline bool fuse_voxel(typename VX::Reg& r, float mx, float my, float mz, const float* __restrict__ depth,
                                  const uchar4* __restrict__ rgb, const FuseParams& p) {
  bool touched = false;
  const float eta = fuse_depth<VX>(r, mx, my, mz, depth, p, touched);
  if constexpr (VX::kColor) {
    if (in_colour_band(eta, p)) {
      fuse_colour<VX>(r, mx, my, mz, rgb, p);
      touched = true;
    }
  }
  return touched;
}

// Work item = kSlices consecutive z-slices of one visible block, done by ONE WAVE: lane <-> (x, y), one 64-voxel run per slice.
// The persistent waves stride over the items (item i = block i / (8 / kSlices), slice group i % (8 / kSlices)), so the waves of a
// workgroup sit on the same block or its list neighbours, and no wave ever waits for another.
//
// Why a wave and not a workgroup per block (rounds 1 and 2 until here: 512 lanes on one block, one voxel per lane): the counters
// (profiles/r2_integrate_counters.md) show the hash kernels at 43-46 % VALU utilisation with waves waiting 62-80 % of their time:
// every block costs a chain of dependent round trips -- list position -> hash entry -> voxels -> depth pixel (-> colour pixels) ->
// store -- and a wave had just 64 voxels (256-768 B) in flight along it.  32 waves x 768 B per CU over ~2 us of loaded latency is
// 3.1 TB/s for the whole chip, which is exactly what BASELINE configs[4] measured.  Here a wave has kSlices runs in flight at once
// (their loads are issued together, then the projections, then the depth gathers together), the entry of its NEXT item is
// requested before it starts on the current one, and the x / y partial sums of the projection are shared by the slices.
// (Two x-neighbours per lane with packed fp32 arithmetic -- v_pk_mul / add / fma_f32, bit-exact -- was built and measured first:
// -8 % VALU instructions, but +5 % time; the kernels are not VALU bound.)
// Slices per item: 4 (1 / 2 / 8: 243 / 211 / 195 us against 186 on BASELINE configs[4] when it was chosen).  ITMVoxel_s (4 bytes, 36 registers)
// takes the whole block per item -- eight runs in flight per wave; end of round 4, BASELINE configs[1], integration + projection launch:
// 2 slices 22.5-23.0 us, 4: 20.4-20.5, 8: 19.2-19.5; the 12-byte ITMVoxel_f_rgb loses with 8 (configs[4]: 178 -> 196 us, registers).
// (Also measured and not kept, profiles/r4_integrate_notes.md: a whole block per wave with 16 bytes per lane, for all four voxel types --
// configs[1] 19.9 -> 20.8-22.1 us, configs[4] 180 -> 264; the next item's voxel runs requested behind the depth gathers -- configs[4]
// 189 -> 198 us, the second register set costs a wave of occupancy; non-temporal stores of the voxels or of the mirror values: +-0 / +2 us
// on the ray cast that follows.)
constexpr int kIntegrateSlices = 4, kIntegrateSlicesShort = 8;
template <class VX> __host__ __device__ constexpr int slices_of() { return VX::kBytes == 4 ? kIntegrateSlicesShort : kIntegrateSlices; }

template <class VX>
__device__ inline void load_item(const HashEntry& he, int z0, int lane, const void* __restrict__ vba, typename VX::Reg r[slices_of<VX>()]) {
  constexpr int kSlices = slices_of<VX>();
  const size_t vi = (size_t)(he.ptr < 0 ? 0 : he.ptr) * kBlockVoxels + (size_t)z0 * 64 + lane;      // block 0 is always there
#pragma unroll
  for (int k = 0; k < kSlices; ++k) r[k] = VX::load(vba, vi + 64 * k);
}

// r: the item's voxels (already requested)
template <class VX>
__device__ inline void integrate_item(const HashEntry& he, int z0, int lane, typename VX::Reg r[slices_of<VX>()], void* __restrict__ vba, void* __restrict__ sdfMirror,
                                      const float* __restrict__ depth, const uchar4* __restrict__ rgb, const FuseParams& p) {
  constexpr int kSlices = slices_of<VX>();
  const bool present = he.ptr >= 0;
  const int x = lane & 7, y = lane >> 3;
  const size_t vi = (size_t)(present ? he.ptr : 0) * kBlockVoxels + (size_t)z0 * 64 + lane;
  const float mx = (float)(he.px * kBlockSide + x) * p.voxelSize;
  const float my = (float)(he.py * kBlockSide + y) * p.voxelSize;
  using MC = MirrorCodec<VX::kShort>;
  size_t mbase = 0;
  typename MC::T* mirror = nullptr;
  // (the block's page was mapped when the block was allocated; the table entry is requested here, beside the voxels, and used at the end.
  // The float voxel types carry no mirror unless built with ITM_MIRROR_FLOAT_TYPES: their kernels do not carry its code either -- two
  // registers more and ITMVoxel_f_rgb drops from six waves per SIMD to five, 180 -> 203 us on BASELINE configs[4])
  if constexpr (VX::kShort || ITM_MIRROR_FLOAT_TYPES) {
    if (sdfMirror && mirror_block_base<false>(p.org, he.px, he.py, he.pz, mbase)) mirror = (typename MC::T*)sdfMirror;
  }
  // stage 1: project every slice's voxel; stage 2: all depth pixels together; stage 3: update (+ colour), store what changed.
  // Stages 1 and 2 do not look at the voxels: the item's runs (requested by the caller) are still on their way while the wave projects
  // and gathers, and the first wait for them stands in front of stage 3.  (Until round 6 the stopIntegratingAtMaxW test -- which reads
  // the voxel's weight -- stood in front of the projection: the compiler's s_waitcnt for run k preceded the projection of slice k, so a
  // wave idled through the voxels' round trip and then through the depth pixels' instead of through the longer of the two.)
  int pix[kSlices];
  float pcz[kSlices], mz[kSlices];
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    mz[k] = (float)(he.pz * kBlockSide + z0 + k) * p.voxelSize;
    pix[k] = present ? fuse_depth_project<false>(mx, my, mz[k], p, pcz[k]) : -2;     // -2: no block
  }
  float dm[kSlices];
#pragma unroll
  for (int k = 0; k < kSlices; ++k) dm[k] = depth[pix[k] >= 0 ? pix[k] : 0];
  // every gather is issued HERE, before the first look at a voxel (left alone the compiler sinks the gather of a slice into that slice's
  // conditional update, behind the wait for the slice's voxels)
#pragma unroll
  for (int k = 0; k < kSlices; ++k) asm volatile("" : "+v"(dm[k]));
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    if (pix[k] == -2 || (p.stopAtMax && VX::w_depth(r[k]) == p.maxW)) continue;      // (a saturated voxel is skipped altogether, colour included)
    bool touched = false;
    const float eta = (pix[k] >= 0) ? fuse_depth_update<VX>(r[k], dm[k], pcz[k], p, touched) : -1.0f;
    if constexpr (VX::kColor) {
      if (in_colour_band(eta, p)) {
        fuse_colour<VX>(r[k], mx, my, mz[k], rgb, p);
        touched = true;
      }
    }
    if (touched) {
      VX::store(vba, vi + 64 * k, r[k]);
      if (mirror) mirror[mbase + mirror_block_voxel((uint32_t)x, (uint32_t)y, (uint32_t)(z0 + k))] = MC::of(VX::raw_sdf(r[k]));     // sdf mirror (itm_types.h)
    }
  }
}

template <class VX>
__device__ inline void integrate_hash_body(int wgIdx, int wgCount, const int32_t* __restrict__ visibleIds, RenderCounters* __restrict__ rc,
                                           const uint4* __restrict__ hash, void* __restrict__ vba, void* __restrict__ sdfMirror,
                                           const float* __restrict__ depth, const uchar4* __restrict__ rgb, const FuseParams& p) {
  constexpr int kSlices = slices_of<VX>();
  constexpr int kItemsPerBlock = kBlockSide / kSlices;
  const int lane = threadIdx.x & 63;
  const int waves = wgCount * (int)(blockDim.x >> 6);
  int i = __builtin_amdgcn_readfirstlane(wgIdx * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
  // What a wave needs before it can ask for its first voxels is a chain of scalar round trips to memory the previous launch has just
  // written: the list's state, the id at the wave's position, the table entry behind the id.  The first two do not depend on each other --
  // the id is fetched speculatively (the position is clamped to the list's capacity; the value is only used once the count says it is
  // one) -- so they travel together: three round trips to the first voxel request instead of five (until round 6: listInvalid, then the
  // count, then the id, then the entry).
  const int idsLast = p.idsCap - 1;
  const int at0 = i / kItemsPerBlock;
  int idCur = visibleIds[at0 < idsLast ? at0 : idsLast];
  int invalid = rc->listInvalid, nv = rc->noVisibleEntries;
  asm volatile("" : "+s"(idCur), "+s"(invalid), "+s"(nv));      // all three requested above this line, looked at below it
  if (invalid) return;                                // the list of this frame is not the reference's: fuse nothing (alloc.hip, statusFlags bit 1)
  const int nItems = nv * kItemsPerBlock;
  if (i >= nItems) return;
  // the id of the NEXT item travels with the first entry; inside the loop the entry one item ahead and the id two items ahead are requested
  // at the top of an item and looked at behind it (positions clamped to the list's end: a wave without a next item re-reads its last)
  int nxt = i + waves;
  int idNext = visibleIds[(nxt < nItems ? nxt : nItems - 1) / kItemsPerBlock];
  HashEntry cur = unpack_entry(hash[idCur]);
  typename VX::Reg r[kSlices];
  load_item<VX>(cur, (i % kItemsPerBlock) * kSlices, lane, vba, r);
  // (the next item's VOXELS in a second register set, so that the launch could have no more waves than the device holds -- every item in
  // flight from the first microsecond -- was built and measured in round 6 on BASELINE configs[1]: 768 workgroups 21.9-22.3 us, 1 024:
  // 20.5, against 19.0-19.4 for 2 048 workgroups without it; and so was a wave that owns TWO blocks and works on them together -- sixteen
  // runs, sixteen projections, sixteen gathers, one wait, 80 registers, every block of the frame in flight from the first microsecond on
  // 768 resident workgroups: 20.5-21.0 us, the workgroups then last 9-18 us each.  A wave with two blocks takes twice as long as a wave
  // with one, whatever is prefetched or interleaved: the launch is bound by what the blocks cost to work through, not by the chain
  // (profiles/r6_notes.md).)
  for (;;) {
    const bool more = nxt < nItems;
    uint4 rawAhead = hash[idNext];
    const int after = nxt + waves;
    int idAfter = visibleIds[(after < nItems ? after : nItems - 1) / kItemsPerBlock];
    integrate_item<VX>(cur, (i % kItemsPerBlock) * kSlices, lane, r, vba, sdfMirror, depth, rgb, p);
    asm volatile("" : "+s"(rawAhead.x), "+s"(rawAhead.y), "+s"(rawAhead.z), "+s"(rawAhead.w), "+s"(idAfter));
    if (!more) break;
    cur = unpack_entry(rawAhead);
    load_item<VX>(cur, (nxt % kItemsPerBlock) * kSlices, lane, vba, r);
    i = nxt; nxt = after; idNext = idAfter;
  }
}

#ifndef ITM_INTEGRATE_BLOCK
#define ITM_INTEGRATE_BLOCK 256          // lanes per workgroup of the stand-alone launch: at 72 registers a SIMD holds 7 waves, which 8-wave workgroups cannot fill (6); measured 134.5 -> 120.9 us on BASELINE configs[4]
#endif
#ifdef ITM_INTEGRATE_WAVES_PER_EU
#define ITM_INTEGRATE_OCC __attribute__((amdgpu_waves_per_eu(ITM_INTEGRATE_WAVES_PER_EU, ITM_INTEGRATE_WAVES_PER_EU)))
#else
#define ITM_INTEGRATE_OCC
#endif
template <class VX>
__global__ void __launch_bounds__(512) ITM_INTEGRATE_OCC integrate_hash_kernel(const int32_t* __restrict__ visibleIds, RenderCounters* __restrict__ rc,
                                                             const uint4* __restrict__ hash, void* __restrict__ vba, void* __restrict__ sdfMirror,
                                                             const float* __restrict__ depth, const uchar4* __restrict__ rgb, FuseParams p) {
  integrate_hash_body<VX>(blockIdx.x, gridDim.x, visibleIds, rc, hash, vba, sdfMirror, depth, rgb, p);
}

// IntegrateIntoScene and the projection half of CreateExpectedDepths in ONE launch: both only depend on the
// visible list, integration is ALU bound on every CU while the kRangeParts projection workgroups are bound by
// LDS atomics on 32 CUs, so they overlap almost perfectly (17 + 11.5 us as two launches -> ~19 us).  The first
// kRangeParts workgroups project, the others are the persistent integration workgroups.
#if ITM_EXP_FUSED_STAMPS
// measurement build: per-workgroup start / end of the fused launch on the constant-rate global clock (100 MHz)
__device__ unsigned long long g_fusedStamps[8192 * 2];
#define ITM_FS(...) __VA_ARGS__
#else
#define ITM_FS(...)
#endif
template <class VX>
__global__ void __launch_bounds__(512) integrate_project_kernel(const int32_t* __restrict__ visibleIds, RenderCounters* __restrict__ rc,
                                                                const uint4* __restrict__ hash, void* __restrict__ vba, void* __restrict__ sdfMirror,
                                                                const float* __restrict__ depth, const uchar4* __restrict__ rgb, FuseParams p,
                                                                float2* __restrict__ range, uint4* __restrict__ projBuf, uint2* __restrict__ partials,
                                                                ProjParams pp, int RW, int RH) {
  extern __shared__ uint2 cells[];
  ITM_FS(if (threadIdx.x == 0 && blockIdx.x < 8192) g_fusedStamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();)
  if (blockIdx.x < kRangeParts) {
    // the few projection workgroups share their CUs with integration workgroups and would otherwise be the last to finish
    // (21.8 us fused against 15.7 us for the integration alone): let their waves win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    project_partial_body(blockIdx.x, cells, visibleIds, rc, hash, range, projBuf, partials, pp, RW, RH);
    ITM_FS(__syncthreads(); if (threadIdx.x == 0) g_fusedStamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();)
    return;
  }
  integrate_hash_body<VX>(blockIdx.x - kRangeParts, gridDim.x - kRangeParts, visibleIds, rc, hash, vba, sdfMirror, depth, rgb, p);
  ITM_FS(__syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 8192) g_fusedStamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();)
}
#if ITM_EXP_FUSED_STAMPS
extern "C" int itm_debug_read_fused_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fusedStamps), (size_t)n * 8); }
#endif

// Dense volume, generic voxel type: one voxel per lane, x fastest (coalesced).
template <class VX>
__global__ void __launch_bounds__(256) integrate_dense_kernel(void* __restrict__ vba, const float* __restrict__ depth,
                                                              const uchar4* __restrict__ rgb, FuseParams p, int sx, int sy, int sz,
                                                              int ox, int oy, int oz) {
  const size_t n = (size_t)sx * sy * sz;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t loc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; loc < n; loc += stride) {
    const int z = (int)(loc / ((size_t)sx * sy));
    const int rem = (int)(loc - (size_t)z * sx * sy);
    const int y = rem / sx;
    const int x = rem - y * sx;
    typename VX::Reg r = VX::load(vba, loc);
    if (p.stopAtMax && VX::w_depth(r) == p.maxW) continue;
    const float mx = (float)(x + ox) * p.voxelSize, my = (float)(y + oy) * p.voxelSize, mz = (float)(z + oz) * p.voxelSize;
    if (fuse_voxel<VX>(r, mx, my, mz, depth, rgb, p)) VX::store(vba, loc, r);
  }
}

// ---- depth tiles: min / max of the depth image over tiles of 8, 16, 32 and 64 pixels ------------------------------------------
// Most voxels a dense volume presents to a frame lie in observed free space (eta >= mu: the observation is clamped to exactly 1)
// or in the shadow of a surface (eta < -mu: the voxel is not touched).  For a 4-voxel group whose whole pixel footprint falls into
// tiles with  min depth - max pc.z >= mu  (resp.  max depth - min pc.z < -mu)  every voxel of the group takes that branch of
// computeUpdatedVoxelDepthInfo, so neither the projection with its two divisions nor the depth gathers are needed: the free-space
// update is the running average with newF = 1 (for a voxel still holding the initial 32767 simply w + 1), the shadow group is not
// even fetched.  The bounds are conservative (one and a half pixels and 1e-5 of the coordinate magnitudes of slack, two orders above
// the rounding of the exact path), so the result is the exact path's, bit for bit; tests/test_dense_cull.py runs the classified
// kernel against the exact one over whole volumes and checks every classified group against the exact per-voxel outcome
// (itm_debug_dense_classify_check).
constexpr int kTileLevels = 4;       // tile sides 8, 16, 32, 64
struct TileLevels {
  int off[kTileLevels], tw[kTileLevels], th[kTileLevels];
  int total;
};
static TileLevels tile_levels(int w, int h) {
  TileLevels t; int o = 0;
  for (int l = 0; l < kTileLevels; ++l) {
    const int side = 8 << l;
    t.off[l] = o; t.tw[l] = (w + side - 1) / side; t.th[l] = (h + side - 1) / side;
    o += t.tw[l] * t.th[l];
  }
  t.total = o;
  return t;
}
// One workgroup per 64 x 64 pixels: lane <-> a 4 x 4 patch, LDS reduction to the 8-, 16-, 32- and 64-pixel tiles.
// A NaN pixel makes its tiles (-inf, +inf): no group over them is ever classified.
__global__ void __launch_bounds__(256) depth_tiles_kernel(const float* __restrict__ depth, int W, int H, float2* __restrict__ tiles, TileLevels tl) {
  __shared__ float2 s8[64], s16[16], s32[4];
  const int t = threadIdx.x;
  const int px = blockIdx.x * 64 + (t & 15) * 4, py = blockIdx.y * 64 + (t >> 4) * 4;
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (py + j >= H) break;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (px + i >= W) break;
      const float d = depth[(size_t)(py + j) * W + px + i];
      if (d != d) { mn = -INFINITY; mx = INFINITY; }
      mn = fminf(mn, d); mx = fmaxf(mx, d);
    }
  }
  // 2 x 2 patches -> one 8-pixel tile: lanes (t & 15) ^ 1 and (t >> 4) ^ 1 (t ^ 1, t ^ 16)
  mn = fminf(mn, __shfl_xor(mn, 1)); mx = fmaxf(mx, __shfl_xor(mx, 1));
  mn = fminf(mn, __shfl_xor(mn, 16)); mx = fmaxf(mx, __shfl_xor(mx, 16));
  const int cx = (t & 15) >> 1, cy = t >> 5;          // 8 x 8 tiles of 8 pixels in this workgroup
  if (((t & 1) | ((t >> 4) & 1)) == 0) s8[cy * 8 + cx] = make_float2(mn, mx);
  __syncthreads();
  auto put = [&](int level, int lx, int ly, float2 v) {
    const int gx = blockIdx.x * (8 >> level) + lx, gy = blockIdx.y * (8 >> level) + ly;
    if (gx < tl.tw[level] && gy < tl.th[level]) tiles[tl.off[level] + gy * tl.tw[level] + gx] = v;
  };
  auto merge = [](float2 a, float2 b) { return make_float2(fminf(a.x, b.x), fmaxf(a.y, b.y)); };
  if (t < 64) put(0, t & 7, t >> 3, s8[t]);
  if (t < 16) {
    const int x = t & 3, y = t >> 2;
    const float2 v = merge(merge(s8[(2 * y) * 8 + 2 * x], s8[(2 * y) * 8 + 2 * x + 1]), merge(s8[(2 * y + 1) * 8 + 2 * x], s8[(2 * y + 1) * 8 + 2 * x + 1]));
    s16[t] = v; put(1, x, y, v);
  }
  __syncthreads();
  if (t < 4) {
    const int x = t & 1, y = t >> 1;
    const float2 v = merge(merge(s16[(2 * y) * 4 + 2 * x], s16[(2 * y) * 4 + 2 * x + 1]), merge(s16[(2 * y + 1) * 4 + 2 * x], s16[(2 * y + 1) * 4 + 2 * x + 1]));
    s32[t] = v; put(2, x, y, v);
  }
  __syncthreads();
  if (t == 0) put(3, 0, 0, merge(merge(s32[0], s32[1]), merge(s32[2], s32[3])));
}

// What the classification of a dense launch needs beside FuseParams.  Camera-space position of a group's centre voxel index
// (x0 + 1.5, y, z): pc = C + x0 * Ax + y * Ay + z * Az (float, evaluated with FMAs: an approximation with a known error bound,
// never stored).  h*: half extent of the group along its x run in camera space, 1.5 * voxelSize * |M column 0|.
struct GroupClassify {
  const float2* tiles;
  TileLevels tl;
  float Ax[3], Ay[3], Az[3], C[3];
  float hx, hy, hz;
  float slackZ;        // absolute slack on pc.z (metres): 1e-5 of the magnitudes that enter it
  int enabled;
};
enum { kGroupMixed = 0, kGroupFree = 1, kGroupShadow = 2 };

__device__ inline int classify_group(const GroupClassify& g, const FuseParams& p, int x0, int y, int z) {
  const float fx0 = (float)x0, fy = (float)y, fz = (float)z;
  const float pcx = __builtin_fmaf(g.Ax[0], fx0, __builtin_fmaf(g.Ay[0], fy, __builtin_fmaf(g.Az[0], fz, g.C[0])));
  const float pcy = __builtin_fmaf(g.Ax[1], fx0, __builtin_fmaf(g.Ay[1], fy, __builtin_fmaf(g.Az[1], fz, g.C[1])));
  const float pcz = __builtin_fmaf(g.Ax[2], fx0, __builtin_fmaf(g.Ay[2], fy, __builtin_fmaf(g.Az[2], fz, g.C[2])));
  const float ez = g.hz + g.slackZ;
  const float zmin = pcz - ez, zmax = pcz + ez;
  if (!(zmin > 1e-3f)) return kGroupMixed;                       // near or behind the camera plane (or NaN): exact path
  const float rz = __builtin_amdgcn_rcpf(pcz), rzmin = __builtin_amdgcn_rcpf(zmin) * 1.00001f;
  const float tx = pcx * rz, ty = pcy * rz;
  const float uc = __builtin_fmaf(p.fx, tx, p.cx), vc = __builtin_fmaf(p.fy, ty, p.cy);
  // |u(voxel) - u(centre)| <= fx (hx + |x / z| hz) / zmin exactly; + 0.5 for (int)(u + 0.5), + 1 pixel for the approximations here
  const float ru = fabsf(p.fx) * (g.hx + fabsf(tx) * g.hz) * rzmin + 1.5f + 1e-5f * fabsf(uc);
  const float rv = fabsf(p.fy) * (g.hy + fabsf(ty) * g.hz) * rzmin + 1.5f + 1e-5f * fabsf(vc);
  const float r = fmaxf(ru, rv);
  if (!(r <= 31.5f)) return kGroupMixed;
  const int level = (r <= 3.5f) ? 0 : (r <= 7.5f) ? 1 : (r <= 15.5f) ? 2 : 3;
  const float ulo = uc - ru, uhi = uc + ru, vlo = vc - rv, vhi = vc + rv;
  if (!(uhi >= 0.0f && vhi >= 0.0f && ulo <= (float)(p.W - 1) && vlo <= (float)(p.H - 1))) return kGroupMixed;   // footprint off the image: the column cull's business
  const bool inside = ulo >= 1.0f && vlo >= 1.0f && uhi <= (float)(p.W - 2) && vhi <= (float)(p.H - 2);
  const int sh = 3 + level;
  const int px0 = max((int)floorf(ulo), 0) >> sh, px1 = min((int)floorf(uhi), p.W - 1) >> sh;
  const int py0 = max((int)floorf(vlo), 0) >> sh, py1 = min((int)floorf(vhi), p.H - 1) >> sh;
  const float2* __restrict__ t = g.tiles + g.tl.off[level];
  const int tw = g.tl.tw[level];
  const float2 a = t[py0 * tw + px0], b = t[py0 * tw + px1], c = t[py1 * tw + px0], d = t[py1 * tw + px1];
  const float mn = fminf(fminf(a.x, b.x), fminf(c.x, d.x)), mx = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
  const float slack = 1e-5f * (fabsf(mx) + fabsf(mn) + zmax) + 1e-6f * p.mu;
  if (inside && (mn - zmax >= p.mu + slack)) return kGroupFree;        // every pixel valid (mn > 0 follows) and at least mu behind every voxel
  if (mx - zmin < -p.mu - slack) return kGroupShadow;                  // every pixel invalid or more than mu in front of every voxel
  return kGroupMixed;
}

// Dense volume of ITMVoxel_s with sx % 4 == 0: 4 voxels (16 B) per lane, one 1 KiB row segment per
// wave instruction; a group is written back only if one of its voxels changed.  blockIdx.y = z slice,
// blockIdx.x splits the (y, x/4) plane; no 64-bit index division in the loop, two 16-byte loads in
// flight per lane.
// Frustum planes of a dense launch in VOXEL-INDEX space, computed by the host in double:
//   g_k(xi, yi, zi) = a_k xi + b_k yi + cz_k zi + cm_k,   k = left, right, top, bottom, front
// with cm_k = c_k + margin_k, so that g_k < 0 for a voxel implies that the exact float test of fuse_depth_project rejects it
// (pc.z <= 0, or u / v outside [1, W-2] x [1, H-2]): margin_k = |a_k| + |b_k| (one voxel of slack along x and y) plus 1e-4 of
// the magnitudes that enter g_k, three orders above the rounding of the float evaluation.  nb_k = -1 / b_k (0 when b_k == 0).
struct ColumnCull {
  double a[5], nb[5], cz[5], cm[5];
  int kind[5];   // sign of b_k: +1 the plane bounds yi from below, -1 from above, 0 not at all (then g_k must be >= 0 as it is)
};

// Rows [rlo, rhi] of the column of groups (x0 .. x0 + 3, slice z) in which some voxel can pass the exact test (empty when rlo > rhi).
// Host and device: the same double-precision operations (fma is exactly rounded on both), so the host-side property test
// (tests/test_dense_cull.py, itm_debug_column_cull_rows) checks what the kernel computes.
__host__ __device__ inline void column_rows(const ColumnCull& cc, int x0, int z, int& rlo, int& rhi) {
  double ylo = -1e9, yhi = 1e9;
  bool none = false;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    // the largest value g_k takes on the group: at x0 + 3 when a_k > 0, at x0 otherwise
    const double t = __builtin_fma(cc.a[k], (double)(cc.a[k] > 0.0 ? x0 + 3 : x0), __builtin_fma(cc.cz[k], (double)z, cc.cm[k]));
    if (cc.kind[k] > 0) ylo = fmax(ylo, t * cc.nb[k]);          // b_k yi + t >= 0  <=>  yi >= -t / b_k
    else if (cc.kind[k] < 0) yhi = fmin(yhi, t * cc.nb[k]);     //                  <=>  yi <= -t / b_k   (b_k < 0)
    else none |= t < 0.0;
  }
  rlo = none ? 0x7fffffff : (int)ceil(fmax(ylo, -1e9));
  rhi = (int)floor(fmin(yhi, 1e9));
}

// CULL: 0 = per group with the exact arithmetic of two end voxels (any size); 1 = the rows [ylo, yhi] that can be inside the
// frustum are computed ONCE per thread for its column of groups (every group a thread visits has the same x0 when the stride
// is a multiple of the row length), the per-group test is then two integer comparisons.  Measured on BASELINE configs[2]
// (512^3): the per-group test was ~40 % of the kernel's VALU work (70 instructions for each of 33.5 M groups).
// CLASSIFY: 0 = every fetched group takes the exact per-voxel path; 1 = groups are classified against the depth tiles BEFORE they
// are fetched (shadow groups are never read); 2 = after the fetch, and only groups that still have an unsaturated voxel;
// 3 = measurement of the classification itself (itm_debug_dense_classify_check): classify, run the exact path, count disagreements.
__device__ int g_classifyCheck[4];     // CLASSIFY == 3: free groups, shadow groups, mixed groups, VIOLATIONS
template <bool POW2, int CULL, int CLASSIFY>
__global__ void __launch_bounds__(256) integrate_dense_s_x4_kernel(uint4* __restrict__ vba, const float* __restrict__ depth, FuseParams p,
                                                                   int sx, int sy, int sz, int ox, int oy, int oz, int log2sx4, ColumnCull cc, GroupClassify gc) {
  const int sx4 = sx >> 2;
  const int z = blockIdx.y;
  const int plane = sx4 * sy;                       // groups per z slice
  uint4* __restrict__ slice = vba + (size_t)z * plane;
  const float mz = (float)(z + oz) * p.voxelSize;
  const int stride = gridDim.x * 256;
  // free-space group: the observation of every voxel is exactly 1 (fuse_depth_update with eta >= mu)
  auto update_free = [&](int idx, uint4 q) {
    uint32_t v[4] = {q.x, q.y, q.z, q.w};
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w = (v[k] >> 16) & 0xffu;
      if (p.stopAtMax && (int)w == p.maxW) continue;
      uint32_t nv;
      if ((v[k] & 0xffffu) == 32767u) {            // (w * 1.0 + 1) / (w + 1) == 1.0 exactly: the sdf stays, the weight counts
        const int nw = ((int)w + 1 < p.maxW) ? (int)w + 1 : p.maxW;
        nv = 32767u | ((uint32_t)(nw & 0xff) << 16);
      } else {
        nv = v[k];
        fuse_average<VoxelS>(nv, 1.0f, p);
      }
      any |= nv != v[k];
      v[k] = nv;
    }
    if (any) slice[idx] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto saturated = [&](const uint4& q) {
    const uint32_t m = (uint32_t)(p.maxW & 0xff);
    return p.stopAtMax && ((q.x >> 16) & 0xffu) == m && ((q.y >> 16) & 0xffu) == m && ((q.z >> 16) & 0xffu) == m && ((q.w >> 16) & 0xffu) == m;
  };
  auto group_class = [&](int idx) {
    const int y = POW2 ? (idx >> log2sx4) : (idx / sx4);
    const int x0 = (POW2 ? (idx & (sx4 - 1)) : (idx - y * sx4)) * 4;
    return classify_group(gc, p, x0, y, z);
  };
  auto process = [&](int idx, uint4 q, int cls) {
    if constexpr (CLASSIFY == 2) {
      if (saturated(q)) return;
      cls = group_class(idx);
    }
    if constexpr (CLASSIFY == 1 || CLASSIFY == 2) {
      if (cls == kGroupShadow) return;
      if (cls == kGroupFree) { update_free(idx, q); return; }
    }
    const int y = POW2 ? (idx >> log2sx4) : (idx / sx4);
    const int x0 = (POW2 ? (idx & (sx4 - 1)) : (idx - y * sx4)) * 4;
    uint32_t v[4] = {q.x, q.y, q.z, q.w};
    const float my = (float)(y + oy) * p.voxelSize;
    // project the four voxels, gather their depth pixels together (four independent loads in flight), then update
    int pix[4]; float pcz[4], dm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pix[k] = -1;
      if (p.stopAtMax && VoxelS::w_depth(v[k]) == p.maxW) continue;
      pix[k] = fuse_depth_project((float)(x0 + k + ox) * p.voxelSize, my, mz, p, pcz[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) dm[k] = (pix[k] >= 0) ? depth[pix[k]] : 0.0f;
    bool any = false;
    [[maybe_unused]] int bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool touched = false;
      [[maybe_unused]] float eta = -1.0f;
      if (pix[k] >= 0) eta = fuse_depth_update<VoxelS>(v[k], dm[k], pcz[k], p, touched);
      if constexpr (CLASSIFY == 3) {
        const bool skipped = p.stopAtMax && VoxelS::w_depth((&q.x)[k]) == p.maxW;
        // free: every voxel that is not skipped must have been updated with an observation that clamps to 1
        if (cls == kGroupFree && !skipped && !(touched && dm[k] > 0.0f && (p.muFast ? div_markstein(eta, p.mu, p.rcpMu) : eta / p.mu) >= 1.0f)) ++bad;
        if (cls == kGroupShadow && touched) ++bad;
      }
      any |= touched;
    }
    if constexpr (CLASSIFY == 3) {
      atomicAdd(&g_classifyCheck[cls == kGroupFree ? 0 : cls == kGroupShadow ? 1 : 2], 1);
      if (bad) atomicAdd(&g_classifyCheck[3], bad);
      if (cls == kGroupFree) {                      // and the shortcut must produce the exact path's words
        uint32_t f[4] = {q.x, q.y, q.z, q.w};
        for (int k = 0; k < 4; ++k) {
          const uint32_t w = (f[k] >> 16) & 0xffu;
          if (p.stopAtMax && (int)w == p.maxW) continue;
          if ((f[k] & 0xffffu) == 32767u) { const int nw = ((int)w + 1 < p.maxW) ? (int)w + 1 : p.maxW; f[k] = 32767u | ((uint32_t)(nw & 0xff) << 16); }
          else fuse_average<VoxelS>(f[k], 1.0f, p);
          if (f[k] != v[k]) atomicAdd(&g_classifyCheck[3], 1);
        }
      }
    }
    if (any) slice[idx] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  // Conservative cull of a whole 4-voxel group BEFORE its 16 bytes are fetched: camera-space
  // coordinates are affine along x, so if both end voxels of the group lie outside the same frustum
  // plane (with half a pixel of margin, as in fuse_depth) every voxel of the group is rejected by the
  // exact per-voxel test as well.  ~70 % of a 512^3 volume around the camera is culled this way.
  auto culled = [&](int idx) {
    const int y = POW2 ? (idx >> log2sx4) : (idx / sx4);
    const int x0 = (POW2 ? (idx & (sx4 - 1)) : (idx - y * sx4)) * 4;
    const float my = (float)(y + oy) * p.voxelSize;
    const Vec3 a = transform_point(p.M_d, (float)(x0 + ox) * p.voxelSize, my, mz);
    const Vec3 b = transform_point(p.M_d, (float)(x0 + 3 + ox) * p.voxelSize, my, mz);
    const float lox = 0.5f - p.cx, hix = (float)(p.W - 2) + 0.5f - p.cx;
    const float loy = 0.5f - p.cy, hiy = (float)(p.H - 2) + 0.5f - p.cy;
    const float eps = 1e-3f * p.voxelSize;                 // margin on the z <= 0 plane
    const float atx = p.fx * a.x, aty = p.fy * a.y, btx = p.fx * b.x, bty = p.fy * b.y;
    // the four side planes are only meaningful for points in front of the camera
    const bool front = (a.z > eps) && (b.z > eps);
    return ((a.z < -eps) && (b.z < -eps)) ||
           (front && (((atx < lox * a.z) && (btx < lox * b.z)) || ((atx > hix * a.z) && (btx > hix * b.z)) ||
                      ((aty < loy * a.z) && (bty < loy * b.z)) || ((aty > hiy * a.z) && (bty > hiy * b.z))));
  };
  int idx = blockIdx.x * 256 + threadIdx.x;
  if constexpr (CULL == 1) {
    // rows of this thread's column (x0 .. x0 + 3, slice z) that may be inside the frustum
    int rlo, rhi;
    column_rows(cc, (idx & (sx4 - 1)) * 4, z, rlo, rhi);
    auto outside = [&](int i) { const int y = i >> log2sx4; return y < rlo || y > rhi; };
    if constexpr (CLASSIFY == 1 || CLASSIFY == 3) {
      // classified before the fetch: a shadow group is never read (CLASSIFY == 3 reads everything: it checks the classes)
      for (; idx + stride < plane; idx += 2 * stride) {
        const int k0 = outside(idx) ? -1 : group_class(idx), k1 = outside(idx + stride) ? -1 : group_class(idx + stride);
        const bool f0 = k0 >= 0 && (CLASSIFY == 3 || k0 != kGroupShadow), f1 = k1 >= 0 && (CLASSIFY == 3 || k1 != kGroupShadow);
        uint4 q0, q1;
        if (f0) q0 = slice[idx];
        if (f1) q1 = slice[idx + stride];
        if (f0) process(idx, q0, k0);
        if (f1) process(idx + stride, q1, k1);
      }
      if (idx < plane && !outside(idx)) { const int k0 = group_class(idx); if (CLASSIFY == 3 || k0 != kGroupShadow) process(idx, slice[idx], k0); }
      return;
    }
    for (; idx + stride < plane; idx += 2 * stride) {
      const bool c0 = outside(idx), c1 = outside(idx + stride);
      uint4 q0, q1;
      if (!c0) q0 = slice[idx];
      if (!c1) q1 = slice[idx + stride];
      if (!c0) process(idx, q0, kGroupMixed);
      if (!c1) process(idx + stride, q1, kGroupMixed);
    }
    if (idx < plane && !outside(idx)) process(idx, slice[idx], kGroupMixed);
    return;
  }
  for (; idx + stride < plane; idx += 2 * stride) {
    const bool c0 = culled(idx), c1 = culled(idx + stride);
    uint4 q0, q1;
    if (!c0) q0 = slice[idx];
    if (!c1) q1 = slice[idx + stride];
    if (!c0) process(idx, q0, kGroupMixed);
    if (!c1) process(idx + stride, q1, kGroupMixed);
  }
  if (idx < plane && !culled(idx)) process(idx, slice[idx], kGroupMixed);
}

// ---- the strip kernel ------------------------------------------------------------------------------------------------------------
// What the counters say about the kernel above (profiles/r3_dense_counters.md, BASELINE configs[2]): 90 M vector instructions per
// launch, issue bound, and hardly fewer (81 M) once the weights have reached maxW -- the voxels in the SHADOW of the surface are
// never updated, so they never saturate and are projected in every frame, and a wave whose 64 lanes lie along x executes the
// projection for all of them as long as one lane needs it.  Classifying the groups against the depth tiles does not help that
// kernel (measured: 119 M instructions): its waves straddle the class boundaries, so nearly every wave still runs the exact path,
// now on top of the classification.
// Here the exact path is taken by PACKED lanes: a wave owns a strip (64 adjacent columns of 4-voxel groups of one z slice, 1 KiB of
// every row), takes four consecutive rows at a time -- a PATCH of 4 x 4 voxels per lane, classified once --, updates free-space patches
// in place, and QUEUES the groups of the patches that need the exact arithmetic in LDS; whenever 64 are waiting, they are taken off the queue and projected / gathered / averaged with every lane
// busy.  Shadow groups and saturated groups cost their classification only.  A strip is dealt out row by row over kStripPhases
// work items so that the waves working on it at one time read adjacent rows (waves each walking a row chunk 64 KB apart took
// 290 us for a 512^3 launch, whatever the arithmetic: all of them at the same offset modulo the chunk at the same time); items go
// round-robin over persistent waves, far slices first (a shared work counter was measured too: an atomic with a return value on
// one address is ~16 ns, serialised at the memory side -- 350 us for the 24 k draws of a launch).
constexpr int kStripPhases = 16;                 // work items per strip
constexpr int kStripQueue = 128;                 // queued groups per wave: fewer than 64 left over + at most 64 of one row

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1))) integrate_dense_strip_kernel(uint4* __restrict__ vba, const float* __restrict__ depth, FuseParams p,
                                                                    int sx, int sy, int sz, int ox, int oy, int oz, ColumnCull cc, GroupClassify gc) {
  __shared__ uint4 sQ[4][kStripQueue];
  __shared__ int sTag[4][kStripQueue];
  const int sx4 = sx >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xblocks = (sx4 + 63) >> 6;
  const unsigned int perSlice = (unsigned int)(xblocks * kStripPhases);
  const unsigned int nItems = (unsigned int)sz * perSlice;
  const unsigned int nWaves = gridDim.x * (blockDim.x >> 6);
  for (unsigned int item = blockIdx.x * (blockDim.x >> 6) + wave; item < nItems; item += nWaves) {
    const int zi = (int)(item / perSlice), rem = (int)(item - (unsigned int)zi * perSlice);
    const int z = sz - 1 - zi;                                // far slices first
    const int phase = rem / xblocks, xb = rem - phase * xblocks;
    const int x4 = xb * 64 + lane, x0 = x4 * 4;
    uint4* __restrict__ slice = vba + (size_t)z * sx4 * sy;
    const float mz = (float)(z + oz) * p.voxelSize;
    int rlo, rhi;
    column_rows(cc, x0, z, rlo, rhi);
    rlo = max(rlo, 0); rhi = min(rhi, sy - 1);
    if (x4 >= sx4) { rlo = 1; rhi = 0; }                       // a row need not be a multiple of 64 groups
    // rows any lane of the wave needs (the loop bounds are the wave's, the test inside is the lane's)
    int wlo = (rlo <= rhi) ? rlo : 0x7fffffff, whi = (rlo <= rhi) ? rhi : -1;
#pragma unroll
    for (int off = 32; off; off >>= 1) { wlo = min(wlo, __shfl_xor(wlo, off)); whi = max(whi, __shfl_xor(whi, off)); }
    if (wlo > whi) continue;

    // the exact path for one queued group: tag = row * 64 + lane that queued it
    auto exact = [&](uint4 q, int tag) {
      const int y = tag >> 6, gx4 = xb * 64 + (tag & 63), gx0 = gx4 * 4;
      uint32_t v[4] = {q.x, q.y, q.z, q.w};
      const float my = (float)(y + oy) * p.voxelSize;
      int pix[4]; float pcz[4], dm[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pix[k] = -1;
        if (p.stopAtMax && VoxelS::w_depth(v[k]) == p.maxW) continue;
        pix[k] = fuse_depth_project((float)(gx0 + k + ox) * p.voxelSize, my, mz, p, pcz[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) dm[k] = (pix[k] >= 0) ? depth[pix[k]] : 0.0f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool touched = false;
        if (pix[k] >= 0) fuse_depth_update<VoxelS>(v[k], dm[k], pcz[k], p, touched);
        any |= touched;
      }
      if (any) slice[(size_t)y * sx4 + gx4] = make_uint4(v[0], v[1], v[2], v[3]);
    };
    int queued = 0;                                            // the same in every lane
    auto take = [&](int n) {                                   // the last n (<= 64) queued groups
      __builtin_amdgcn_wave_barrier();
      if (lane < n) exact(sQ[wave][queued - n + lane], sTag[wave][queued - n + lane]);
      __builtin_amdgcn_wave_barrier();
      queued -= n;
    };

    const uint4* __restrict__ dummy = slice + (size_t)wlo * sx4 + xb * 64;     // what a lane outside its interval reads (one line, dropped)
    const uint32_t wmax = (uint32_t)(p.maxW & 0xff);
    // rows in PATCHES of four: patch G = rows 4 G .. 4 G + 3; this item takes the patches G = phase (mod kStripPhases)
    for (int G = (wlo >> 2) + (((phase - (wlo >> 2)) % kStripPhases + kStripPhases) % kStripPhases); 4 * G <= whi; G += kStripPhases) {
      const int y = 4 * G;
      uint4 qa[4];
      bool need[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = y + u >= rlo && y + u <= rhi;
        const uint4* a = in ? slice + (size_t)(y + u) * sx4 + x4 : dummy;
        qa[u] = *a;
        need[u] = in;
      }
      bool any = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint4 q = qa[u];
        const bool sat = p.stopAtMax && ((q.x >> 16) & 0xffu) == wmax && ((q.y >> 16) & 0xffu) == wmax && ((q.z >> 16) & 0xffu) == wmax && ((q.w >> 16) & 0xffu) == wmax;
        need[u] = need[u] && !sat;
        any |= need[u];
      }
      if (!__any(any)) continue;
      // one class for the 4 x 4 patch (gc was made for patches: centre (x0 + 1.5, y + 1.5), extents of both runs)
      const int cls = any ? classify_group(gc, p, x0, y, z) : kGroupShadow;
      if (cls == kGroupFree) {
        // every voxel of the patch observes free space: the observation is exactly 1 (fuse_depth_update with eta >= mu)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!need[u]) continue;
          uint32_t v[4] = {qa[u].x, qa[u].y, qa[u].z, qa[u].w};
          bool changed = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t w = (v[k] >> 16) & 0xffu;
            if (p.stopAtMax && w == wmax) continue;
            uint32_t nv;
            if ((v[k] & 0xffffu) == 32767u) {          // (w * 1.0 + 1) / (w + 1) == 1.0 exactly: the sdf stays, the weight counts
              const int nw = ((int)w + 1 < p.maxW) ? (int)w + 1 : p.maxW;
              nv = 32767u | ((uint32_t)(nw & 0xff) << 16);
            } else {
              nv = v[k];
              fuse_average<VoxelS>(nv, 1.0f, p);
            }
            changed |= nv != v[k];
            v[k] = nv;
          }
          if (changed) slice[(size_t)(y + u) * sx4 + x4] = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
      if (!__any(cls == kGroupMixed)) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool mixed = cls == kGroupMixed && need[u];
        const unsigned long long m = __ballot(mixed);
        if (m) {
          if (mixed) {
            const int pos = queued + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            sQ[wave][pos] = qa[u]; sTag[wave][pos] = (y + u) * 64 + lane;
          }
          queued += __popcll(m);
          if (queued >= 64) take(64);
        }
      }
    }
    if (queued > 0) take(queued);
  }
}

// the classification's inputs for one launch; false when a coefficient is not finite (then nothing is classified)
// `rows`: 1 = a class per 4-voxel group; 4 = a class per PATCH of four such groups in consecutive rows (the strip kernel)
static bool make_group_classify(const FuseParams& p, const int* size, const int* off, const float2* tiles, const TileLevels& tl, GroupClassify& g, int rows = 1) {
  memset(&g, 0, sizeof g);
  g.tiles = tiles; g.tl = tl;
  const double vs = p.voxelSize;
  double mag = 0.0;
  for (int j = 0; j < 3; ++j) {
    const double m0 = p.M_d.m[j], m1 = p.M_d.m[j + 4], m2 = p.M_d.m[j + 8], m3 = p.M_d.m[j + 12];
    g.Ax[j] = (float)(m0 * vs); g.Ay[j] = (float)(m1 * vs); g.Az[j] = (float)(m2 * vs);
    g.C[j] = (float)((m0 * (off[0] + 1.5) + m1 * (off[1] + 0.5 * (rows - 1)) + m2 * off[2]) * vs + m3);       // centre of the run x0 .. x0 + 3 (of the rows y .. y + rows - 1)
    if (j == 2) mag = (fabs(m0) * (fabs((double)off[0]) + size[0]) + fabs(m1) * (fabs((double)off[1]) + size[1]) + fabs(m2) * (fabs((double)off[2]) + size[2])) * vs + fabs(m3);
    if (!std::isfinite(g.Ax[j]) || !std::isfinite(g.Ay[j]) || !std::isfinite(g.Az[j]) || !std::isfinite(g.C[j])) return false;
  }
  const double ry = 0.5 * (rows - 1);      // half extent along y in voxels
  g.hx = (float)((1.5 * fabs((double)p.M_d.m[0]) + ry * fabs((double)p.M_d.m[4])) * vs * 1.0001);
  g.hy = (float)((1.5 * fabs((double)p.M_d.m[1]) + ry * fabs((double)p.M_d.m[5])) * vs * 1.0001);
  g.hz = (float)((1.5 * fabs((double)p.M_d.m[2]) + ry * fabs((double)p.M_d.m[6])) * vs * 1.0001);
  g.slackZ = (float)(1e-5 * mag + 1e-7);
  g.enabled = 1;
  return std::isfinite(g.slackZ) && std::isfinite(p.fx) && std::isfinite(p.fy) && std::isfinite(p.cx) && std::isfinite(p.cy);
}

// The five frustum planes of ColumnCull from the launch parameters; false when a coefficient is not finite (then the per-group
// test runs).  pc_j = M[j] mx + M[j+4] my + M[j+8] mz + M[j+12] with m = (index + offset) * voxelSize is affine in the indices:
//   left   fx pc_x - (1 - cx) pc_z >= 0      right   ((W - 2) - cx) pc_z - fx pc_x >= 0
//   top    fy pc_y - (1 - cy) pc_z >= 0      bottom  ((H - 2) - cy) pc_z - fy pc_y >= 0      front  pc_z > 0
static bool make_column_cull(const FuseParams& p, const int* size, const int* off, ColumnCull& cc) {
  memset(&cc, 0, sizeof cc);
  const double vs = p.voxelSize;
  double Ax[3], Ay[3], Az[3], C[3];
  for (int j = 0; j < 3; ++j) {
    const double m0 = p.M_d.m[j], m1 = p.M_d.m[j + 4], m2 = p.M_d.m[j + 8], m3 = p.M_d.m[j + 12];
    Ax[j] = m0 * vs; Ay[j] = m1 * vs; Az[j] = m2 * vs;
    C[j] = (m0 * off[0] + m1 * off[1] + m2 * off[2]) * vs + m3;
  }
  const double lox = 1.0 - (double)p.cx, hix = (double)(p.W - 2) - (double)p.cx, loy = 1.0 - (double)p.cy, hiy = (double)(p.H - 2) - (double)p.cy;
  const double fx = p.fx, fy = p.fy;
  // plane k as weights (wx, wy, wz) on (pc_x, pc_y, pc_z)
  const double w[5][3] = {{fx, 0, -lox}, {-fx, 0, hix}, {0, fy, -loy}, {0, -fy, hiy}, {0, 0, 1}};
  for (int k = 0; k < 5; ++k) {
    const double a = w[k][0] * Ax[0] + w[k][1] * Ax[1] + w[k][2] * Ax[2];
    const double b = w[k][0] * Ay[0] + w[k][1] * Ay[1] + w[k][2] * Ay[2];
    const double cz = w[k][0] * Az[0] + w[k][1] * Az[1] + w[k][2] * Az[2];
    const double c = w[k][0] * C[0] + w[k][1] * C[1] + w[k][2] * C[2];
    // magnitudes that enter g_k (not their sum, which may cancel)
    double mag = 0.0;
    for (int j = 0; j < 3; ++j)
      mag += fabs(w[k][j]) * (fabs(Ax[j]) * size[0] + fabs(Ay[j]) * size[1] + fabs(Az[j]) * size[2] +
                              (fabs((double)p.M_d.m[j] * off[0]) + fabs((double)p.M_d.m[j + 4] * off[1]) + fabs((double)p.M_d.m[j + 8] * off[2])) * vs + fabs((double)p.M_d.m[j + 12]));
    const double margin = fabs(a) + fabs(b) + 1e-4 * mag;
    if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(cz) || !std::isfinite(c) || !std::isfinite(margin)) return false;
    cc.a[k] = a; cc.cz[k] = cz; cc.cm[k] = c + margin;
    cc.kind[k] = (b > 0.0) ? 1 : (b < 0.0) ? -1 : 0;
    cc.nb[k] = (b != 0.0) ? -1.0 / b : 0.0;
    if (!std::isfinite(cc.nb[k])) { cc.kind[k] = 0; cc.nb[k] = 0.0; cc.cm[k] += fabs(b) * size[1]; }   // a slope too small to divide by: let the plane pass
  }
  return true;
}

// `fuseProjection`: also run the projection half of CreateExpectedDepths (hash scenes whose sub-sampled range image
// fits four times in LDS; the caller checked can_fuse_projection and launches range_reduce afterwards).
// what an integration of view `v` would be refused for (checked at once by the entry point that records the call, pending.hip)
int validate_integrate(const itm_scene* s, const itm_view* v) {
  const bool colour = (s->cfg.voxelType == ITM_VOXEL_S_RGB || s->cfg.voxelType == ITM_VOXEL_F_RGB);
  if (colour && (!v->rgb || v->w_rgb <= 0 || v->h_rgb <= 0)) return set_error(ITM_ERR_INVALID, "colour voxels need an rgb image");
  return ITM_OK;
}

int launch_integrate(itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st, bool fuseProjection) {
  FuseParams p;
  memcpy(p.M_d.m, v->M_d, 64);
  matmul4(v->rgb_to_depth_inv, v->M_d, p.M_rgb.m);  // calib_inv * M_d (_CPU.cpp:61)
  p.fx = v->intr_d[0]; p.fy = v->intr_d[1]; p.cx = v->intr_d[2]; p.cy = v->intr_d[3];
  p.fxc = v->intr_rgb[0]; p.fyc = v->intr_rgb[1]; p.cxc = v->intr_rgb[2]; p.cyc = v->intr_rgb[3];
  p.mu = s->prm.mu; p.voxelSize = s->prm.voxelSize; p.maxW = s->prm.maxW;
  {
    // 1/mu correctly rounded (host division); the fast quotient needs a normal mu whose significand is
    // not all ones and a magnitude that keeps eta/mu far from overflow/underflow
    p.rcpMu = 1.0f / p.mu;
    uint32_t bits; memcpy(&bits, &p.mu, 4);
    p.muFast = ((bits & 0x7fffffu) != 0x7fffffu) && p.mu >= 1e-6f && p.mu <= 1e6f;
  }
  p.W = v->w; p.H = v->h; p.Wc = v->w_rgb; p.Hc = v->h_rgb;
  p.stopAtMax = s->prm.stopIntegratingAtMaxW;
  p.org = s->org;
  p.idsCap = rs->capIds > 0 ? rs->capIds : 1;
  const uchar4* rgb = (const uchar4*)v->rgb;
  { const int rc = validate_integrate(s, v); if (rc) return rc; }

  // (the integration timer brackets the integration kernel alone; the small depth-tile launch of the dense path goes in front of it)
  std::unique_ptr<KernelTimer> tk;
  if (s->cfg.indexType == ITM_INDEX_HASH) {
    tk.reset(new KernelTimer(s, ITM_TK_INTEGRATE, st));
    // 2048 workgroups of 8 waves: more waves than the chip holds at once (768-1024 workgroups), so that the dispatcher evens out what
    // the static striding does not, but few enough that most waves have work -- a workgroup without any still holds a slot for ~1 us
    // (measured, configs[4] / configs[1]: 768-1024 workgroups 189 / 21.3 us, 1536: 177 / 19.9, 2048: 175 / 20.3, 4096: 170 / 20.4,
    // where the last workgroup of configs[1] only STARTS after 15 us)
    // re-measured at the end of round 4 on the final kernels (ITMVoxel_s with a whole block per item): 640 x 480 (9.6 k visible blocks)
    // 1024: 19.5-19.8 us, 1536: 19.4-19.8, 2048: 19.0-19.5, 3072: 18.9-19.8, 8192: 18.2-19.6; 1280 x 960 colour (55.8 k blocks) 1024: 193 us,
    // 2048: 178, 3072: 174, 4096: 173-174, 6144: 172-173, 8192: 171 -- the number of visible blocks is not known on the host (nothing is read
    // back), the image size stands in for it
    const int grid = g_debug_integrate_wgs > 0 ? g_debug_integrate_wgs : ((size_t)rs->w * rs->h > (size_t)640 * 480 ? 8192 : 2048);
    ProjParams pp;
    const int RW = (rs->w + 7) / 8, RH = (rs->h + 7) / 8;
    if (fuseProjection) {
      memcpy(pp.M.m, v->M_d, 64);
      pp.fx = v->intr_d[0]; pp.fy = v->intr_d[1]; pp.cx = v->intr_d[2]; pp.cy = v->intr_d[3];
      pp.voxelSize = s->prm.voxelSize; pp.W = rs->w; pp.H = rs->h; pp.maxBlocks = s->cfg.maxRenderingBlocks;
    }
    int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
      using VX = decltype(vx);
      if (fuseProjection)
        integrate_project_kernel<VX><<<grid, 512, (size_t)RW * RH * sizeof(uint2), st>>>(rs->visibleIds, rs->counters, s->hash, s->vba, s->sdfMirror, v->depth, rgb, p,
                                                                                         rs->range, rs->projBuf, rs->rangePartials, pp, RW, RH);
      else integrate_hash_kernel<VX><<<grid * (512 / ITM_INTEGRATE_BLOCK), ITM_INTEGRATE_BLOCK, 0, st>>>(rs->visibleIds, rs->counters, s->hash, s->vba, s->sdfMirror, v->depth, rgb, p);
      return ITM_OK;
    });
    if (rc) return rc;
  } else {
    const int* sz = s->cfg.denseSize; const int* of = s->cfg.denseOffset;
    if (s->cfg.voxelType == ITM_VOXEL_S && (sz[0] % 4) == 0) {
      const int sx4 = sz[0] / 4;
      int lg = 0; while ((1 << lg) < sx4) ++lg;
      const bool pow2 = (1 << lg) == sx4;
      const int plane = sx4 * sz[1];
      int splits = (plane + 2 * 256 - 1) / (2 * 256);        // every lane gets ~2 groups per pass
      if (splits > 64) splits = 64;
      if (g_debug_dense_no_strips > 1) splits = g_debug_dense_no_strips;      // measurement: debug key 17 = n > 1 sets the number of splits of a slice
      if (splits < 1) splits = 1;
      const dim3 grid(splits, sz[2]);
      ColumnCull cc;
      const bool planes = !g_debug_dense_group_cull && make_column_cull(p, sz, of, cc);
      const bool columns = pow2 && ((splits * 256) % sx4) == 0 && planes;
      GroupClassify gc;
      memset(&gc, 0, sizeof gc);
      int classify = 0;
      const bool wantStrips = planes && !g_debug_dense_no_strips && g_debug_dense_classify != 1 && g_debug_dense_classify != 3 && sz[1] < (1 << 24);
      if (planes && g_debug_dense_classify != 1) {
        // min / max depth tiles of this frame (one small launch), then groups are classified against them
        const TileLevels tl = tile_levels(v->w, v->h);
        if (s->depthTilesCap < (size_t)tl.total) {
          if (s->depthTiles) (void)hipFree(s->depthTiles);
          s->depthTiles = nullptr; s->depthTilesCap = 0;
          if (hipMalloc((void**)&s->depthTiles, (size_t)tl.total * sizeof(float2)) == hipSuccess) s->depthTilesCap = (size_t)tl.total;
          else (void)hipGetLastError();
        }
        if (s->depthTiles && make_group_classify(p, sz, of, s->depthTiles, tl, gc, wantStrips ? 4 : 1)) {
          depth_tiles_kernel<<<dim3((v->w + 63) / 64, (v->h + 63) / 64), 256, 0, st>>>(v->depth, v->w, v->h, s->depthTiles, tl);
          classify = g_debug_dense_classify == 2 ? 2 : g_debug_dense_classify == 3 ? 3 : 1;
        }
      }
      tk.reset(new KernelTimer(s, ITM_TK_INTEGRATE, st));
#define ITM_DENSE(P2, CU, CL) integrate_dense_s_x4_kernel<P2, CU, CL><<<grid, 256, 0, st>>>((uint4*)s->vba, v->depth, p, sz[0], sz[1], sz[2], of[0], of[1], of[2], lg, cc, gc)
      // the strip kernel: volumes whose rows are whole strips of 64 groups (debug key 17 keeps the launch shape of rounds 1-2)
      const bool strips = wantStrips && classify != 0;
      if (strips) {
        // 16 384 waves for the 16 384 items of a 512^3 volume: one item each, dispatched as slots free up (measured, BASELINE configs[2]:
        // 1 024 workgroups 102-110 us, 2 048: 95-104, 3 072: 93-107, 4 096: 91-105)
        const int wgs = g_debug_integrate_wgs > 0 ? g_debug_integrate_wgs : 4096;
        integrate_dense_strip_kernel<<<wgs, 256, 0, st>>>((uint4*)s->vba, v->depth, p, sz[0], sz[1], sz[2], of[0], of[1], of[2], cc, gc);
      }
      else if (columns && classify == 1) ITM_DENSE(true, 1, 1);
      else if (columns && classify == 2) ITM_DENSE(true, 1, 2);
      else if (columns && classify == 3) ITM_DENSE(true, 1, 3);
      else if (columns) ITM_DENSE(true, 1, 0);
      else if (pow2) ITM_DENSE(true, 0, 0);
      else ITM_DENSE(false, 0, 0);
#undef ITM_DENSE
    } else {
      tk.reset(new KernelTimer(s, ITM_TK_INTEGRATE, st));
      int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
        using VX = decltype(vx);
        integrate_dense_kernel<VX><<<256 * 32, 256, 0, st>>>(s->vba, v->depth, rgb, p, sz[0], sz[1], sz[2], of[0], of[1], of[2]);
        return ITM_OK;
      });
      if (rc) return rc;
    }
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

// Test hook (host only, no device work): the row interval the dense integration would use for the column of 4-voxel groups
// (x0 .. x0 + 3, slice z) of a volume `size` / `offset` seen from pose M_d -- so that the conservativeness of the cull can be
// checked against the exact per-voxel test for thousands of poses on a machine without a GPU.  Returns 1 when the planes could not be formed.
extern "C" int itm_debug_column_cull_rows(const float M_d[16], const float intr[4], int w, int h, float voxelSize, const int size[3], const int offset[3],
                                          int x0, int z, int* rlo, int* rhi) {
  if (!M_d || !intr || !size || !offset || !rlo || !rhi) return set_error(ITM_ERR_INVALID, "null argument");
  FuseParams p;
  memset(&p, 0, sizeof p);
  memcpy(p.M_d.m, M_d, 64);
  p.fx = intr[0]; p.fy = intr[1]; p.cx = intr[2]; p.cy = intr[3];
  p.W = w; p.H = h; p.voxelSize = voxelSize;
  ColumnCull cc;
  if (!make_column_cull(p, size, offset, cc)) return 1;
  column_rows(cc, x0, z, *rlo, *rhi);
  return ITM_OK;
}

// Test hook: the counters of the classification check (debug key 16 = 3): {free groups, shadow groups, mixed groups, violations};
// reset != 0 clears them.  A violation is a voxel of a classified group whose exact per-voxel outcome differs from its class.
extern "C" int itm_debug_dense_classify_check(int32_t out[4], int reset) {
  if (!out) return set_error(ITM_ERR_INVALID, "null argument");
  ITM_HIP(hipDeviceSynchronize());
  ITM_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_classifyCheck), 16));
  if (reset) { const int32_t z[4] = {0, 0, 0, 0}; ITM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_classifyCheck), z, 16)); }
  return ITM_OK;
}

