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

#include "itm_internal.h"
#include "range_device.h"

namespace itm {

#ifndef ITM_EXP_FUSED_STAMPS
#define ITM_EXP_FUSED_STAMPS 0
#endif
#ifndef ITM_PROJECTION_PRIORITY
#define ITM_PROJECTION_PRIORITY 1
#endif
int g_debug_integrate_wgs = 0;
int g_debug_no_fused_projection = 0;
int g_debug_dense_group_cull = 0;   // debug key 9: per-group frustum test instead of the per-column row interval

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
};

// Depth part, stage 1: project the voxel; returns the index of the depth pixel it falls on, or -1 when the voxel
// cannot be touched (behind the camera / outside the image).  Operation order as SURVEY.md Appendix A.5.
__device__ inline int fuse_depth_project(float mx, float my, float mz, const FuseParams& p, float& pcz) {
  Vec3 pc = transform_point(p.M_d, mx, my, mz);
  pcz = pc.z;
  if (pc.z <= 0) return -1;
  const float tx = p.fx * pc.x, ty = p.fy * pc.y;
#if ITM_FAST_DIVISIONS
  // Conservative early-out before the two divisions: the voxel is rejected below when
  // u = tx/z + cx is outside [1, W-2]; half a pixel of margin dwarfs every rounding error, so this
  // never rejects a voxel the exact test would keep (most voxels of a dense volume leave here).
  if (tx < (0.5f - p.cx) * pc.z || tx > ((float)(p.W - 2) + 0.5f - p.cx) * pc.z ||
      ty < (0.5f - p.cy) * pc.z || ty > ((float)(p.H - 2) + 0.5f - p.cy) * pc.z) return -1;
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
  return (int)(u + 0.5f) + (int)(v + 0.5f) * p.W;
}

// Stage 2: the running average with the measured depth dm of that pixel.  Returns eta (or -1 when the voxel is not
// touched); `touched` tells whether the register image changed.
template <class VX>
__device__ inline float fuse_depth_update(typename VX::Reg& r, float dm, float pcz, const FuseParams& p, bool& touched) {
  if (dm <= 0.0f) return -1;
  const float eta = dm - pcz;
  if (eta < -p.mu) return eta;
  const float oldF = VX::to_float(VX::raw_sdf(r));
  const int oldW = VX::w_depth(r);
#if ITM_FAST_DIVISIONS
  float newF = p.muFast ? div_markstein(eta, p.mu, p.rcpMu) : eta / p.mu;
#else
  float newF = eta / p.mu;
#endif
  newF = (1.0f < newF) ? 1.0f : newF;
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
  const uchar4 zero = make_uchar4(0, 0, 0, 0);
  const uchar4 A = rgb[px + py * p.Wc];
  uchar4 B = zero, C = zero, D = zero;
  if (dx != 0) B = rgb[(px + 1) + py * p.Wc];
  if (dy != 0) C = rgb[px + (py + 1) * p.Wc];
  if (dx != 0 && dy != 0) D = rgb[(px + 1) + (py + 1) * p.Wc];
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
    if (!((eta > p.mu) || (fabsf(eta / p.mu) > 0.25f))) {
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
#ifndef ITM_INTEGRATE_SLICES
#define ITM_INTEGRATE_SLICES 4
#endif
constexpr int kSlices = ITM_INTEGRATE_SLICES;
constexpr int kItemsPerBlock = kBlockSide / kSlices;
static_assert(kSlices == 1 || kSlices == 2 || kSlices == 4 || kSlices == 8, "slice groups tile the block");

#ifndef ITM_INTEGRATE_PREFETCH
// 1: the voxel runs of the wave's NEXT item are requested right behind the depth gathers of the current one.  Measured: BASELINE
// configs[4] 189 -> 198 us (the second register set costs a wave of occupancy, 76 -> 92 VGPRs), configs[1] +-0.  Off.
#define ITM_INTEGRATE_PREFETCH 0
#endif

template <class VX>
__device__ inline void load_item(const HashEntry& he, int z0, int lane, const void* __restrict__ vba, typename VX::Reg r[kSlices]) {
  const size_t vi = (size_t)(he.ptr < 0 ? 0 : he.ptr) * kBlockVoxels + (size_t)z0 * 64 + lane;      // block 0 is always there
#pragma unroll
  for (int k = 0; k < kSlices; ++k) r[k] = VX::load(vba, vi + 64 * k);
}

// r: the item's voxels (already requested).  hasNext / next / nextR: the voxel runs of the following item are requested right
// behind the depth gathers -- the wave then waits for those gathers only (the memory counter retires in issue order), and the new
// runs travel while this item is updated and stored.
template <class VX>
__device__ inline void integrate_item(const HashEntry& he, int z0, int lane, typename VX::Reg r[kSlices], void* __restrict__ vba, void* __restrict__ sdfMirror,
                                      const float* __restrict__ depth, const uchar4* __restrict__ rgb, const FuseParams& p,
                                      bool hasNext, const HashEntry& next, int nextZ0, typename VX::Reg nextR[kSlices]) {
  const bool present = he.ptr >= 0;
  const int x = lane & 7, y = lane >> 3;
  const size_t vi = (size_t)(present ? he.ptr : 0) * kBlockVoxels + (size_t)z0 * 64 + lane;
  const float mx = (float)(he.px * kBlockSide + x) * p.voxelSize;
  const float my = (float)(he.py * kBlockSide + y) * p.voxelSize;
  using MC = MirrorCodec<VX::kShort>;
  size_t mbase = 0;
  typename MC::T* mirror = nullptr;
  if (sdfMirror && mirror_index(he.px * kBlockSide, he.py * kBlockSide, he.pz * kBlockSide, mbase)) mirror = (typename MC::T*)sdfMirror;
  // stage 1: project every slice's voxel; stage 2: all depth pixels together; stage 3: update (+ colour), store what changed
  int pix[kSlices];
  float pcz[kSlices], mz[kSlices];
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    mz[k] = (float)(he.pz * kBlockSide + z0 + k) * p.voxelSize;
    pix[k] = -2;                                                            // -2: voxel skipped altogether (no block / stopIntegratingAtMaxW)
    if (!present || (p.stopAtMax && VX::w_depth(r[k]) == p.maxW)) continue;
    pix[k] = fuse_depth_project(mx, my, mz[k], p, pcz[k]);
  }
  float dm[kSlices];
#pragma unroll
  for (int k = 0; k < kSlices; ++k) dm[k] = depth[pix[k] >= 0 ? pix[k] : 0];
  if (hasNext) load_item<VX>(next, nextZ0, lane, vba, nextR);
#pragma unroll
  for (int k = 0; k < kSlices; ++k) {
    if (pix[k] == -2) continue;
    bool touched = false;
    const float eta = (pix[k] >= 0) ? fuse_depth_update<VX>(r[k], dm[k], pcz[k], p, touched) : -1.0f;
    if constexpr (VX::kColor) {
      if (!((eta > p.mu) || (fabsf(eta / p.mu) > 0.25f))) {
        fuse_colour<VX>(r[k], mx, my, mz[k], rgb, p);
        touched = true;
      }
    }
    if (touched) {
      VX::store(vba, vi + 64 * k, r[k]);
      if (mirror) mirror[mbase + (size_t)(z0 + k) * 64 + lane] = MC::of(VX::raw_sdf(r[k]));   // sdf mirror (itm_types.h)
    }
  }
}

template <class VX>
__device__ inline void integrate_hash_body(int wgIdx, int wgCount, const int32_t* __restrict__ visibleIds, RenderCounters* __restrict__ rc,
                                           const uint4* __restrict__ hash, void* __restrict__ vba, void* __restrict__ sdfMirror,
                                           const float* __restrict__ depth, const uchar4* __restrict__ rgb, const FuseParams& p) {
  const int nItems = rc->noVisibleEntries * kItemsPerBlock;
  const int lane = threadIdx.x & 63;
  const int waves = wgCount * (int)(blockDim.x >> 6);
  int i = __builtin_amdgcn_readfirstlane(wgIdx * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
  if (i >= nItems) return;
  HashEntry cur = unpack_entry(hash[visibleIds[i / kItemsPerBlock]]);
  typename VX::Reg r[kSlices], rn[kSlices];
  load_item<VX>(cur, (i % kItemsPerBlock) * kSlices, lane, vba, r);
  for (;;) {
    const int nxt = i + waves;
    const bool more = nxt < nItems;
    HashEntry ahead = cur;
    if (more) ahead = unpack_entry(hash[visibleIds[nxt / kItemsPerBlock]]);
    const int zn = (nxt % kItemsPerBlock) * kSlices;
#if ITM_INTEGRATE_PREFETCH
    integrate_item<VX>(cur, (i % kItemsPerBlock) * kSlices, lane, r, vba, sdfMirror, depth, rgb, p, more, ahead, zn, rn);
    if (!more) break;
#pragma unroll
    for (int k = 0; k < kSlices; ++k) r[k] = rn[k];
#else
    integrate_item<VX>(cur, (i % kItemsPerBlock) * kSlices, lane, r, vba, sdfMirror, depth, rgb, p, false, ahead, 0, rn);
    if (!more) break;
    load_item<VX>(ahead, zn, lane, vba, r);
#endif
    cur = ahead; i = nxt;
  }
}

template <class VX>
__global__ void __launch_bounds__(512) integrate_hash_kernel(const int32_t* __restrict__ visibleIds, RenderCounters* __restrict__ rc,
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
#if ITM_PROJECTION_PRIORITY
    __builtin_amdgcn_s_setprio(3);
#endif
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
template <bool POW2, int CULL>
__global__ void __launch_bounds__(256) integrate_dense_s_x4_kernel(uint4* __restrict__ vba, const float* __restrict__ depth, FuseParams p,
                                                                   int sx, int sy, int sz, int ox, int oy, int oz, int log2sx4, ColumnCull cc) {
  const int sx4 = sx >> 2;
  const int z = blockIdx.y;
  const int plane = sx4 * sy;                       // groups per z slice
  uint4* __restrict__ slice = vba + (size_t)z * plane;
  const float mz = (float)(z + oz) * p.voxelSize;
  const int stride = gridDim.x * 256;
  auto process = [&](int idx, uint4 q) {
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
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool touched = false;
      if (pix[k] >= 0) fuse_depth_update<VoxelS>(v[k], dm[k], pcz[k], p, touched);
      any |= touched;
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
    for (; idx + stride < plane; idx += 2 * stride) {
      const bool c0 = outside(idx), c1 = outside(idx + stride);
      uint4 q0, q1;
      if (!c0) q0 = slice[idx];
      if (!c1) q1 = slice[idx + stride];
      if (!c0) process(idx, q0);
      if (!c1) process(idx + stride, q1);
    }
    if (idx < plane && !outside(idx)) process(idx, slice[idx]);
    return;
  }
  for (; idx + stride < plane; idx += 2 * stride) {
    const bool c0 = culled(idx), c1 = culled(idx + stride);
    uint4 q0, q1;
    if (!c0) q0 = slice[idx];
    if (!c1) q1 = slice[idx + stride];
    if (!c0) process(idx, q0);
    if (!c1) process(idx + stride, q1);
  }
  if (idx < plane && !culled(idx)) process(idx, slice[idx]);
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
  const uchar4* rgb = (const uchar4*)v->rgb;
  const bool colour = (s->cfg.voxelType == ITM_VOXEL_S_RGB || s->cfg.voxelType == ITM_VOXEL_F_RGB);
  if (colour && (!rgb || v->w_rgb <= 0 || v->h_rgb <= 0)) return set_error(ITM_ERR_INVALID, "colour voxels need an rgb image");

  KernelTimer tk(s, ITM_TK_INTEGRATE, st);
  if (s->cfg.indexType == ITM_INDEX_HASH) {
    // 2048 workgroups of 8 waves: more waves than the chip holds at once (768-1024 workgroups), so that the dispatcher evens out what
    // the static striding does not, but few enough that most waves have work -- a workgroup without any still holds a slot for ~1 us
    // (measured, configs[4] / configs[1]: 768-1024 workgroups 189 / 21.3 us, 1536: 177 / 19.9, 2048: 175 / 20.3, 4096: 170 / 20.4,
    // where the last workgroup of configs[1] only STARTS after 15 us)
    const int grid = g_debug_integrate_wgs > 0 ? g_debug_integrate_wgs : 2048;
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
      else
        integrate_hash_kernel<VX><<<grid, 512, 0, st>>>(rs->visibleIds, rs->counters, s->hash, s->vba, s->sdfMirror, v->depth, rgb, p);
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
      if (splits < 1) splits = 1;
      const dim3 grid(splits, sz[2]);
      ColumnCull cc;
      const bool columns = pow2 && ((splits * 256) % sx4) == 0 && !g_debug_dense_group_cull && make_column_cull(p, sz, of, cc);
      if (columns) integrate_dense_s_x4_kernel<true, 1><<<grid, 256, 0, st>>>((uint4*)s->vba, v->depth, p, sz[0], sz[1], sz[2], of[0], of[1], of[2], lg, cc);
      else if (pow2) integrate_dense_s_x4_kernel<true, 0><<<grid, 256, 0, st>>>((uint4*)s->vba, v->depth, p, sz[0], sz[1], sz[2], of[0], of[1], of[2], lg, cc);
      else integrate_dense_s_x4_kernel<false, 0><<<grid, 256, 0, st>>>((uint4*)s->vba, v->depth, p, sz[0], sz[1], sz[2], of[0], of[1], of[2], lg, cc);
    } else {
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

extern "C" int itm_integrate_into_scene(itm_scene* s, const itm_view* v, itm_render_state* rs, itm_stream stream) {
  if (!s || !v || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (!v->depth) return set_error(ITM_ERR_INVALID, "null depth image");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  return launch_integrate(s, v, rs, as_stream(stream), false);
}
