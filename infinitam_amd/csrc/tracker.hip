// tracker.hip -- ICP depth tracker on the ICP maps this path writes.
//
// Reference behaviour:
//   filterSubsampleWithHoles (float)   DeviceAgnostic/ITMLowLevelEngine.h:26-47
//   computePerPointGH_Depth(_Ab)       DeviceAgnostic/ITMDepthTracker.h:8-106
//   interpolateBilinear_withHoles      DeviceAgnostic/ITMPixelUtils.h:41-71
//   ITMDepthTracker_CPU::ComputeGandH  DeviceSpecific/CPU/ITMDepthTracker_CPU.cpp:15-79
//   ITMDepthTracker::TrackCamera & co. Engine/ITMDepthTracker.cpp:79-200
//   ITMPose::SetParamsFromModelView / SetModelViewFromParams / Coerce   Objects/ITMPose.cpp:84-253,322-326
//   ORUtils::Cholesky                  ORUtils/Cholesky.h
//
// Device part: one lane per depth pixel computes its residual row (A, b) with the reference's float
// operations; the 1 + 6 + 21 sums and the valid count are reduced per wave with DPP row shifts, one partial per
// workgroup, written as a record of tagged granules; the records are added in one fixed order (kSegBlocks) in double
// precision wherever that happens (deterministic; the reference adds floats in raster order, so sums agree to float
// rounding, the count exactly).  Host part: a damped Gauss-Newton iteration over SE(3) with the reference's accept / reject schedule, written
// independently of the reference's code (double precision, LDL^T solve, exp/log projection: se3.h); the tracked pose
// agrees with the reference's to 2e-5 (tests/test_tracker.py).
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "itm_internal.h"
#include "icp_solver.h"
#include "se3.h"
#include "wave_utils.h"

namespace itm {

__global__ void __launch_bounds__(256) subsample_holes_kernel(const float* __restrict__ in, int wIn, float* __restrict__ out, int wOut, int hOut) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= wOut || y >= hOut) return;
  const int sx = x * 2, sy = y * 2;
  float acc = 0.0f, good = 0.0f, v;
  v = in[sx + sy * wIn];           if (v > 0.0f) { acc += v; good++; }
  v = in[(sx + 1) + sy * wIn];     if (v > 0.0f) { acc += v; good++; }
  v = in[sx + (sy + 1) * wIn];     if (v > 0.0f) { acc += v; good++; }
  v = in[(sx + 1) + (sy + 1) * wIn]; if (v > 0.0f) { acc += v; good++; }
  if (good > 0) acc /= good;
  out[x + y * wOut] = acc;
}

// The whole FilterSubsampleWithHoles pyramid in one launch: a level-l pixel depends on a 2^l x 2^l block of the input only, so a
// workgroup that owns a 16x16 input tile can produce every coarser pixel that lies under it (8x8, 4x4, 2x2, 1) through LDS --
// the same nested averages of valid values, in the same order, as four launches of subsample_holes_kernel (4 x ~8 us of host time
// per TrackCamera call).  Up to four coarser levels (the default hierarchy has exactly four).
struct PyramidLevels { float* out[4]; int w[5], h[5]; int levels; };   // w[0], h[0]: the input; levels = number of coarser levels (1..4)

__device__ inline float average_valid(float a, float b, float c, float d) {
  float acc = 0.0f, good = 0.0f;
  if (a > 0.0f) { acc += a; good++; }
  if (b > 0.0f) { acc += b; good++; }
  if (c > 0.0f) { acc += c; good++; }
  if (d > 0.0f) { acc += d; good++; }
  if (good > 0) acc /= good;
  return acc;
}

__global__ void __launch_bounds__(256) pyramid_kernel(const float* __restrict__ in, PyramidLevels L) {
  __shared__ float tile[2][16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int gx = blockIdx.x * 16 + tx, gy = blockIdx.y * 16 + ty;
  // pixels outside the input are never read by a pixel that exists on a coarser level (w[l+1] = w[l] / 2 rounds down)
  tile[0][ty][tx] = (gx < L.w[0] && gy < L.h[0]) ? in[gx + gy * L.w[0]] : 0.0f;
  __syncthreads();
  int side = 8, cur = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < L.levels) {
      if (tx < side && ty < side) {
        const float v = average_valid(tile[cur][2 * ty][2 * tx], tile[cur][2 * ty][2 * tx + 1], tile[cur][2 * ty + 1][2 * tx], tile[cur][2 * ty + 1][2 * tx + 1]);
        tile[cur ^ 1][ty][tx] = v;
        const int ox = blockIdx.x * side + tx, oy = blockIdx.y * side + ty;
        if (ox < L.w[l + 1] && oy < L.h[l + 1]) L.out[l][ox + oy * L.w[l + 1]] = v;
      }
      __syncthreads();
      cur ^= 1; side >>= 1;
    }
  }
}

struct GHParams {
  Mat4 approxInvPose, scenePose;
  float vfx, vfy, vcx, vcy;   // view intrinsics (fx, fy, cx, cy)
  float sfx, sfy, scx, scy;   // scene intrinsics
  float distThresh;
  int w, h, sceneW, sceneH;
  int tileH;                  // rows of a tile (gh_tiling)
};

constexpr int kGHValues = 1 + 6 + 21;   // f, nabla, packed lower-triangular hessian

// Everything the tracker hands between host and device (and between workgroups) travels as TAGGED GRANULES: 8 aligned bytes = 4 bytes
// of payload + the 4-byte sequence number of the evaluation they belong to.  An aligned 8-byte store is indivisible for the device,
// across PCIe and for the host, so a reader that finds the expected number in a granule holds that granule's payload: no stamp
// written after the values, and therefore no wait for the values to have left before the stamp may follow (one PCIe or memory
// round trip per hand-off, 1.5-2 us each, measured).
constexpr int kRecordWords = 2 * kGHValues + 1;   // 28 doubles as two words each, then the count
struct GHBlockRecord { unsigned long long g[64]; };   // one per workgroup (pinned host memory) and the session's result; granule i = tag << 32 | word i
static_assert(kRecordWords <= 64, "record granules");
__host__ __device__ inline unsigned int next_seq(unsigned int s) { ++s; return (s == 0u || s == 0xffffffffu) ? 1u : s; }   // 0 and ~0 are never sequence numbers

// ORDER OF THE ADDITIONS over the workgroups' records, the same wherever they are added (host: per-launch path and coarse levels
// of a session; device: the gathering workgroup of a session), so that every path yields the same bits: records in segments of
// kSegBlocks consecutive workgroups, each segment added up in block order, then the kSegs segment sums in segment order (an
// empty segment contributes +0.0).  A single chain over 240 records was 2.5 us of dependent additions in the gathering workgroup.
constexpr int kSegBlocks = 32;

// thread i < kGHValues holds value i of the workgroup (`mine`); every thread holds `cnt`.  Lane i of the first wave stores granule i:
// one store instruction over 512 contiguous bytes, which leaves the CU as whole 64-byte lines (granules written two per lane,
// 16 bytes apart, crossed PCIe one by one and cost the host's memory a partial-line update each: 7 us per record, measured)
template <int SCOPE>
__device__ inline void send_record(GHBlockRecord* r, double mine, int cnt, unsigned int tag) {
  if (threadIdx.x >= 64) return;
  const int i = threadIdx.x;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(__shfl(mine, i >> 1, 64));
  unsigned int w = (i & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
  if (i == 2 * kGHValues) w = (unsigned int)cnt;
  if (i > 2 * kGHValues) w = 0u;
  __hip_atomic_store(&r->g[i], ((unsigned long long)tag << 32) | w, __ATOMIC_RELAXED, SCOPE);
}

#ifndef ITM_EXP_TRACKER_TRACE
#define ITM_EXP_TRACKER_TRACE 0   // measurement build: per-evaluation host-side latencies on stderr
#endif
#if ITM_EXP_TRACKER_TRACE
#define ITM_TT(...) __VA_ARGS__
#else
#define ITM_TT(...)
#endif

// interpolateBilinear_withHoles for a Vector4f map; returns false when any tap is a hole (w < 0)
__device__ inline bool bilinear_holes(const float4* __restrict__ src, float px, float py, int W, float4& r) {
  const int ix = (int)(int16_t)(int)floorf(px), iy = (int)(int16_t)(int)floorf(py);   // (short)floor(...)
  const float dx = px - (float)ix, dy = py - (float)iy;
  const float4 a = src[ix + iy * W], b = src[(ix + 1) + iy * W], c = src[ix + (iy + 1) * W], d = src[(ix + 1) + (iy + 1) * W];
  if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return false;
  r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
  r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
  r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
  r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
  return true;
}

// MODE: 1 rotation only (3 parameters), 2 translation only (3), 3 both (6)
//
// At most kGHGroups workgroups: each walks its 16x16-pixel tiles (tile = group, group + G, ...) and keeps its sums in
// registers (double), so a call delivers at most kGHGroups records to the host whatever the image size -- with one record per
// tile, the 1 200 records (288 KB over PCIe, two system fences per workgroup) of a 640x480 level cost 113 us per evaluation
// against 20 us for the 40x30 level (tools/tracker_bench.py).
constexpr int kGHGroups = 256;
constexpr int kSegs = (kGHGroups + kSegBlocks - 1) / kSegBlocks;
constexpr int kGHWaves = 4;      // waves per workgroup (measured per 640x480 evaluation: 4 waves 43 us, 8 waves 51 us, 16 waves 73 us)
static_assert(kSegBlocks % kGHWaves == 0, "a gathering workgroup splits its segment evenly over its waves");
constexpr int kGHThreads = 64 * kGHWaves;
constexpr int kGHTileH = kGHThreads / 16;      // a tile is 16 pixels wide and at most kGHTileH tall
// Tiling of a w x h level: full tiles (one pixel per thread) where there are many; on the coarse levels, where a full tiling
// would occupy a handful of compute units and each would push its four waves' divergent map taps through one texture path,
// tiles of one or two waves' height -- more workgroups with one busy wave each -- as long as their count stays within `limit`
// (the number of per-workgroup records the host adds itself).  Returns the number of tiles.
__host__ __device__ inline int gh_tiling(int w, int h, int limit, int& tileH) {
  const int tilesX = (w + 15) / 16;
  for (tileH = 4; tileH < kGHTileH; tileH *= 2)
    if (tilesX * ((h + tileH - 1) / tileH) <= limit) break;
  return tilesX * ((h + tileH - 1) / tileH);
}

// One depth pixel of the evaluation in three stages, so that the loads of a stage -- of several pixels -- are in flight together
// instead of one dependent round trip after the other inside nested branches: (1) depth -> point in the scene frame -> position in
// the rendered maps; (2) the four taps of the point map and of the normal map, fetched unconditionally (tap 0 of the map when the
// pixel has dropped out); (3) the residual row.  The arithmetic is that of computePerPointGH_Depth_Ab, operation for operation.
struct GHPixel { bool live; float qx, qy, qz, u, v; };

__device__ inline GHPixel gh_project(const float* __restrict__ depth, const GHParams& p, int x, int y, bool inImage) {
  GHPixel r{false, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const float d = depth[inImage ? x + y * p.w : 0];
  if (!inImage || d <= 1e-8f) return r;
  // back-project, move to the scene frame, re-project into the rendered maps
  const float cx3 = d * (((float)x - p.vcx) / p.vfx), cy3 = d * (((float)y - p.vcy) / p.vfy);
  const Vec3 q = transform_point(p.approxInvPose, cx3, cy3, d);
  const Vec3 rp = transform_point(p.scenePose, q.x, q.y, q.z);
  if (rp.z <= 0.0f) return r;
  const float u = p.sfx * rp.x / rp.z + p.scx, v = p.sfy * rp.y / rp.z + p.scy;
  if (!((u >= 0.0f) && (u <= p.sceneW - 2) && (v >= 0.0f) && (v <= p.sceneH - 2))) return r;
  r.live = true; r.qx = q.x; r.qy = q.y; r.qz = q.z; r.u = u; r.v = v;
  return r;
}

struct GHTaps { float4 a, b, c, d; };
__device__ inline GHTaps gh_taps(const float4* __restrict__ src, const GHPixel& px, int W) {
  const int ix = (int)(int16_t)(int)floorf(px.u), iy = (int)(int16_t)(int)floorf(px.v);   // (short)floor(...)
  const int i0 = px.live ? ix + iy * W : 0, i1 = px.live ? (ix + 1) + iy * W : 0, i2 = px.live ? ix + (iy + 1) * W : 0, i3 = px.live ? (ix + 1) + (iy + 1) * W : 0;
  return GHTaps{src[i0], src[i1], src[i2], src[i3]};
}
// interpolateBilinear_withHoles on fetched taps; false when any tap is a hole (w < 0)
__device__ inline bool gh_blend(const GHTaps& t, float px, float py, float4& r) {
  const int ix = (int)(int16_t)(int)floorf(px), iy = (int)(int16_t)(int)floorf(py);
  const float dx = px - (float)ix, dy = py - (float)iy;
  if (t.a.w < 0 || t.b.w < 0 || t.c.w < 0 || t.d.w < 0) return false;
  r.x = (t.a.x * (1.0f - dx) * (1.0f - dy) + t.b.x * dx * (1.0f - dy) + t.c.x * (1.0f - dx) * dy + t.d.x * dx * dy);
  r.y = (t.a.y * (1.0f - dx) * (1.0f - dy) + t.b.y * dx * (1.0f - dy) + t.c.y * (1.0f - dx) * dy + t.d.y * dx * dy);
  r.z = (t.a.z * (1.0f - dx) * (1.0f - dy) + t.b.z * dx * (1.0f - dy) + t.c.z * (1.0f - dx) * dy + t.d.z * dx * dy);
  r.w = (t.a.w * (1.0f - dx) * (1.0f - dy) + t.b.w * dx * (1.0f - dy) + t.c.w * (1.0f - dx) * dy + t.d.w * dx * dy);
  return true;
}

template <int MODE>
__device__ inline void gh_row(const GHPixel& px, const GHTaps& tp, const GHTaps& tn, const GHParams& p, double acc[kGHValues], int& valid) {
  constexpr int NP = (MODE == 3) ? 6 : 3;
  constexpr int NH = NP * (NP + 1) / 2;
  float vals[kGHValues];
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) vals[i] = 0.0f;
  float4 cp;
  if (px.live && gh_blend(tp, px.u, px.v, cp) && !(cp.w < 0.0f)) {
    const float ex = cp.x - px.qx, ey = cp.y - px.qy, ez = cp.z - px.qz;
    const float dist = ex * ex + ey * ey + ez * ez;
    if (!(dist > p.distThresh)) {
      // a hole in the normals map yields the zero normal but still counts (the reference's check is commented out)
      float4 n;
      if (!gh_blend(tn, px.u, px.v, n)) n = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
      const float b = n.x * ex + n.y * ey + n.z * ez;
      float A[NP];
      if (MODE == 2) { A[0] = n.x; A[1] = n.y; A[2] = n.z; }
      else {
        A[0] = +px.qz * n.y - px.qy * n.z;
        A[1] = -px.qz * n.x + px.qx * n.z;
        A[2] = +px.qy * n.x - px.qx * n.y;
        if constexpr (MODE == 3) { A[3] = n.x; A[4] = n.y; A[5] = n.z; }
      }
      vals[0] = b * b;
      int k = 0;
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        vals[1 + r] = b * A[r];
#pragma unroll
        for (int c = 0; c <= r; ++c, ++k) vals[7 + k] = A[r] * A[c];
      }
      ++valid;
    }
  }
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) {
    const bool used = (i == 0) || (i >= 1 && i < 1 + NP) || (i >= 7 && i < 7 + NH);
    if (used) acc[i] += (double)vals[i];
  }
}

// The tiles blk, blk + nBlocks, ... of one workgroup, two at a time: every lane adds the residual rows of its pixels to `acc`
// (double, in tile order) and counts them
template <int MODE>
__device__ inline void gh_accumulate(const float* __restrict__ depth, const float4* __restrict__ pointsMap, const float4* __restrict__ normalsMap,
                                     const GHParams& p, int blk, int nBlocks, double acc[kGHValues], int& valid ITM_TT(, unsigned long long* tt = nullptr)) {
  const int tilesX = (p.w + 15) / 16, tiles = tilesX * ((p.h + p.tileH - 1) / p.tileH);
  const int row = threadIdx.x >> 4;
  if (row >= p.tileH) return;                   // a short tile: the waves below it have no pixels (whole waves: tileH is a multiple of 4)
  for (int tile = blk; tile < tiles; tile += 2 * nBlocks) {
    const int tileB = tile + nBlocks;
    const int xA = (tile % tilesX) * 16 + (threadIdx.x & 15), yA = (tile / tilesX) * p.tileH + row;
    const int xB = (tileB % tilesX) * 16 + (threadIdx.x & 15), yB = (tileB / tilesX) * p.tileH + row;
    const GHPixel pa = gh_project(depth, p, xA, yA, xA < p.w && yA < p.h);
    const GHPixel pb = gh_project(depth, p, xB, yB, tileB < tiles && xB < p.w && yB < p.h);
    ITM_TT(if (tt && tile == blk) { __builtin_amdgcn_s_waitcnt(0); tt[0] = __builtin_amdgcn_s_memrealtime(); })
    const GHTaps ta = gh_taps(pointsMap, pa, p.sceneW), na = gh_taps(normalsMap, pa, p.sceneW);
    const GHTaps tb = gh_taps(pointsMap, pb, p.sceneW), nb = gh_taps(normalsMap, pb, p.sceneW);
    ITM_TT(if (tt && tile == blk) { __builtin_amdgcn_s_waitcnt(0); tt[1] = __builtin_amdgcn_s_memrealtime(); })
    gh_row<MODE>(pa, ta, na, p, acc, valid);
    gh_row<MODE>(pb, tb, nb, p, acc, valid);
  }
}

// Sum of `s` over the 64 lanes of the wave, the same value in every lane: an inclusive scan inside each row of 16 lanes with DPP
// row shifts (lanes that would read from outside their row add 0), then the four row totals in row order.  Data-parallel
// primitives keep the exchange in the ALU; the ds_bpermute butterfly this replaces was one LDS round trip per step and value
// (2.7 us per evaluation for the block reduction, measured).  The order of the additions is fixed: deterministic sums.
template <int CTRL>
__device__ inline double dpp_shifted(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo));
}
__device__ inline double lane_value(double v, int lane) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane), hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ inline double wave_sum(double s) {
  s += dpp_shifted<0x111>(s);   // row_shr:1
  s += dpp_shifted<0x112>(s);   // row_shr:2
  s += dpp_shifted<0x114>(s);   // row_shr:4
  s += dpp_shifted<0x118>(s);   // row_shr:8  -> lane 15 of every row holds the row's sum
  return ((lane_value(s, 15) + lane_value(s, 31)) + lane_value(s, 47)) + lane_value(s, 63);
}

// wave sums in double (fixed order), then the waves in order: thread i < kGHValues ends up with value i of the workgroup, every
// thread with its count
template <int MODE>
__device__ inline void gh_block_reduce(const double acc[kGHValues], int valid, double (*lds)[kGHValues], int* ldsCount, double& mine, int& cnt) {
  constexpr int NP = (MODE == 3) ? 6 : 3;
  constexpr int NH = NP * (NP + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) {
    const bool used = (i == 0) || (i >= 1 && i < 1 + NP) || (i >= 7 && i < 7 + NH);
    const double s = used ? wave_sum(acc[i]) : 0.0;
    if (lane == 0) lds[wave][i] = s;
  }
  const int total = wave_reduce_sum(valid);
  if (lane == 0) ldsCount[wave] = total;
  __syncthreads();
  mine = 0.0; cnt = 0;
#pragma unroll
  for (int wv = 0; wv < kGHWaves; ++wv) {          // fixed order: deterministic
    if (threadIdx.x < kGHValues) mine += lds[wv][threadIdx.x];
    cnt += ldsCount[wv];
  }
}

template <int MODE>
__global__ void __launch_bounds__(kGHThreads) gh_partial_kernel(const float* __restrict__ depth, const float4* __restrict__ pointsMap,
                                                        const float4* __restrict__ normalsMap, GHParams p, GHBlockRecord* __restrict__ hostRec, unsigned int seq) {
  __shared__ double lds[kGHWaves][kGHValues];
  __shared__ int ldsCount[kGHWaves];
  double acc[kGHValues];
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) acc[i] = 0.0;
  int valid = 0;
  gh_accumulate<MODE>(depth, pointsMap, normalsMap, p, blockIdx.x, gridDim.x, acc, valid);
  const int blk = blockIdx.x;
  double mine; int cnt;
  gh_block_reduce<MODE>(acc, valid, lds, ldsCount, mine, cnt);
  // the partial goes to a tagged record in pinned host memory: the host adds the records itself (one launch per
  // Levenberg-Marquardt iteration, no reduction launch, no copy)
  send_record<__HIP_MEMORY_SCOPE_SYSTEM>(hostRec + blk, mine, cnt, seq);
}

// ---- evaluation session: ONE launch serves every evaluation of a TrackCamera call -------------------------------------------
// TrackCamera evaluates cost / gradient / Hessian 5-12 times, each at a pose the host derives from the previous answer.  With a
// launch per evaluation the floor was ~25-30 us each (launch, kernel, up to 256 records over PCIe, the host's summation).  Here
// the workgroups stay resident for the duration of the call: the host writes a command (pose, level, mode) as tagged granules,
// the first wave of every workgroup polls them -- the moment all carry the expected number the workgroup has the whole command --,
// every workgroup accumulates its tiles, and answers with a tagged record: on coarse levels straight to pinned host memory (the host
// adds the records in the common order, kSegBlocks), on fine ones to device memory, where workgroup g < 8 collects the records of
// segment g, adds them in block order and sends the segment's sums to the host, which adds the segments -- the same order again, so
// the sums are bit-identical to the per-launch path.
// The kernel cannot outlive its host: a workgroup leaves when no command has arrived for kSessionIdleTicks (2 ms on the 100 MHz
// clock; workgroup 0, which reports the exit; the others have a longer limit) or when the host says so; a host that finds the
// session gone starts another one at the pending sequence number (same stream: it begins when the old one has left entirely).
constexpr unsigned int kSessionExit = 0xffffffffu;
constexpr unsigned long long kSessionIdleTicks = 200000ull;   // 2 ms
constexpr int kSessionPollSleep = 4;       // s_sleep argument between two polls of the command granules (x64 clocks)
constexpr int kSessionToHostBlocks = 96;   // evaluations with at most this many workgroups answer with per-workgroup records
struct GHCommand {            // the payload words of the command granules
  GHParams p;
  const float* depth; const float4* points; const float4* normals;
  int mode, activeBlocks;
  int toHost;                 // few workgroups: each sends its record straight to pinned host memory, the host adds them
  unsigned int session;       // granule kCommandWords - 1; with the tag kSessionExit it tells session `session` to leave
};
constexpr int kCommandWords = (int)(sizeof(GHCommand) / 4);
static_assert(sizeof(GHCommand) % 4 == 0 && kCommandWords <= 64, "one granule per lane of the polling wave");
constexpr int kExitGranule = kCommandWords - 1;
struct GHResult { GHBlockRecord segment[kSegs]; volatile unsigned int exited; unsigned long long stamps[6]; };   // device -> host (stamps: measurement builds)

__global__ void __launch_bounds__(kGHThreads) gh_session_kernel(const unsigned long long* __restrict__ hostCmd, unsigned long long* __restrict__ devCmd,
                                                                GHBlockRecord* __restrict__ devRec, GHResult* __restrict__ hostRes, GHBlockRecord* __restrict__ hostRec,
                                                                unsigned int session, unsigned int firstSeq, int direct) {
  __shared__ __attribute__((aligned(16))) unsigned int cmdWords[kCommandWords];
  __shared__ unsigned int nextSeq;
  __shared__ int gatherFailed;
  __shared__ double lds[kGHWaves][kGHValues];
  __shared__ int ldsCount[kGHWaves];
  __shared__ double gathered[kSegBlocks][kGHValues + 1];          // a gathering workgroup's copy of its segment's records (+ counts)
  const GHCommand& cmd = *(const GHCommand*)cmdWords;
  unsigned int expected = firstSeq;
  unsigned long long idleSince = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    // ---- wait for the next command: lane i of the first wave watches granule i ----
    // `direct`: the granules lie in fine-grained DEVICE memory that the host writes through the PCIe BAR and every workgroup reads
    // there; otherwise they lie in pinned HOST memory, workgroup 0 fetches them over PCIe and republishes them in device memory.
    // Everything that crosses workgroups or the PCIe link travels in relaxed atomic accesses of agent / system scope, which bypass
    // the non-coherent caches one granule at a time.  Agent- or system-scope FENCES would write back / invalidate the whole L2 of
    // the XCD on every evaluation, and the depth and map tiles the evaluations re-read live there (measured: 40 us per evaluation).
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      const bool fromHost = blockIdx.x == 0 || direct;
      const unsigned long long limit = (blockIdx.x == 0) ? kSessionIdleTicks : 4 * kSessionIdleTicks;
      unsigned int s;
      for (;;) {
        unsigned long long v = 0;
        if (lane < kCommandWords)
          v = fromHost ? __hip_atomic_load(hostCmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(devCmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int tag = (unsigned int)(v >> 32);
        const unsigned int tag0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)tag);
        // the last granule says whom the block is for: its payload is a session number, under a command's tag or under the exit tag
        const unsigned int lastTag = (unsigned int)__builtin_amdgcn_readlane((int)tag, kExitGranule);
        const unsigned int forSession = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, kExitGranule);
        // A command is taken when every granule carries the same number and that number is not behind the expected one.  It may be
        // AHEAD: only the workgroups with tiles answer an evaluation, so the host can move on while a workgroup without tiles has not
        // looked yet (on a GPU that other streams keep busy a poll can take longer than a coarse evaluation) -- it then simply joins
        // at the command it finds.
        if (__all(lane >= kCommandWords || tag == tag0) && tag0 != 0u && tag0 != kSessionExit && (int)(tag0 - expected) >= 0 && forSession == session) {
          if (lane < kCommandWords) {
            cmdWords[lane] = (unsigned int)v;
            if (!direct && blockIdx.x == 0) __hip_atomic_store(devCmd + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          s = tag0;
          break;
        }
        // Leave when told to (exit tag with this session's number), when the block is addressed to ANOTHER session (this one is
        // over: its exit was overwritten before this workgroup looked, or it has been replaced after an idle exit), or after the idle limit
        const bool told = lastTag == kSessionExit && forSession == session;
        const bool replaced = lastTag != 0u && lastTag != kSessionExit && forSession != session;
        if (told || replaced || (__builtin_amdgcn_s_memrealtime() - idleSince > limit)) {
          if (!direct && blockIdx.x == 0 && lane == kExitGranule) __hip_atomic_store(devCmd + lane, ((unsigned long long)kSessionExit << 32) | session, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s = kSessionExit;
          break;
        }
        __builtin_amdgcn_s_sleep(kSessionPollSleep);
      }
      if (lane == 0) { nextSeq = s; gatherFailed = 0; }
    }
    __syncthreads();
    const unsigned int s = nextSeq;
    if (s == kSessionExit) break;
    ITM_TT(const unsigned long long ttSeen = __builtin_amdgcn_s_memrealtime(); unsigned long long ttAcc = 0, ttRed = 0; unsigned long long ttIn[2] = {0, 0};)
    // ---- this workgroup's share ----
    const int nBlocks = cmd.activeBlocks;
    double mine = 0.0; int cnt = 0;
    if ((int)blockIdx.x < nBlocks) {
      double acc[kGHValues];
#pragma unroll
      for (int i = 0; i < kGHValues; ++i) acc[i] = 0.0;
      int valid = 0;
      if (cmd.mode == 1) { gh_accumulate<1>(cmd.depth, cmd.points, cmd.normals, cmd.p, blockIdx.x, nBlocks, acc, valid ITM_TT(, ttIn)); ITM_TT(ttAcc = __builtin_amdgcn_s_memrealtime();) gh_block_reduce<1>(acc, valid, lds, ldsCount, mine, cnt); }
      else if (cmd.mode == 2) { gh_accumulate<2>(cmd.depth, cmd.points, cmd.normals, cmd.p, blockIdx.x, nBlocks, acc, valid ITM_TT(, ttIn)); ITM_TT(ttAcc = __builtin_amdgcn_s_memrealtime();) gh_block_reduce<2>(acc, valid, lds, ldsCount, mine, cnt); }
      else { gh_accumulate<3>(cmd.depth, cmd.points, cmd.normals, cmd.p, blockIdx.x, nBlocks, acc, valid ITM_TT(, ttIn)); ITM_TT(ttAcc = __builtin_amdgcn_s_memrealtime();) gh_block_reduce<3>(acc, valid, lds, ldsCount, mine, cnt); }
      ITM_TT(ttRed = __builtin_amdgcn_s_memrealtime();)
      if (cmd.toHost) {
        // a coarse level: the record goes to the host as it is; no arrival counter, no gathering workgroup, no second trip
        // through device memory
        send_record<__HIP_MEMORY_SCOPE_SYSTEM>(hostRec + blockIdx.x, mine, cnt, s);
        ITM_TT(if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                 __hip_atomic_store(&hostRes->stamps[0], ttSeen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[1], ttAcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                 __hip_atomic_store(&hostRes->stamps[2], ttRed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[3], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                 __hip_atomic_store(&hostRes->stamps[4], ttIn[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[5], ttIn[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); })
      } else {
        send_record<__HIP_MEMORY_SCOPE_AGENT>(devRec + blockIdx.x, mine, cnt, s);
      }
    }
    const int segments = (nBlocks + kSegBlocks - 1) / kSegBlocks;
    if (!cmd.toHost && (int)blockIdx.x < segments) {
      // ---- a fine level: workgroup g < segments collects the tagged records of segment g from device memory (no arrival counter,
      // nobody waits for a store to have completed), adds them in block order and sends the segment's sums to the host, which adds
      // the segments: the common order.  (One workgroup collecting all 240 records: 7-8 us, bound by the uncached loads it can keep in flight.)
      ITM_TT(const unsigned long long ttLast = __builtin_amdgcn_s_memrealtime();)
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      constexpr int kPerWave = kSegBlocks / kGHWaves;
      const int base = blockIdx.x * kSegBlocks, count = min(kSegBlocks, nBlocks - base);
      // lane i fetches granule i: one record = one 512-byte load; the wave's records in flight together, re-fetched until all are there
      unsigned long long g[kPerWave];
      for (;;) {
        bool mineOk = true;
#pragma unroll
        for (int k = 0; k < kPerWave; ++k) {
          const int local = wave * kPerWave + k;
          g[k] = (local < count) ? __hip_atomic_load(&devRec[base + local].g[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)s << 32);
        }
#pragma unroll
        for (int k = 0; k < kPerWave; ++k) mineOk = mineOk && (unsigned int)(g[k] >> 32) == s;
        if (__all(mineOk)) break;
        if (__builtin_amdgcn_s_memrealtime() - idleSince > 4 * kSessionIdleTicks) { gatherFailed = 1; break; }   // a workgroup is gone: no answer, the host's time-out ends the call
      }
#pragma unroll
      for (int k = 0; k < kPerWave; ++k) {
        const int local = wave * kPerWave + k;
        const unsigned int lo = (unsigned int)g[k], hi = (unsigned int)__shfl_down((int)lo, 1, 64);
        if (local < count) {
          if (lane < 2 * kGHValues && !(lane & 1)) gathered[local][lane >> 1] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
          if (lane == 2 * kGHValues) gathered[local][kGHValues] = (double)(int)lo;          // counts are small integers: exact in double
        }
      }
      __syncthreads();
      const bool complete = gatherFailed == 0;
      double sum = 0.0;
      if (complete && threadIdx.x <= kGHValues) {
        int b = 0;
        for (; b + 8 <= count; b += 8) {            // eight LDS reads in flight per step, the additions stay a chain
          double v8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v8[k] = gathered[b + k][threadIdx.x];
#pragma unroll
          for (int k = 0; k < 8; ++k) sum += v8[k];
        }
        for (; b < count; ++b) sum += gathered[b][threadIdx.x];
      }
      const int total = (int)__shfl(sum, kGHValues, 64);
      if (complete) send_record<__HIP_MEMORY_SCOPE_SYSTEM>(&hostRes->segment[blockIdx.x], sum, total, s);
      // (workgroup 0's own times: command seen, tiles accumulated, gathering begun, segment sent)
      ITM_TT(if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime();
               __hip_atomic_store(&hostRes->stamps[0], ttSeen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[1], ttAcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
               __hip_atomic_store(&hostRes->stamps[2], ttLast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[3], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
               __hip_atomic_store(&hostRes->stamps[4], ttIn[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&hostRes->stamps[5], ttIn[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); })
    }
    __syncthreads();          // the next command overwrites the command words in LDS
    expected = next_seq(s);
    idleSince = __builtin_amdgcn_s_memrealtime();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store((unsigned int*)&hostRes->exited, session, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- tracker object -------------------------------------------------------------------------------------
// What ITMDepthTracker owns in the reference (Engine/ITMDepthTracker.cpp:18-44: the hierarchy and the device-side reduction
// buffers of ITMDepthTracker_CUDA) lives in a handle here: the workgroups' records (device memory and pinned host memory), the command block, the
// depth pyramid.  One handle = one tracker = one caller at a time (the handle's mutex serialises callers); the handle-less
// entry points of round 1 use a handle private to the calling thread, so two host threads (or two streams driven by two
// threads) never share records or sequence numbers.
}  // namespace itm

struct itm_tracker {
  std::mutex mu;
  int device = -1;
  itm::GHBlockRecord* devRec = nullptr;       // device memory: the workgroups' records of a session's fine levels
  itm::GHBlockRecord* rec = nullptr; itm::GHBlockRecord* recDev = nullptr;   // pinned host records + their device address
  size_t blocks = 0;
  unsigned int seq = 0;
  std::vector<float*> pyramid; std::vector<size_t> pyramidBytes;
  double pollTimeoutSeconds = 5.0;
  double sessionTimeoutSeconds = 0.5;     // an evaluation through the resident kernel answers in microseconds; after this long it goes through a launch
  int sessionUsable = -1;                 // -1 not probed yet, 0 launch per evaluation, 1 resident evaluation kernel
  int sessionFallbacks = 0;
  int debugEvaluations = 0;
  // evaluation session (gh_session_kernel)
  unsigned long long* cmd = nullptr; unsigned long long* cmdDev = nullptr;   // command granules (BAR-mapped device memory or pinned host memory) + their device address
  itm::GHResult* res = nullptr; itm::GHResult* resDev = nullptr;     // pinned result + its device address
  unsigned long long* devCmd = nullptr;   // device memory: republished command granules
  unsigned int session = 0;
  bool sessionOpen = false;
  bool cmdDirect = false;        // the command block is device memory the host writes through the BAR
};

namespace itm {

static void write_exit(itm_tracker* t) {       // tells session t->session to leave: the exit tag with the session's number as payload
  __atomic_thread_fence(__ATOMIC_RELEASE);
  *(volatile unsigned long long*)&t->cmd[kExitGranule] = ((unsigned long long)kSessionExit << 32) | t->session;
  __builtin_ia32_sfence();
}

static void tracker_release(itm_tracker* t) {
  if (t->cmd && t->sessionOpen) {                 // a call that failed half way: tell its resident kernel to leave before the buffers go
    write_exit(t);
    t->sessionOpen = false;
  }
  (void)hipFree(t->devRec);
  if (t->rec) (void)hipHostFree(t->rec);
  if (t->cmd) { if (t->cmdDirect) (void)hipFree(t->cmd); else (void)hipHostFree(t->cmd); }
  if (t->res) (void)hipHostFree(t->res);
  (void)hipFree(t->devCmd);
  t->cmd = nullptr; t->cmdDev = nullptr; t->res = nullptr; t->resDev = nullptr; t->devCmd = nullptr;
  t->sessionOpen = false;
  for (float* q : t->pyramid) (void)hipFree(q);
  t->devRec = nullptr; t->rec = nullptr; t->recDev = nullptr; t->blocks = 0;
  t->pyramid.clear(); t->pyramidBytes.clear();
}

// (re)sizes the reduction buffers for `blocks` workgroups on the current device; on failure the handle is left empty
static int tracker_reserve(itm_tracker* t, size_t blocks) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (t->device == dev && t->blocks >= blocks) return ITM_OK;
  if (t->device != dev) tracker_release(t);           // buffers of another device (incl. the pyramid) are of no use here
  else {
    (void)hipFree(t->devRec);
    if (t->rec) (void)hipHostFree(t->rec);
    t->devRec = nullptr; t->rec = nullptr; t->recDev = nullptr;
  }
  t->blocks = 0; t->device = dev;
  hipError_t e = hipMalloc((void**)&t->devRec, blocks * sizeof(GHBlockRecord));
  if (e == hipSuccess) e = hipMemset(t->devRec, 0, blocks * sizeof(GHBlockRecord));        // tag 0: no record
  // coherent + mapped: device stores become visible to the polling host without a kernel boundary
  if (e == hipSuccess) e = hipHostMalloc((void**)&t->rec, blocks * sizeof(GHBlockRecord), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { memset(t->rec, 0, blocks * sizeof(GHBlockRecord)); e = hipHostGetDevicePointer((void**)&t->recDev, t->rec, 0); }
  if (e != hipSuccess) {
    (void)hipFree(t->devRec);
    if (t->rec) (void)hipHostFree(t->rec);
    t->devRec = nullptr; t->rec = nullptr; t->recDev = nullptr;
    return hip_fail(e, "tracker buffers", __FILE__, __LINE__);
  }
  t->blocks = blocks;
  return ITM_OK;
}

// the calling thread's own tracker for the handle-less entry points (never destroyed: HIP may already be gone at thread exit)
static itm_tracker* thread_tracker() {
  static thread_local itm_tracker* mine = nullptr;
  if (!mine) mine = new (std::nothrow) itm_tracker();
  return mine;
}

// Adds one tagged record to `sums` / `count` once every granule carries `tag`.  `slow(granule)` is called every 1024 polls of a
// granule that has not arrived and decides whether the wait goes on (ITM_OK) or ends with an error code.
// the common order of the additions (kSegBlocks above) on the host
struct OrderedSums {
  double seg[kSegs][kGHValues];
  OrderedSums() { for (int g = 0; g < kSegs; ++g) for (int i = 0; i < kGHValues; ++i) seg[g][i] = 0.0; }
  double* of_block(size_t b) { return seg[b / kSegBlocks]; }
  void total(double out[kGHValues]) const {
    for (int i = 0; i < kGHValues; ++i) {
      double s = 0.0;
      for (int g = 0; g < kSegs; ++g) s += seg[g][i];
      out[i] = s;
    }
  }
};

template <class Slow>
static inline int read_record(const GHBlockRecord* r, unsigned int tag, double* sums, int* count, Slow&& slow) {
  unsigned int w[kRecordWords];
  for (int i = 0; i < kRecordWords; ++i) {
    const volatile unsigned long long* g = &r->g[i];
    unsigned long long v;
    unsigned spins = 0;
    while ((unsigned int)((v = *g) >> 32) != tag) {
      __builtin_ia32_pause();
      if ((++spins & 0x3ffu) != 0u) continue;
      const int rc = slow(g);
      if (rc) return rc;
    }
    w[i] = (unsigned int)v;
  }
  for (int i = 0; i < kGHValues; ++i) {
    const unsigned long long bits = (unsigned long long)w[2 * i] | ((unsigned long long)w[2 * i + 1] << 32);
    double d;
    memcpy(&d, &bits, 8);
    sums[i] += d;
  }
  *count += (int)w[2 * kGHValues];
  return ITM_OK;
}

static int compute_g_and_h(itm_tracker* trk, const float* depth, int w, int h, const float* viewIntr, const float* pointsMap, const float* normalsMap,
                           int sceneW, int sceneH, const float* sceneIntr, const float* approxInvPose, const float* scenePose,
                           float distThresh, int iterationType, itm_tracker_gh* out, hipStream_t st) {
  memset(out, 0, sizeof *out);
  if (iterationType == ITM_TRACKER_ITERATION_NONE) return ITM_OK;
  if (iterationType < 1 || iterationType > 3) return set_error(ITM_ERR_INVALID, "bad iteration type");
  int tileH;
  const int tiles = gh_tiling(w, h, kSessionToHostBlocks, tileH);
  const int rounds = (tiles + kGHGroups - 1) / kGHGroups;                  // tiles per workgroup, then as few workgroups as that needs
  const dim3 grid((tiles + rounds - 1) / rounds);
  const size_t blocks = grid.x;
  int rc = tracker_reserve(trk, blocks);
  if (rc) return rc;
  GHParams p;
  memcpy(p.approxInvPose.m, approxInvPose, 64); memcpy(p.scenePose.m, scenePose, 64);
  p.vfx = viewIntr[0]; p.vfy = viewIntr[1]; p.vcx = viewIntr[2]; p.vcy = viewIntr[3];
  p.sfx = sceneIntr[0]; p.sfy = sceneIntr[1]; p.scx = sceneIntr[2]; p.scy = sceneIntr[3];
  p.distThresh = distThresh; p.w = w; p.h = h; p.sceneW = sceneW; p.sceneH = sceneH; p.tileH = tileH;
  const float4* pm = (const float4*)pointsMap; const float4* nm = (const float4*)normalsMap;
  const int np = (iterationType == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
  const unsigned int seq = trk->seq = next_seq(trk->seq);
  if (iterationType == 1) gh_partial_kernel<1><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, p, trk->recDev, seq);
  else if (iterationType == 2) gh_partial_kernel<2><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, p, trk->recDev, seq);
  else gh_partial_kernel<3><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, p, trk->recDev, seq);
  ITM_LAUNCH_CHECK();
  // Wait for every workgroup's tagged record and add them in block order (fixed order => deterministic, in double).  The
  // poll is bounded in TIME: after 20 ms without a granule the stream is queried between polls -- a drained stream without
  // it, a device error, or pollTimeoutSeconds without progress end the call with ITM_ERR_DEVICE instead of stalling the
  // host on a kernel that will never finish.
  OrderedSums ordered;
  int n = 0;
  using clock = std::chrono::steady_clock;
  clock::time_point t0; bool timing = false;
  for (size_t b = 0; b < blocks; ++b) {
    rc = read_record(trk->rec + b, seq, ordered.of_block(b), &n, [&](const volatile unsigned long long* g) -> int {
      if (!timing) { t0 = clock::now(); timing = true; return ITM_OK; }
      const double waited = std::chrono::duration<double>(clock::now() - t0).count();
      if (waited < 0.02) return ITM_OK;
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) {
        if ((unsigned int)(*g >> 32) == seq) return ITM_OK;
        return set_error(ITM_ERR_DEVICE, "tracker reduction: the stream drained without delivering every record");
      }
      if (q != hipErrorNotReady) return hip_fail(q, "tracker reduction", __FILE__, __LINE__);
      if (waited > trk->pollTimeoutSeconds) return set_error(ITM_ERR_DEVICE, "tracker reduction timed out");
      return ITM_OK;
    });
    if (rc) return rc;
  }
  double sums[kGHValues];
  ordered.total(sums);
  for (int r = 0, k = 0; r < np; ++r)
    for (int c = 0; c <= r; ++c, ++k) out->hessian[r + c * 6] = (float)sums[7 + k];
  for (int r = 0; r < np; ++r)
    for (int c = r + 1; c < np; ++c) out->hessian[r + c * 6] = out->hessian[c + r * 6];
  for (int r = 0; r < np; ++r) out->nabla[r] = (float)sums[1 + r];
  out->noValidPoints = n;
  out->f = (n > 100) ? std::sqrt((float)sums[0]) / n : 1e5f;
  return ITM_OK;
}

// ---- evaluation session, host side -------------------------------------------------------------------------
int g_debug_tracker_launch_per_evaluation = 0;   // debug key 10: TrackCamera with one launch per evaluation (the path before the session kernel)
int g_debug_tracker_session_unusable = 0;        // debug key 18: the resident kernel reports itself unusable at the n-th evaluation of a call (tests the fall-back)
int g_debug_tracker_host_command = 0;            // debug key 11: session commands through pinned host memory even where the device has a large BAR

constexpr int kSessionUnusable = -1000;     // internal: the resident kernel cannot serve this evaluation, use a launch instead

static int session_reserve(itm_tracker* t) {
  int rc = tracker_reserve(t, kGHGroups);
  if (rc) return rc;
  if (t->cmd) return ITM_OK;
  // the command granules: fine-grained device memory written by the host through the PCIe BAR where the device has a large BAR (every
  // workgroup then polls local memory), pinned host memory fetched by workgroup 0 otherwise.  All zero = no command (0 is never a tag).
  const size_t cmdBytes = 64 * sizeof(unsigned long long);
  hipError_t e = hipSuccess;
  int dev = 0, largeBar = 0;
  (void)hipGetDevice(&dev);
  if (!g_debug_tracker_host_command && hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && largeBar &&
      hipExtMallocWithFlags((void**)&t->cmd, cmdBytes, hipDeviceMallocFinegrained) == hipSuccess) {
    t->cmdDirect = true;
    t->cmdDev = t->cmd;
    e = hipMemset(t->cmd, 0, cmdBytes);
  } else {
    (void)hipGetLastError();
    t->cmdDirect = false;
    e = hipHostMalloc((void**)&t->cmd, cmdBytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { memset(t->cmd, 0, cmdBytes); e = hipHostGetDevicePointer((void**)&t->cmdDev, t->cmd, 0); }
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&t->res, sizeof(GHResult), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { memset(t->res, 0, sizeof(GHResult)); e = hipHostGetDevicePointer((void**)&t->resDev, t->res, 0); }
  if (e == hipSuccess) e = hipMalloc((void**)&t->devCmd, cmdBytes);
  if (e == hipSuccess) e = hipMemset(t->devCmd, 0, cmdBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (t->cmd) { if (t->cmdDirect) (void)hipFree(t->cmd); else (void)hipHostFree(t->cmd); }
    if (t->res) (void)hipHostFree(t->res);
    (void)hipFree(t->devCmd);
    t->cmd = nullptr; t->res = nullptr; t->devCmd = nullptr;
    return hip_fail(e, "tracker session buffers", __FILE__, __LINE__);
  }
  return ITM_OK;
}

static int session_launch(itm_tracker* t, unsigned int firstSeq, hipStream_t st) {
  ++t->session;
  gh_session_kernel<<<kGHGroups, kGHThreads, 0, st>>>(t->cmdDev, t->devCmd, t->devRec, t->resDev, t->recDev, t->session, firstSeq, t->cmdDirect ? 1 : 0);
  ITM_LAUNCH_CHECK();
  t->sessionOpen = true;
  return ITM_OK;
}

// tells the resident kernel to leave (it does so within microseconds; nothing waits for it: later work on the stream queues behind it)
static void session_close(itm_tracker* t) {
  if (!t->sessionOpen) return;
  write_exit(t);
  t->sessionOpen = false;
}

// the command as tagged granules; the last granule's payload is the session the command is for
static void write_command(itm_tracker* t, const GHCommand& c, unsigned int seq) {
  unsigned int w[kCommandWords];
  memcpy(w, &c, sizeof c);
  const unsigned long long tag = (unsigned long long)seq << 32;
  volatile unsigned long long* g = t->cmd;
  __atomic_thread_fence(__ATOMIC_RELEASE);
  for (int i = 0; i < 64; ++i) g[i] = tag | (i < kCommandWords ? w[i] : 0u);     // 512 bytes: whole lines
  __builtin_ia32_sfence();                   // the granules may sit in a write-combining buffer (BAR memory): send them now
}

// one evaluation through the session: same arguments and the same sums as compute_g_and_h
static int session_g_and_h(itm_tracker* trk, const float* depth, int w, int h, const float* viewIntr, const float* pointsMap, const float* normalsMap,
                           int sceneW, int sceneH, const float* sceneIntr, const float* approxInvPose, const float* scenePose,
                           float distThresh, int iterationType, itm_tracker_gh* out, hipStream_t st) {
  memset(out, 0, sizeof *out);
  if (iterationType == ITM_TRACKER_ITERATION_NONE) return ITM_OK;
  if (iterationType < 1 || iterationType > 3) return set_error(ITM_ERR_INVALID, "bad iteration type");
  int rc = session_reserve(trk);
  if (rc) return rc;
  int tileH;
  const int tiles = gh_tiling(w, h, kSessionToHostBlocks, tileH);
  const int rounds = (tiles + kGHGroups - 1) / kGHGroups;
  using clock = std::chrono::steady_clock;
  if (!trk->sessionOpen && trk->session != 0u) {
    // the previous session was told to leave; make sure it has before its exit granule is overwritten (it polls every
    // microsecond, so this is a formality -- but a session that missed its exit would wait out its idle limit)
    const clock::time_point t0 = clock::now();
    unsigned spins = 0;
    while (trk->res->exited != trk->session) {
      __builtin_ia32_pause();
      if ((++spins & 0xfffu) != 0u) continue;
      if (hipStreamQuery(st) != hipErrorNotReady) break;                 // nothing is running on the stream any more
      if (std::chrono::duration<double>(clock::now() - t0).count() > trk->pollTimeoutSeconds) return set_error(ITM_ERR_DEVICE, "tracker session did not end");
    }
  }
  ITM_TT(const auto tt0 = clock::now();)
  GHCommand c;
  memset(&c, 0, sizeof c);
  memcpy(c.p.approxInvPose.m, approxInvPose, 64); memcpy(c.p.scenePose.m, scenePose, 64);
  c.p.vfx = viewIntr[0]; c.p.vfy = viewIntr[1]; c.p.vcx = viewIntr[2]; c.p.vcy = viewIntr[3];
  c.p.sfx = sceneIntr[0]; c.p.sfy = sceneIntr[1]; c.p.scx = sceneIntr[2]; c.p.scy = sceneIntr[3];
  c.p.distThresh = distThresh; c.p.w = w; c.p.h = h; c.p.sceneW = sceneW; c.p.sceneH = sceneH; c.p.tileH = tileH;
  c.depth = depth; c.points = (const float4*)pointsMap; c.normals = (const float4*)normalsMap;
  c.mode = iterationType; c.activeBlocks = (tiles + rounds - 1) / rounds;       // the grid of the per-launch path: same tiles per block, same sums
  const int nBlocks = c.activeBlocks;
  const bool toHost = nBlocks <= kSessionToHostBlocks;
  c.toHost = toHost ? 1 : 0;
  c.session = trk->sessionOpen ? trk->session : trk->session + 1u;
  const unsigned int seq = trk->seq = next_seq(trk->seq);
  write_command(trk, c, seq);
  ITM_TT(const auto ttA = clock::now();)
  if (!trk->sessionOpen && (rc = session_launch(trk, seq, st))) return rc;
  ITM_TT(const auto ttB = clock::now();)
  // Wait for the tagged answer: on coarse levels one record per workgroup, added here in block order (the order of the device-side
  // gather and of the per-launch path), otherwise the one record of the gathering workgroup.  A session that has left without
  // answering (idle limit hit while this thread was away) is replaced; the wait is bounded in time like the per-launch path's.
  clock::time_point t0; bool timing = false;
  int relaunches = 0;
  auto slow = [&](const volatile unsigned long long* g) -> int {
    if (trk->res->exited == trk->session && (unsigned int)(*g >> 32) != seq) {
      ITM_TT(fprintf(stderr, "[tracker trace] %p session %u left without answering %u: relaunch\n", (void*)trk, trk->session, seq);)
      // a session whose workgroups cannot all be resident (compute units masked or taken by another process) leaves again and
      // again without an answer: after a few replacements the caller takes the launch-per-evaluation path instead
      if (++relaunches > 3) { trk->sessionOpen = false; return set_error(kSessionUnusable, "tracker session keeps leaving without an answer"); }
      c.session = trk->session + 1u;           // addressed to the new session: workgroups of the old one that are still around leave at once
      write_command(trk, c, seq);
      return session_launch(trk, seq, st);
    }
    if (!timing) { t0 = clock::now(); timing = true; return ITM_OK; }
    const double waited = std::chrono::duration<double>(clock::now() - t0).count();
    if (waited < 0.02) return ITM_OK;
    const hipError_t q = hipStreamQuery(st);
    if (q != hipSuccess && q != hipErrorNotReady) { trk->sessionOpen = false; return hip_fail(q, "tracker session", __FILE__, __LINE__); }
    if (waited > trk->sessionTimeoutSeconds) { session_close(trk); return set_error(kSessionUnusable, "tracker session timed out"); }
    return ITM_OK;
  };
  double sums[kGHValues];
  for (int i = 0; i < kGHValues; ++i) sums[i] = 0.0;
  int n = 0;
  if (toHost) {
    OrderedSums ordered;
    for (int b = 0; b < nBlocks && !rc; ++b) rc = read_record(trk->rec + b, seq, ordered.of_block(b), &n, slow);
    ordered.total(sums);
  } else {
    const int segments = (nBlocks + kSegBlocks - 1) / kSegBlocks;          // the device added each segment; the segments in order here
    for (int g = 0; g < segments && !rc; ++g) rc = read_record(&trk->res->segment[g], seq, sums, &n, slow);
  }
  if (rc) return rc;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  ITM_TT({ const auto ttC = clock::now();
           const GHResult* r = trk->res;
           fprintf(stderr, "[tracker trace] %dx%d mode %d: command %.2f us, launch %.1f us, answer after %.1f us; on the device, from the command seen: first depth %.2f, first taps %.2f, tiles accumulated %.2f, %s %.2f, record sent %.2f us\n", w, h, iterationType,
                   std::chrono::duration<double, std::micro>(ttA - tt0).count(), std::chrono::duration<double, std::micro>(ttB - ttA).count(),
                   std::chrono::duration<double, std::micro>(ttC - ttB).count(),
                   (double)(r->stamps[4] - r->stamps[0]) / 100.0, (double)(r->stamps[5] - r->stamps[0]) / 100.0, (double)(r->stamps[1] - r->stamps[0]) / 100.0, toHost ? "reduced" : "gathering from", (double)(r->stamps[2] - r->stamps[0]) / 100.0, (double)(r->stamps[3] - r->stamps[0]) / 100.0); })
  const int np = (iterationType == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
  for (int a = 0, k = 0; a < np; ++a)
    for (int b = 0; b <= a; ++b, ++k) out->hessian[a + b * 6] = (float)sums[7 + k];
  for (int a = 0; a < np; ++a)
    for (int b = a + 1; b < np; ++b) out->hessian[a + b * 6] = out->hessian[b + a * 6];
  for (int a = 0; a < np; ++a) out->nabla[a] = (float)sums[1 + a];
  out->noValidPoints = n;
  out->f = (n > 100) ? std::sqrt((float)sums[0]) / n : 1e5f;
  return ITM_OK;
}

// FilterSubsampleWithHoles pyramid of the view's depth image in the tracker's own buffers (PrepareForEvaluation)
struct DepthLevel { const float* depth; int w, h; float intr[4]; };

static int build_pyramid(itm_tracker* trk, const itm_view* view, int levels, std::vector<DepthLevel>& out, hipStream_t st) {
  out.resize(levels);
  out[0].depth = view->depth; out[0].w = view->w; out[0].h = view->h;
  for (int k = 0; k < 4; ++k) out[0].intr[k] = view->intr_d[k];
  if ((int)trk->pyramid.size() < levels) { trk->pyramid.resize(levels, nullptr); trk->pyramidBytes.resize(levels, 0); }
  for (int i = 1; i < levels; ++i) {
    DepthLevel& L = out[i];
    L.w = out[i - 1].w / 2; L.h = out[i - 1].h / 2;
    if (L.w < 1 || L.h < 1) return set_error(ITM_ERR_INVALID, "image too small for the hierarchy");
    const size_t bytes = (size_t)L.w * L.h * 4;
    if (trk->pyramidBytes[i] < bytes) {
      (void)hipFree(trk->pyramid[i]); trk->pyramid[i] = nullptr; trk->pyramidBytes[i] = 0;
      ITM_HIP(hipMalloc((void**)&trk->pyramid[i], bytes));
      trk->pyramidBytes[i] = bytes;
    }
    L.depth = trk->pyramid[i];
    for (int k = 0; k < 4; ++k) L.intr[k] = out[i - 1].intr[k] * 0.5f;
  }
  if (levels >= 2 && levels <= 5 && !g_debug_tracker_launch_per_evaluation) {
    PyramidLevels P;
    memset(&P, 0, sizeof P);
    P.levels = levels - 1;
    for (int i = 0; i < levels; ++i) { P.w[i] = out[i].w; P.h[i] = out[i].h; if (i > 0) P.out[i - 1] = trk->pyramid[i]; }
    pyramid_kernel<<<dim3((view->w + 15) / 16, (view->h + 15) / 16), 256, 0, st>>>(view->depth, P);
  } else {
    for (int i = 1; i < levels; ++i)
      subsample_holes_kernel<<<dim3((out[i].w + 15) / 16, (out[i].h + 15) / 16), 256, 0, st>>>(out[i - 1].depth, out[i - 1].w, trk->pyramid[i], out[i].w, out[i].h);
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

static std::mutex& session_gate(int device) {
  static std::mutex gates[64];
  return gates[(device >= 0 ? device : 0) & 63];
}

static int track_camera(itm_tracker* trk, const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap, const float* normalsMap,
                        const float scenePose[16], float M_d_out[16], hipStream_t st) {
  const int levels = cfg->noHierarchyLevels;
  if (levels < 1 || levels > 8) return set_error(ITM_ERR_INVALID, "noHierarchyLevels must be 1..8");
  int rc = tracker_reserve(trk, 1);
  if (rc) return rc;
  std::vector<DepthLevel> pyr;
  ITM_TT(const auto tc0 = std::chrono::steady_clock::now();)
  if ((rc = build_pyramid(trk, view, levels, pyr, st))) return rc;
  ITM_TT(const auto tc1 = std::chrono::steady_clock::now();)
  if (g_debug_tracker_launch_per_evaluation)
    return icp_track(cfg, view->M_d, M_d_out, [&](int level, int mode, const float invPose[16], float distThresh, itm_tracker_gh* e) {
      return compute_g_and_h(trk, pyr[level].depth, pyr[level].w, pyr[level].h, pyr[level].intr, pointsMap, normalsMap, view->w, view->h,
                             pyr[0].intr, invPose, scenePose, distThresh, mode, e, st);
    });
  // One resident kernel for all evaluations of this call -- and one such kernel per DEVICE at a time: its workgroups take a compute
  // unit's whole register budget for one wave per SIMD (256 VGPRs), so two sessions cannot share a compute unit, and two sessions
  // that have each got hold of SOME compute units wait for the rest until their idle limits fire (measured with four closed loops
  // on four streams: 10 frames/s).  Calls from other handles / streams / threads queue here for the ~100 us of a call; everything
  // else those streams do (view building, fusion, ray casting) runs beside the session.
  // The resident kernel needs all kGHGroups workgroups on the device at once (workgroups wait for each other's records).  Where the
  // device cannot hold them -- fewer compute units than workgroups (partitioned or masked devices, smaller parts) -- and whenever a
  // session proves unusable at run time (it keeps leaving without an answer, or times out: compute units taken by another process),
  // evaluations go through one launch each, which needs no co-residency; the sums and the pose are the same bit for bit.
  if (trk->sessionUsable < 0) {
    int perCU = 0, dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    const bool ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, gh_session_kernel, kGHThreads, 0) == hipSuccess &&
                    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
    (void)hipGetLastError();
    trk->sessionUsable = (ok && (long long)perCU * cus >= kGHGroups) ? 1 : 0;
  }
  auto per_launch = [&](int level, int mode, const float invPose[16], float distThresh, itm_tracker_gh* e) {
    return compute_g_and_h(trk, pyr[level].depth, pyr[level].w, pyr[level].h, pyr[level].intr, pointsMap, normalsMap, view->w, view->h,
                           pyr[0].intr, invPose, scenePose, distThresh, mode, e, st);
  };
  if (!trk->sessionUsable) return icp_track(cfg, view->M_d, M_d_out, per_launch);
  std::lock_guard<std::mutex> oneSession(session_gate(trk->device));
  rc = icp_track(cfg, view->M_d, M_d_out, [&](int level, int mode, const float invPose[16], float distThresh, itm_tracker_gh* e) {
    if (trk->sessionUsable) {
      const int r = (g_debug_tracker_session_unusable > 0 && ++trk->debugEvaluations == g_debug_tracker_session_unusable) ? kSessionUnusable : session_g_and_h(trk, pyr[level].depth, pyr[level].w, pyr[level].h, pyr[level].intr, pointsMap, normalsMap, view->w, view->h,
                                    pyr[0].intr, invPose, scenePose, distThresh, mode, e, st);
      if (r != kSessionUnusable) return r;
      session_close(trk);
      trk->sessionUsable = 0;          // this handle stays on the launch-per-evaluation path
      ++trk->sessionFallbacks;
    }
    return per_launch(level, mode, invPose, distThresh, e);
  });
  session_close(trk);
  ITM_TT(fprintf(stderr, "[tracker trace] call: pyramid launch %.1f us, evaluations %.1f us\n", std::chrono::duration<double, std::micro>(tc1 - tc0).count(),
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tc1).count());)
  return rc;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_filter_subsample_with_holes(const float* in, int w_in, int h_in, float* out, itm_stream stream) {
  if (!in || !out || w_in < 2 || h_in < 2) return set_error(ITM_ERR_INVALID, "bad argument");
  const int w = w_in / 2, h = h_in / 2;
  subsample_holes_kernel<<<dim3((w + 15) / 16, (h + 15) / 16), 256, 0, as_stream(stream)>>>(in, w_in, out, w, h);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_tracker_compute_g_and_h(const float* depth, int w, int h, const float viewIntr[4], const float* pointsMap, const float* normalsMap,
                                int sceneW, int sceneH, const float sceneIntr[4], const float approxInvPose[16], const float scenePose[16],
                                float distThresh, int iterationType, itm_tracker_gh* out, itm_stream stream) {
  return itm_tracker_g_and_h(thread_tracker(), depth, w, h, viewIntr, pointsMap, normalsMap, sceneW, sceneH, sceneIntr, approxInvPose, scenePose,
                             distThresh, iterationType, out, stream);
}

int itm_track_camera(const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap, const float* normalsMap,
                     const float scenePose[16], float M_d_out[16], itm_stream stream) {
  return itm_tracker_track_camera(thread_tracker(), cfg, view, pointsMap, normalsMap, scenePose, M_d_out, stream);
}

int itm_debug_icp_track(const itm_tracker_config* cfg, const float M_d[16], itm_icp_evaluate_fn evaluate, void* user, float M_d_out[16]) {
  if (!cfg || !M_d || !evaluate || !M_d_out) return set_error(ITM_ERR_INVALID, "null argument");
  if (cfg->noHierarchyLevels < 1 || cfg->noHierarchyLevels > 8) return set_error(ITM_ERR_INVALID, "noHierarchyLevels must be 1..8");
  return icp_track(cfg, M_d, M_d_out, [&](int level, int mode, const float invPose[16], float distThresh, itm_tracker_gh* e) {
    return evaluate(user, level, mode, invPose, distThresh, e);
  });
}

int itm_tracker_create(itm_tracker** out) {
  if (!out) return set_error(ITM_ERR_INVALID, "null argument");
  *out = new (std::nothrow) itm_tracker();
  return *out ? ITM_OK : set_error(ITM_ERR_DEVICE, "out of host memory");
}

int itm_tracker_destroy(itm_tracker* t) {
  if (!t) return ITM_OK;
  tracker_release(t);
  delete t;
  return ITM_OK;
}

int itm_tracker_g_and_h(itm_tracker* t, const float* depth, int w, int h, const float viewIntr[4], const float* pointsMap, const float* normalsMap,
                        int sceneW, int sceneH, const float sceneIntr[4], const float approxInvPose[16], const float scenePose[16],
                        float distThresh, int iterationType, itm_tracker_gh* out, itm_stream stream) {
  if (!t || !depth || !viewIntr || !pointsMap || !normalsMap || !sceneIntr || !approxInvPose || !scenePose || !out || w <= 0 || h <= 0)
    return set_error(ITM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lock(t->mu);
  return compute_g_and_h(t, depth, w, h, viewIntr, pointsMap, normalsMap, sceneW, sceneH, sceneIntr, approxInvPose, scenePose, distThresh,
                         iterationType, out, as_stream(stream));
}

int itm_tracker_track_camera(itm_tracker* t, const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap, const float* normalsMap,
                             const float scenePose[16], float M_d_out[16], itm_stream stream) {
  if (!t || !cfg || !view || !view->depth || !pointsMap || !normalsMap || !scenePose || !M_d_out) return set_error(ITM_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(t->mu);
  return track_camera(t, cfg, view, pointsMap, normalsMap, scenePose, M_d_out, as_stream(stream));
}

}  // extern "C"
