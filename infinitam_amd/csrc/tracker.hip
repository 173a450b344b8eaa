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
// operations; the 1 + 6 + 21 sums and the valid count are reduced with wave shuffles, one partial per
// workgroup, written to a stamped record in pinned host memory; the host adds the records in block order in double
// precision (deterministic; the reference adds floats in raster order, so sums agree to float rounding, the count
// exactly).  Host part: a damped Gauss-Newton iteration over SE(3) with the reference's accept / reject schedule, written
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

struct GHParams {
  Mat4 approxInvPose, scenePose;
  float vfx, vfy, vcx, vcy;   // view intrinsics (fx, fy, cx, cy)
  float sfx, sfy, scx, scy;   // scene intrinsics
  float distThresh;
  int w, h, sceneW, sceneH;
};

constexpr int kGHValues = 1 + 6 + 21;   // f, nabla, packed lower-triangular hessian

struct GHBlockRecord { double sums[kGHValues]; int count; volatile unsigned int seq; };   // one per workgroup, in pinned host memory

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
#ifndef ITM_GH_GROUPS
#define ITM_GH_GROUPS 256
#endif
#ifndef ITM_GH_WAVES
#define ITM_GH_WAVES 4           // waves per workgroup (measured per 640x480 evaluation: 4 waves 43 us, 8 waves 51 us, 16 waves 73 us)
#endif
constexpr int kGHGroups = ITM_GH_GROUPS;
constexpr int kGHWaves = ITM_GH_WAVES;
constexpr int kGHThreads = 64 * kGHWaves;
constexpr int kGHTileH = kGHThreads / 16;      // a tile is 16 pixels wide and kGHTileH tall

template <int MODE>
__global__ void __launch_bounds__(kGHThreads) gh_partial_kernel(const float* __restrict__ depth, const float4* __restrict__ pointsMap,
                                                        const float4* __restrict__ normalsMap, double* __restrict__ partial,
                                                        int* __restrict__ partialCount, GHParams p, GHBlockRecord* __restrict__ hostRec, unsigned int seq) {
  constexpr int NP = (MODE == 3) ? 6 : 3;
  constexpr int NH = NP * (NP + 1) / 2;
  __shared__ double lds[kGHWaves][kGHValues];
  __shared__ int ldsCount[kGHWaves];
  double acc[kGHValues];
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) acc[i] = 0.0;
  int valid = 0;
  const int tilesX = (p.w + 15) / 16, tiles = tilesX * ((p.h + kGHTileH - 1) / kGHTileH);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
  float vals[kGHValues];
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) vals[i] = 0.0f;
  const int x = (tile % tilesX) * 16 + (threadIdx.x & 15), y = (tile / tilesX) * kGHTileH + (threadIdx.x >> 4);
  if (x < p.w && y < p.h) {
    const float d = depth[x + y * p.w];
    if (!(d <= 1e-8f)) {
      // back-project, move to the scene frame, re-project into the rendered maps
      const float cx3 = d * (((float)x - p.vcx) / p.vfx), cy3 = d * (((float)y - p.vcy) / p.vfy);
      const Vec3 q = transform_point(p.approxInvPose, cx3, cy3, d);
      const Vec3 rp = transform_point(p.scenePose, q.x, q.y, q.z);
      if (!(rp.z <= 0.0f)) {
        const float u = p.sfx * rp.x / rp.z + p.scx, v = p.sfy * rp.y / rp.z + p.scy;
        if ((u >= 0.0f) && (u <= p.sceneW - 2) && (v >= 0.0f) && (v <= p.sceneH - 2)) {
          float4 cp;
          if (bilinear_holes(pointsMap, u, v, p.sceneW, cp) && !(cp.w < 0.0f)) {
            const float ex = cp.x - q.x, ey = cp.y - q.y, ez = cp.z - q.z;
            const float dist = ex * ex + ey * ey + ez * ez;
            float4 n;
            if (!(dist > p.distThresh)) {
              // a hole in the normals map yields the zero normal but still counts (the reference's check is commented out)
              if (!bilinear_holes(normalsMap, u, v, p.sceneW, n)) n = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
              const float b = n.x * ex + n.y * ey + n.z * ez;
              float A[NP];
              if (MODE == 2) { A[0] = n.x; A[1] = n.y; A[2] = n.z; }
              else {
                A[0] = +q.z * n.y - q.y * n.z;
                A[1] = -q.z * n.x + q.x * n.z;
                A[2] = +q.y * n.x - q.x * n.y;
                if (MODE == 3) { A[3] = n.x; A[4] = n.y; A[5] = n.z; }
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
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) {
    const bool used = (i == 0) || (i >= 1 && i < 1 + NP) || (i >= 7 && i < 7 + NH);
    if (used) acc[i] += (double)vals[i];
  }
  }   // tiles of this workgroup
  // wave reduction in double (fixed butterfly order), then one partial per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) {
    const bool used = (i == 0) || (i >= 1 && i < 1 + NP) || (i >= 7 && i < 7 + NH);
    double s = 0.0;
    if (used) {
      s = acc[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    }
    if (lane == 0) lds[wave][i] = s;
  }
  int c = valid;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if (lane == 0) ldsCount[wave] = c;
  __syncthreads();
  const int blk = blockIdx.x;
  double mine = 0.0;
  int cnt = 0;
#pragma unroll
  for (int wv = 0; wv < kGHWaves; ++wv) {          // fixed order: deterministic
    if (threadIdx.x < kGHValues) mine += lds[wv][threadIdx.x];
    cnt += ldsCount[wv];
  }
  if (threadIdx.x < kGHValues) partial[(size_t)blk * kGHValues + threadIdx.x] = mine;
  if (threadIdx.x == 0) partialCount[blk] = cnt;
  if (hostRec) {
    // the same partial goes to a record in pinned host memory, stamped with the call's sequence number: the host adds
    // the records in block order itself (one launch per Levenberg-Marquardt iteration, no reduction launch, no copy)
    GHBlockRecord* r = hostRec + blk;
    if (threadIdx.x < kGHValues) r->sums[threadIdx.x] = mine;
    if (threadIdx.x == 0) r->count = cnt;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { r->seq = seq; __threadfence_system(); }
  }
}

// ---- tracker object -------------------------------------------------------------------------------------
// What ITMDepthTracker owns in the reference (Engine/ITMDepthTracker.cpp:18-44: the hierarchy and the device-side reduction
// buffers of ITMDepthTracker_CUDA) lives in a handle here: device partials, the stamped records in pinned host memory, the
// depth pyramid.  One handle = one tracker = one caller at a time (the handle's mutex serialises callers); the handle-less
// entry points of round 1 use a handle private to the calling thread, so two host threads (or two streams driven by two
// threads) never share records or sequence numbers.
}  // namespace itm

struct itm_tracker {
  std::mutex mu;
  int device = -1;
  double* partial = nullptr; int* partialCount = nullptr;
  itm::GHBlockRecord* rec = nullptr; itm::GHBlockRecord* recDev = nullptr;   // pinned host records + their device address
  size_t blocks = 0;
  unsigned int seq = 0;
  std::vector<float*> pyramid; std::vector<size_t> pyramidBytes;
  double pollTimeoutSeconds = 5.0;
};

namespace itm {

static void tracker_release(itm_tracker* t) {
  (void)hipFree(t->partial); (void)hipFree(t->partialCount);
  if (t->rec) (void)hipHostFree(t->rec);
  for (float* q : t->pyramid) (void)hipFree(q);
  t->partial = nullptr; t->partialCount = nullptr; t->rec = nullptr; t->recDev = nullptr; t->blocks = 0;
  t->pyramid.clear(); t->pyramidBytes.clear();
}

// (re)sizes the reduction buffers for `blocks` workgroups on the current device; on failure the handle is left empty
static int tracker_reserve(itm_tracker* t, size_t blocks) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (t->device == dev && t->blocks >= blocks) return ITM_OK;
  if (t->device != dev) tracker_release(t);           // buffers of another device (incl. the pyramid) are of no use here
  else {
    (void)hipFree(t->partial); (void)hipFree(t->partialCount);
    if (t->rec) (void)hipHostFree(t->rec);
    t->partial = nullptr; t->partialCount = nullptr; t->rec = nullptr; t->recDev = nullptr;
  }
  t->blocks = 0; t->device = dev;
  hipError_t e = hipMalloc((void**)&t->partial, blocks * kGHValues * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&t->partialCount, blocks * sizeof(int));
  // coherent + mapped: device stores become visible to the polling host without a kernel boundary
  if (e == hipSuccess) e = hipHostMalloc((void**)&t->rec, blocks * sizeof(GHBlockRecord), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) { memset(t->rec, 0, blocks * sizeof(GHBlockRecord)); e = hipHostGetDevicePointer((void**)&t->recDev, t->rec, 0); }
  if (e != hipSuccess) {
    (void)hipFree(t->partial); (void)hipFree(t->partialCount);
    if (t->rec) (void)hipHostFree(t->rec);
    t->partial = nullptr; t->partialCount = nullptr; t->rec = nullptr; t->recDev = nullptr;
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

static int compute_g_and_h(itm_tracker* trk, const float* depth, int w, int h, const float* viewIntr, const float* pointsMap, const float* normalsMap,
                           int sceneW, int sceneH, const float* sceneIntr, const float* approxInvPose, const float* scenePose,
                           float distThresh, int iterationType, itm_tracker_gh* out, hipStream_t st) {
  memset(out, 0, sizeof *out);
  if (iterationType == ITM_TRACKER_ITERATION_NONE) return ITM_OK;
  if (iterationType < 1 || iterationType > 3) return set_error(ITM_ERR_INVALID, "bad iteration type");
  const int tiles = ((w + 15) / 16) * ((h + kGHTileH - 1) / kGHTileH);
  const int rounds = (tiles + kGHGroups - 1) / kGHGroups;                  // tiles per workgroup, then as few workgroups as that needs
  const dim3 grid((tiles + rounds - 1) / rounds);
  const size_t blocks = grid.x;
  int rc = tracker_reserve(trk, blocks);
  if (rc) return rc;
  GHParams p;
  memcpy(p.approxInvPose.m, approxInvPose, 64); memcpy(p.scenePose.m, scenePose, 64);
  p.vfx = viewIntr[0]; p.vfy = viewIntr[1]; p.vcx = viewIntr[2]; p.vcy = viewIntr[3];
  p.sfx = sceneIntr[0]; p.sfy = sceneIntr[1]; p.scx = sceneIntr[2]; p.scy = sceneIntr[3];
  p.distThresh = distThresh; p.w = w; p.h = h; p.sceneW = sceneW; p.sceneH = sceneH;
  const float4* pm = (const float4*)pointsMap; const float4* nm = (const float4*)normalsMap;
  const int np = (iterationType == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
  const int nh = np * (np + 1) / 2;
  const unsigned int seq = ++trk->seq;
  if (iterationType == 1) gh_partial_kernel<1><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, trk->partial, trk->partialCount, p, trk->recDev, seq);
  else if (iterationType == 2) gh_partial_kernel<2><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, trk->partial, trk->partialCount, p, trk->recDev, seq);
  else gh_partial_kernel<3><<<grid, kGHThreads, 0, st>>>(depth, pm, nm, trk->partial, trk->partialCount, p, trk->recDev, seq);
  ITM_LAUNCH_CHECK();
  // Wait for every workgroup's stamped record and add them in block order (fixed order => deterministic, in double).  The
  // poll is bounded in TIME: after 20 ms without the stamp the stream is queried between polls -- a drained stream without
  // the stamp, a device error, or pollTimeoutSeconds without progress end the call with ITM_ERR_DEVICE instead of
  // stalling the host on a kernel that will never finish.
  double sums[kGHValues];
  for (int i = 0; i < kGHValues; ++i) sums[i] = 0.0;
  int n = 0;
  using clock = std::chrono::steady_clock;
  clock::time_point t0; bool timing = false;
  for (size_t b = 0; b < blocks; ++b) {
    const GHBlockRecord* r = trk->rec + b;
    unsigned spins = 0;
    while (r->seq != seq) {
      __builtin_ia32_pause();
      if ((++spins & 0x3ffu) != 0u) continue;
      if (!timing) { t0 = clock::now(); timing = true; continue; }
      const double waited = std::chrono::duration<double>(clock::now() - t0).count();
      if (waited < 0.02) continue;
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) {
        if (r->seq == seq) break;
        return set_error(ITM_ERR_DEVICE, "tracker reduction: the stream drained without delivering every record");
      }
      if (q != hipErrorNotReady) return hip_fail(q, "tracker reduction", __FILE__, __LINE__);
      if (waited > trk->pollTimeoutSeconds) return set_error(ITM_ERR_DEVICE, "tracker reduction timed out");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < kGHValues; ++i) sums[i] += r->sums[i];
    n += r->count;
  }
  for (int r = 0, k = 0; r < np; ++r)
    for (int c = 0; c <= r; ++c, ++k) out->hessian[r + c * 6] = (float)sums[7 + k];
  for (int r = 0; r < np; ++r)
    for (int c = r + 1; c < np; ++c) out->hessian[r + c * 6] = out->hessian[c + r * 6];
  for (int r = 0; r < np; ++r) out->nabla[r] = (float)sums[1 + r];
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
    subsample_holes_kernel<<<dim3((L.w + 15) / 16, (L.h + 15) / 16), 256, 0, st>>>(out[i - 1].depth, out[i - 1].w, trk->pyramid[i], L.w, L.h);
    L.depth = trk->pyramid[i];
    for (int k = 0; k < 4; ++k) L.intr[k] = out[i - 1].intr[k] * 0.5f;
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

static int track_camera(itm_tracker* trk, const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap, const float* normalsMap,
                        const float scenePose[16], float M_d_out[16], hipStream_t st) {
  const int levels = cfg->noHierarchyLevels;
  if (levels < 1 || levels > 8) return set_error(ITM_ERR_INVALID, "noHierarchyLevels must be 1..8");
  int rc = tracker_reserve(trk, 1);
  if (rc) return rc;
  std::vector<DepthLevel> pyr;
  if ((rc = build_pyramid(trk, view, levels, pyr, st))) return rc;
  return icp_track(cfg, view->M_d, M_d_out, [&](int level, int mode, const float invPose[16], float distThresh, itm_tracker_gh* e) {
    return compute_g_and_h(trk, pyr[level].depth, pyr[level].w, pyr[level].h, pyr[level].intr, pointsMap, normalsMap, view->w, view->h,
                           pyr[0].intr, invPose, scenePose, distThresh, mode, e, st);
  });
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
