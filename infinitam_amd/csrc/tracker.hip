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
// exactly).  Host part: the Levenberg-Marquardt loop, 3x3 / 6x6 Cholesky and the SE(3) re-projection of
// the pose, restated from the reference (plain C++ on the host, as in the reference's CUDA back-end).
#include <cmath>
#include <cstring>
#include <vector>

#include "itm_internal.h"
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
template <int MODE>
__global__ void __launch_bounds__(256) gh_partial_kernel(const float* __restrict__ depth, const float4* __restrict__ pointsMap,
                                                        const float4* __restrict__ normalsMap, double* __restrict__ partial,
                                                        int* __restrict__ partialCount, GHParams p, GHBlockRecord* __restrict__ hostRec, unsigned int seq) {
  constexpr int NP = (MODE == 3) ? 6 : 3;
  constexpr int NH = NP * (NP + 1) / 2;
  __shared__ double lds[4][kGHValues];
  __shared__ int ldsCount[4];
  float vals[kGHValues];
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) vals[i] = 0.0f;
  int valid = 0;
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
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
              valid = 1;
            }
          }
        }
      }
    }
  }
  // wave reduction in double (fixed butterfly order), then one partial per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kGHValues; ++i) {
    const bool used = (i == 0) || (i >= 1 && i < 1 + NP) || (i >= 7 && i < 7 + NH);
    double s = 0.0;
    if (used) {
      s = (double)vals[i];
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
  const int blk = blockIdx.x + blockIdx.y * gridDim.x;
  const double mine = (threadIdx.x < kGHValues) ? ((lds[0][threadIdx.x] + lds[1][threadIdx.x]) + lds[2][threadIdx.x]) + lds[3][threadIdx.x] : 0.0;
  const int cnt = ((ldsCount[0] + ldsCount[1]) + ldsCount[2]) + ldsCount[3];
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

// adds the per-workgroup partials in index order (fixed order => deterministic)
// ---- scratch (one per process and device; calls synchronise anyway) ------------------------------
struct TrackerScratch {
  int device = -1;
  double* partial = nullptr; int* partialCount = nullptr;
  GHBlockRecord* rec = nullptr; GHBlockRecord* recDev = nullptr; size_t recBlocks = 0; unsigned int seq = 0;   // pinned host records + device address
  size_t blocks = 0;
  std::vector<float*> pyramid; std::vector<size_t> pyramidBytes;
};
static TrackerScratch g_scratch;

static int ensure_scratch(size_t blocks) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (g_scratch.device != dev || g_scratch.blocks < blocks) {
    (void)hipFree(g_scratch.partial); (void)hipFree(g_scratch.partialCount);
    g_scratch.partial = nullptr; g_scratch.partialCount = nullptr;
    ITM_HIP(hipMalloc((void**)&g_scratch.partial, blocks * kGHValues * sizeof(double)));
    ITM_HIP(hipMalloc((void**)&g_scratch.partialCount, blocks * sizeof(int)));
    if (g_scratch.rec && (g_scratch.device != dev || g_scratch.recBlocks < blocks)) { (void)hipHostFree(g_scratch.rec); g_scratch.rec = nullptr; }
    if (!g_scratch.rec) {
      ITM_HIP(hipHostMalloc((void**)&g_scratch.rec, blocks * sizeof(GHBlockRecord), hipHostMallocMapped));
      memset(g_scratch.rec, 0, blocks * sizeof(GHBlockRecord));
      ITM_HIP(hipHostGetDevicePointer((void**)&g_scratch.recDev, g_scratch.rec, 0));
      g_scratch.recBlocks = blocks;
    }
    if (g_scratch.device != dev) { g_scratch.pyramid.clear(); g_scratch.pyramidBytes.clear(); }
    g_scratch.device = dev; g_scratch.blocks = blocks;
  }
  return ITM_OK;
}

static int compute_g_and_h(const float* depth, int w, int h, const float* viewIntr, const float* pointsMap, const float* normalsMap,
                           int sceneW, int sceneH, const float* sceneIntr, const float* approxInvPose, const float* scenePose,
                           float distThresh, int iterationType, itm_tracker_gh* out, hipStream_t st) {
  memset(out, 0, sizeof *out);
  if (iterationType == ITM_TRACKER_ITERATION_NONE) return ITM_OK;
  if (iterationType < 1 || iterationType > 3) return set_error(ITM_ERR_INVALID, "bad iteration type");
  const dim3 grid((w + 15) / 16, (h + 15) / 16);
  const size_t blocks = (size_t)grid.x * grid.y;
  int rc = ensure_scratch(blocks);
  if (rc) return rc;
  GHParams p;
  memcpy(p.approxInvPose.m, approxInvPose, 64); memcpy(p.scenePose.m, scenePose, 64);
  p.vfx = viewIntr[0]; p.vfy = viewIntr[1]; p.vcx = viewIntr[2]; p.vcy = viewIntr[3];
  p.sfx = sceneIntr[0]; p.sfy = sceneIntr[1]; p.scx = sceneIntr[2]; p.scy = sceneIntr[3];
  p.distThresh = distThresh; p.w = w; p.h = h; p.sceneW = sceneW; p.sceneH = sceneH;
  const float4* pm = (const float4*)pointsMap; const float4* nm = (const float4*)normalsMap;
  const int np = (iterationType == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
  const int nh = np * (np + 1) / 2;
  const unsigned int seq = ++g_scratch.seq;
  if (iterationType == 1) gh_partial_kernel<1><<<grid, 256, 0, st>>>(depth, pm, nm, g_scratch.partial, g_scratch.partialCount, p, g_scratch.recDev, seq);
  else if (iterationType == 2) gh_partial_kernel<2><<<grid, 256, 0, st>>>(depth, pm, nm, g_scratch.partial, g_scratch.partialCount, p, g_scratch.recDev, seq);
  else gh_partial_kernel<3><<<grid, 256, 0, st>>>(depth, pm, nm, g_scratch.partial, g_scratch.partialCount, p, g_scratch.recDev, seq);
  ITM_LAUNCH_CHECK();
  // wait for every workgroup's record (bounded poll, then a stream synchronisation, which also surfaces device errors)
  // and add them in block order: fixed order => deterministic, in double precision
  double sums[kGHValues];
  for (int i = 0; i < kGHValues; ++i) sums[i] = 0.0;
  int n = 0;
  bool synced = false;
  for (size_t b = 0; b < blocks; ++b) {
    const GHBlockRecord* r = g_scratch.rec + b;
    int spin = 0;
    while (r->seq != seq) {
      __builtin_ia32_pause();
      if (++spin > 4000000) {
        if (synced) return set_error(ITM_ERR_DEVICE, "tracker reduction did not complete");
        ITM_HIP(hipStreamSynchronize(st)); synced = true; spin = 0;
      }
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

// ---- host side: pose algebra and the LM loop -------------------------------------------------------
namespace hostpose {

struct Pose { float t[3], r[3]; float M[16]; };   // params (tx,ty,tz,rx,ry,rz) + model-view matrix

inline float dot3(const float* a, const float* b) { float r = 0; for (int i = 0; i < 3; ++i) r += a[i] * b[i]; return r; }
inline void cross3(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }

// ITMPose::SetModelViewFromParams (Objects/ITMPose.cpp:84-153); R is column-major 3x3, R[r + 3c]
void rotation_from_params(const float* w, const float* t, float* R, float* T) {
  const float one_6th = 1.0f / 6.0f, one_20th = 1.0f / 20.0f;
  const float theta_sq = dot3(w, w);
  const float theta = std::sqrt(theta_sq);
  float A, B;
  float cv[3]; cross3(w, t, cv);
  if (theta_sq < 1e-8f) {
    A = 1.0f - one_6th * theta_sq; B = 0.5f;
    for (int i = 0; i < 3; ++i) T[i] = t[i] + 0.5f * cv[i];
  } else {
    float C;
    if (theta_sq < 1e-6f) {
      C = one_6th * (1.0f - one_20th * theta_sq);
      A = 1.0f - theta_sq * C;
      B = 0.5f - 0.25f * one_6th * theta_sq;
    } else {
      const float inv_theta = 1.0f / theta;
      A = sinf(theta) * inv_theta;
      B = (1.0f - cosf(theta)) * (inv_theta * inv_theta);
      C = (1.0f - A) * (inv_theta * inv_theta);
    }
    float c2[3]; cross3(w, cv, c2);
    for (int i = 0; i < 3; ++i) T[i] = t[i] + B * cv[i] + C * c2[i];
  }
  const float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
  R[0 + 3 * 0] = 1.0f - B * (wy2 + wz2);
  R[1 + 3 * 1] = 1.0f - B * (wx2 + wz2);
  R[2 + 3 * 2] = 1.0f - B * (wx2 + wy2);
  float a, b;
  a = A * w[2]; b = B * (w[0] * w[1]); R[0 + 3 * 1] = b - a; R[1 + 3 * 0] = b + a;
  a = A * w[1]; b = B * (w[0] * w[2]); R[0 + 3 * 2] = b + a; R[2 + 3 * 0] = b - a;
  a = A * w[0]; b = B * (w[1] * w[2]); R[1 + 3 * 2] = b - a; R[2 + 3 * 1] = b + a;
}

void model_view_from_params(Pose& p) {
  float R[9], T[3];
  rotation_from_params(p.r, p.t, R, T);
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) p.M[r + 4 * c] = R[r + 3 * c];
  p.M[12] = T[0]; p.M[13] = T[1]; p.M[14] = T[2];
  p.M[3] = 0.0f; p.M[7] = 0.0f; p.M[11] = 0.0f; p.M[15] = 1.0f;
}

// ITMPose::SetParamsFromModelView (Objects/ITMPose.cpp:155-236)
void params_from_model_view(Pose& p) {
  float R[9], T[3];
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R[r + 3 * c] = p.M[r + 4 * c];
  T[0] = p.M[12]; T[1] = p.M[13]; T[2] = p.M[14];
  float rot[3];
  const float cos_angle = (R[0] + R[4] + R[8] - 1.0f) * 0.5f;
  rot[0] = (R[2 + 3 * 1] - R[1 + 3 * 2]) * 0.5f;
  rot[1] = (R[0 + 3 * 2] - R[2 + 3 * 0]) * 0.5f;
  rot[2] = (R[1 + 3 * 0] - R[0 + 3 * 1]) * 0.5f;
  const float sin_angle_abs = std::sqrt(dot3(rot, rot));
  if (cos_angle > M_SQRT1_2) {
    if (sin_angle_abs) { const float s = asinf(sin_angle_abs) / sin_angle_abs; for (int i = 0; i < 3; ++i) rot[i] *= s; }
  } else if (cos_angle > -M_SQRT1_2) {
    const float s = acosf(cos_angle) / sin_angle_abs; for (int i = 0; i < 3; ++i) rot[i] *= s;
  } else {
    const float angle = (float)M_PI - asinf(sin_angle_abs);
    const float d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle;
    float r2[3];
    if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) {
      r2[0] = d0; r2[1] = (R[1 + 3 * 0] + R[0 + 3 * 1]) * 0.5f; r2[2] = (R[0 + 3 * 2] + R[2 + 3 * 0]) * 0.5f;
    } else if (fabsf(d1) > fabsf(d2)) {
      r2[0] = (R[1 + 3 * 0] + R[0 + 3 * 1]) * 0.5f; r2[1] = d1; r2[2] = (R[2 + 3 * 1] + R[1 + 3 * 2]) * 0.5f;
    } else {
      r2[0] = (R[0 + 3 * 2] + R[2 + 3 * 0]) * 0.5f; r2[1] = (R[2 + 3 * 1] + R[1 + 3 * 2]) * 0.5f; r2[2] = d2;
    }
    if (dot3(r2, rot) < 0.0f) { r2[0] *= -1.0f; r2[1] *= -1.0f; r2[2] *= -1.0f; }
    const float len = std::sqrt(dot3(r2, r2));       // normalize(): vec / length, zero vector stays zero
    if (len == 0) { r2[0] = r2[1] = r2[2] = 0; } else { r2[0] /= len; r2[1] /= len; r2[2] /= len; }
    for (int i = 0; i < 3; ++i) rot[i] = angle * r2[i];
  }
  float shtot = 0.5f;
  const float theta = std::sqrt(dot3(rot, rot));
  if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
  // halfrotor = ITMPose(0,0,0, -rot/2): only its rotation is used
  float hw[3] = {rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f}, zero[3] = {0, 0, 0}, HR[9], HT[3];
  rotation_from_params(hw, zero, HR, HT);
  float rt[3];   // Matrix3 * Vector3: r[i] = m[i]*x + m[i+3]*y + m[i+6]*z
  for (int i = 0; i < 3; ++i) rt[i] = HR[i] * T[0] + HR[i + 3] * T[1] + HR[i + 6] * T[2];
  if (theta > 0.001f) {
    const float denom = dot3(rot, rot);
    const float param = dot3(T, rot) * (1 - 2 * shtot) / denom;
    for (int i = 0; i < 3; ++i) rt[i] -= rot[i] * param;
  } else {
    const float param = dot3(T, rot) / 24;
    for (int i = 0; i < 3; ++i) rt[i] -= rot[i] * param;
  }
  for (int i = 0; i < 3; ++i) rt[i] /= 2 * shtot;
  for (int i = 0; i < 3; ++i) { p.r[i] = rot[i]; p.t[i] = rt[i]; }
}

// ORUtils::Cholesky + Backsub
void cholesky_solve(const float* mat, int n, const float* v, float* result) {
  std::vector<float> ch(mat, mat + n * n);
  for (int c = 0; c < n; ++c) {
    float inv_diag = 1;
    for (int r = c; r < n; ++r) {
      float val = ch[c + r * n];
      for (int c2 = 0; c2 < c; ++c2) val -= ch[c + c2 * n] * ch[c2 + r * n];
      if (r == c) { ch[c + r * n] = val; inv_diag = 1.0f / val; }
      else { ch[r + c * n] = val; ch[c + r * n] = val * inv_diag; }
    }
  }
  std::vector<float> y(n);
  for (int i = 0; i < n; ++i) { float val = v[i]; for (int j = 0; j < i; ++j) val -= ch[j + i * n] * y[j]; y[i] = val; }
  for (int i = 0; i < n; ++i) y[i] /= ch[i + i * n];
  for (int i = n - 1; i >= 0; --i) { float val = y[i]; for (int j = i + 1; j < n; ++j) val -= ch[i + j * n] * result[j]; result[i] = val; }
}

}  // namespace hostpose

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
  if (!depth || !viewIntr || !pointsMap || !normalsMap || !sceneIntr || !approxInvPose || !scenePose || !out || w <= 0 || h <= 0)
    return set_error(ITM_ERR_INVALID, "bad argument");
  return compute_g_and_h(depth, w, h, viewIntr, pointsMap, normalsMap, sceneW, sceneH, sceneIntr, approxInvPose, scenePose, distThresh,
                         iterationType, out, as_stream(stream));
}

int itm_track_camera(const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap, const float* normalsMap,
                     const float scenePose[16], float M_d_out[16], itm_stream stream) {
  using namespace hostpose;
  if (!cfg || !view || !view->depth || !pointsMap || !normalsMap || !scenePose || !M_d_out) return set_error(ITM_ERR_INVALID, "null argument");
  const int L = cfg->noHierarchyLevels;
  if (L < 1 || L > 8) return set_error(ITM_ERR_INVALID, "noHierarchyLevels must be 1..8");
  hipStream_t st = as_stream(stream);
  int rc = ensure_scratch(1);
  if (rc) return rc;
  // ---- depth pyramid (PrepareForEvaluation): level i = FilterSubsampleWithHoles(level i-1), intrinsics * 0.5
  std::vector<const float*> depthL(L); std::vector<int> wL(L), hL(L); std::vector<float> intrL(4 * L);
  depthL[0] = view->depth; wL[0] = view->w; hL[0] = view->h;
  for (int k = 0; k < 4; ++k) intrL[k] = view->intr_d[k];
  if ((int)g_scratch.pyramid.size() < L) { g_scratch.pyramid.resize(L, nullptr); g_scratch.pyramidBytes.resize(L, 0); }
  for (int i = 1; i < L; ++i) {
    wL[i] = wL[i - 1] / 2; hL[i] = hL[i - 1] / 2;
    if (wL[i] < 1 || hL[i] < 1) return set_error(ITM_ERR_INVALID, "image too small for the hierarchy");
    const size_t bytes = (size_t)wL[i] * hL[i] * 4;
    if (g_scratch.pyramidBytes[i] < bytes) {
      (void)hipFree(g_scratch.pyramid[i]); g_scratch.pyramid[i] = nullptr;
      ITM_HIP(hipMalloc((void**)&g_scratch.pyramid[i], bytes));
      g_scratch.pyramidBytes[i] = bytes;
    }
    subsample_holes_kernel<<<dim3((wL[i] + 15) / 16, (hL[i] + 15) / 16), 256, 0, st>>>(depthL[i - 1], wL[i - 1], g_scratch.pyramid[i], wL[i], hL[i]);
    depthL[i] = g_scratch.pyramid[i];
    for (int k = 0; k < 4; ++k) intrL[4 * i + k] = intrL[4 * (i - 1) + k] * 0.5f;
  }
  ITM_LAUNCH_CHECK();
  // distance thresholds and iteration counts per level (ITMDepthTracker ctor, :18-33)
  std::vector<float> distT(L); std::vector<int> iters(L);
  iters[0] = 2; for (int i = 1; i < L; ++i) iters[i] = iters[i - 1] + 2;
  const float stepT = cfg->distThresh / L;
  distT[L - 1] = cfg->distThresh;
  for (int i = L - 2; i >= 0; --i) distT[i] = distT[i + 1] - stepT;

  Pose pose;
  memcpy(pose.M, view->M_d, 64);
  params_from_model_view(pose);                 // ITMPose::SetM keeps M and refreshes the parameters
  float hessian_good[36], nabla_good[6], A[36], step[6];
  memset(hessian_good, 0, sizeof hessian_good); memset(nabla_good, 0, sizeof nabla_good);
  for (int levelId = L - 1; levelId >= cfg->noICPRunTillLevel; --levelId) {
    const int it = cfg->trackingRegime[levelId];
    if (it == ITM_TRACKER_ITERATION_NONE) continue;
    float approxInv[16];
    invert4(pose.M, approxInv);
    Pose lastGood = pose;
    float f_old = 1e20f, lambda = 1.0f;
    const bool shortIt = it != ITM_TRACKER_ITERATION_BOTH;
    for (int iterNo = 0; iterNo < iters[levelId]; ++iterNo) {
      itm_tracker_gh gh;
      rc = compute_g_and_h(depthL[levelId], wL[levelId], hL[levelId], &intrL[4 * levelId], pointsMap, normalsMap, view->w, view->h,
                           &intrL[0], approxInv, scenePose, distT[levelId], it, &gh, st);
      if (rc) return rc;
      if ((gh.noValidPoints <= 0) || (gh.f > f_old)) {
        pose = lastGood;
        invert4(pose.M, approxInv);
        lambda *= 10.0f;
      } else {
        lastGood = pose;
        f_old = gh.f;
        for (int i = 0; i < 36; ++i) hessian_good[i] = gh.hessian[i] / gh.noValidPoints;
        for (int i = 0; i < 6; ++i) nabla_good[i] = gh.nabla[i] / gh.noValidPoints;
        lambda /= 10.0f;
      }
      for (int i = 0; i < 36; ++i) A[i] = hessian_good[i];
      for (int i = 0; i < 6; ++i) A[i + i * 6] *= 1.0f + lambda;
      for (int i = 0; i < 6; ++i) step[i] = 0;
      if (shortIt) {
        float small[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) small[r + c * 3] = A[r + c * 6];
        cholesky_solve(small, 3, nabla_good, step);
      } else {
        cholesky_solve(A, 6, nabla_good, step);
      }
      // ApplyDelta: para_new = Tinc * para_old
      float s6[6] = {0, 0, 0, 0, 0, 0};
      if (it == ITM_TRACKER_ITERATION_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
      else if (it == ITM_TRACKER_ITERATION_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
      else { for (int i = 0; i < 6; ++i) s6[i] = step[i]; }
      float Tinc[16];   // m[4*col + row]
      Tinc[0] = 1.0f;    Tinc[4] = s6[2];   Tinc[8] = -s6[1];  Tinc[12] = s6[3];
      Tinc[1] = -s6[2];  Tinc[5] = 1.0f;    Tinc[9] = s6[0];   Tinc[13] = s6[4];
      Tinc[2] = s6[1];   Tinc[6] = -s6[0];  Tinc[10] = 1.0f;   Tinc[14] = s6[5];
      Tinc[3] = 0.0f;    Tinc[7] = 0.0f;    Tinc[11] = 0.0f;   Tinc[15] = 1.0f;
      float newInv[16];
      matmul4(Tinc, approxInv, newInv);
      // SetInvM + Coerce + GetInvM
      invert4(newInv, pose.M);
      params_from_model_view(pose);
      params_from_model_view(pose);
      model_view_from_params(pose);
      invert4(pose.M, approxInv);
      float len = 0.0f;
      for (int i = 0; i < 6; ++i) len += step[i] * step[i];
      if (std::sqrt(len) / 6 < cfg->terminationThreshold) break;
    }
  }
  memcpy(M_d_out, pose.M, 64);
  return ITM_OK;
}

}  // extern "C"
