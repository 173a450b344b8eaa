// viewbuilder.hip -- the device part of the view builder that precedes the path: 5x5 bilateral depth
// filter, normal + depth-uncertainty images, and the UpdateView sequence for a raw frame resident in HBM.
//
// Reference behaviour:
//   filterDepth / computeNormalAndWeight   DeviceAgnostic/ITMViewBuilder.h:30-117
//   DepthFiltering / ComputeNormalAndWeights / UpdateView   DeviceSpecific/CPU/ITMViewBuilder_CPU.cpp:14-63, 119-145
// Arithmetic follows the reference operation for operation (no contraction); the only non-IEEE functions are
// exp (filter weights) and acos (uncertainty), for which the device library and the host libm may differ in the last
// place -- parity tolerance 1e-6 relative for the filtered depth, exact for everything else that does not depend on them.
#include <atomic>
#include <chrono>
#include <memory>
#include <new>
#include <vector>

#include "itm_internal.h"

namespace itm {

// 5x5 bilateral filter of the depth image (filterDepth).  A workgroup filters a 16x16 tile from a 20x20 copy in LDS (25 taps per
// pixel: 6.25 reads of global memory per pixel become 1.6).  Per pixel the taps are visited row by row, left to right, and the two
// running sums grow in that order, as in the reference; a tap with a negative (invalid) depth is skipped, an invalid centre gives -1,
// the two-pixel border of the image stays 0.
constexpr int kFilterRadius = 2;
constexpr int kFilterTile = 16 + 2 * kFilterRadius;

__global__ void __launch_bounds__(256) filter_depth_kernel(const float* __restrict__ in, float* __restrict__ out, int w, int h) {
  __shared__ float tile[kFilterTile][kFilterTile + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tx = lane & 15, ty = wave * 4 + (lane >> 4);
  const int x0 = blockIdx.x * 16 - kFilterRadius, y0 = blockIdx.y * 16 - kFilterRadius;
  for (int i = threadIdx.x; i < kFilterTile * kFilterTile; i += 256) {
    const int lx = i % kFilterTile, ly = i / kFilterTile;
    const int gx = x0 + lx, gy = y0 + ly;
    tile[ly][lx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? in[gx + gy * w] : -1.0f;      // never used by a pixel that is filtered
  }
  __syncthreads();
  const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
  if (x >= w || y >= h) return;
  float filtered = 0.0f;
  const bool interior = x >= kFilterRadius && x < w - kFilterRadius && y >= kFilterRadius && y < h - kFilterRadius;
  if (interior) {
    const float centre = tile[ty + kFilterRadius][tx + kFilterRadius];
    if (centre < 0.0f) filtered = -1.0f;
    else {
      // range term: the sensor's depth uncertainty at this distance (its reciprocal); spatial term: city-block distance
      const float off = centre - 0.4f;
      const float invSigma = 1.0f / (0.0012f + 0.0019f * off * off + 0.0001f / sqrtf(centre) * 0.25f);
      const float kSpatial = 1.2232f;
      float weighted = 0.0f, weights = 0.0f;
#pragma unroll
      for (int dy = -kFilterRadius; dy <= kFilterRadius; ++dy)
#pragma unroll
        for (int dx = -kFilterRadius; dx <= kFilterRadius; ++dx) {
          const float tap = tile[ty + kFilterRadius + dy][tx + kFilterRadius + dx];
          if (tap < 0.0f) continue;
          float diff = tap - centre; diff *= diff;
          const int blocks = (dy < 0 ? -dy : dy) + (dx < 0 ? -dx : dx);
          const float g = expf(-0.5f * ((float)blocks * kSpatial * kSpatial + diff * invSigma * invSigma));
          weights += g;
          weighted += g * tap;
        }
      filtered = weighted / weights;
    }
  }
  out[x + y * w] = filtered;
}

// Normal and depth uncertainty per pixel (computeNormalAndWeight): the four axis neighbours are back-projected, the normal is the
// cross product of the two central differences, the uncertainty grows with distance and with the angle between normal and view axis.
// Rejected pixels only get normal.w = -1 and sigma = -1 (the other components keep their old value, as in the reference); the
// two-pixel border is never written.
__global__ void __launch_bounds__(256) normal_weight_kernel(const float* __restrict__ depth, float4* __restrict__ normals, float* __restrict__ sigmaZ,
                                                            int w, int h, float invFx, float invFy, float cx, float cy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (lane & 15);
  const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
  if (x < 2 || x >= w - 2 || y < 2 || y >= h - 2) return;
  const int at = x + y * w;
  auto reject = [&]() { normals[at].w = -1.0f; sigmaZ[at] = -1.0f; };
  const float z = depth[at];
  if (z < 0.0f) { reject(); return; }
  const float zE = depth[at + 1], zS = depth[at + w], zW = depth[at - 1], zN = depth[at - w];
  if (zE <= 0 || zS <= 0 || zW <= 0 || zN <= 0) { reject(); return; }
  // back-projection of pixel (u, v) at depth d: (d (u - cx) / fx, d (v - cy) / fy, d), with the reference's multiplication order
  const float u = (float)x, v = (float)y;
  struct P3 { float x, y, z; };
  auto lift = [&](float d, float pu, float pv) { return P3{d * (pu - cx) * invFx, d * (pv - cy) * invFy, d}; };
  const P3 e = lift(zE, u + 1.0f, v), wv = lift(zW, u - 1.0f, v), sv = lift(zS, u, v + 1.0f), nv = lift(zN, u, v - 1.0f);
  const P3 ax{e.x - wv.x, e.y - wv.y, e.z - wv.z};        // along the row
  const P3 ay{sv.x - nv.x, sv.y - nv.y, sv.z - nv.z};     // along the column
  float nx = ax.y * ay.z - ax.z * ay.y;
  float ny = ax.z * ay.x - ax.x * ay.z;
  float nz = ax.x * ay.y - ax.y * ay.x;
  if (nx == 0.0f && ny == 0.0f && nz == 0.0f) { reject(); return; }
  const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
  nx *= inv; ny *= inv; nz *= inv;
  normals[at] = make_float4(nx, ny, nz, 1.0f);
  const float halfPi = 3.1415926535897932384626433832795f * 0.5f;
  const float tilt = acosf(nz);
  const float slope = tilt / (halfPi - tilt);
  const float off = z - 0.4f;
  sigmaZ[at] = 0.0012f + 0.0019f * off * off + 0.0001f / sqrtf(z) * slope * slope;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_filter_depth(const float* in, float* out, int w, int h, itm_stream stream) {
  if (!in || !out || in == out || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  { const int rc = flush_overlapping(out, (size_t)w * h * 4, as_stream(stream)); if (rc) return rc; }
  const dim3 grid((w + 15) / 16, (h + 15) / 16);
  filter_depth_kernel<<<grid, 256, 0, as_stream(stream)>>>(in, out, w, h);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_compute_normal_and_weights(const float* depth, float* normals, float* sigmaZ, int w, int h, const float intr[4], itm_stream stream) {
  if (!depth || !normals || !sigmaZ || !intr || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  const dim3 grid((w + 15) / 16, (h + 15) / 16);
  normal_weight_kernel<<<grid, 256, 0, as_stream(stream)>>>(depth, (float4*)normals, sigmaZ, w, h, intr[0], intr[1], intr[2], intr[3]);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_update_view(const int16_t* raw, int w, int h, int calibType, float c0, float c1, const float intr_d[4], int useBilateralFilter,
                    int modelSensorNoise, float* depth_out, float* scratch, float* normals, float* sigmaZ, itm_stream stream) {
  if (!raw || !depth_out || !intr_d || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  if (useBilateralFilter && !scratch) return set_error(ITM_ERR_INVALID, "the bilateral filter needs a scratch image");
  if (modelSensorNoise && (!normals || !sigmaZ)) return set_error(ITM_ERR_INVALID, "sensor-noise model needs normal / uncertainty images");
  int rc;
  if (calibType == 0) rc = itm_convert_disparity(raw, depth_out, w, h, c0, c1, intr_d[0], stream);
  else if (calibType == 1) rc = itm_convert_depth_affine(raw, depth_out, w, h, c0, c1, stream);
  else return set_error(ITM_ERR_INVALID, "unknown disparity calibration type");
  if (rc) return rc;
  if (useBilateralFilter) {   // five passes, then the result is copied back into the view's depth image
    if ((rc = itm_filter_depth(depth_out, scratch, w, h, stream))) return rc;
    if ((rc = itm_filter_depth(scratch, depth_out, w, h, stream))) return rc;
    if ((rc = itm_filter_depth(depth_out, scratch, w, h, stream))) return rc;
    if ((rc = itm_filter_depth(scratch, depth_out, w, h, stream))) return rc;
    if ((rc = itm_filter_depth(depth_out, scratch, w, h, stream))) return rc;
    ITM_HIP(hipMemcpyAsync(depth_out, scratch, (size_t)w * h * 4, hipMemcpyDeviceToDevice, as_stream(stream)));
  }
  if (modelSensorNoise) return itm_compute_normal_and_weights(depth_out, normals, sigmaZ, w, h, intr_d, stream);
  return ITM_OK;
}

}  // extern "C"
namespace itm {
// The copy of a raw frame from mapped page-locked host memory, by a kernel on the stager's copy stream.  SIXTEEN workgroups with eight
// 16-byte loads in flight per lane, not a lane per 16 bytes: 150 workgroups' worth of PCIe reads issued at once slowed the frame they
// run beside -- the ray cast's latency-bound tail, 36 -> 47 us while the copy was in flight (rocprofv3 trace of main_engine_demo
// --bench-map-host) -- frames from host memory 9.94 k -> 10.6-11.1 k frames/s with 8-24 workgroups (4: the copy itself becomes the bound).
constexpr int kStageWorkgroups = 16;
// convertDepthAffineToFloat / convertDisparityToDepth (DeviceAgnostic/ITMViewBuilder.h:7-28): the operations of depth_affine_kernel /
// depth_disparity_kernel (scene.hip)
__device__ inline float stage_convert(int16_t raw, int calibType, float c0, float c1, float fx) {
  if (calibType == 1) return ((raw <= 0) || (raw > 32000)) ? -1.0f : (float)raw * c0 + c1;
  const float t = c0 - (float)raw;
  float depth;
  if (t == 0) depth = 0.0f; else depth = 8.0f * c1 * fx / t;
  return (depth > 0) ? depth : -1.0f;
}
// CONVERT: the float depth image of itm_update_view's first step is written beside the raw copy (8 shorts -> 8 floats per 16 bytes read)
template <bool CONVERT>
__global__ void __launch_bounds__(256) stage_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, float4* __restrict__ depth,
                                                         int calibType, float c0, float c1, float fx) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t i = i0 + k * stride; v[k] = src[i < n ? i : i0]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = i0 + k * stride;
      if (i >= n) continue;
      dst[i] = v[k];
      if constexpr (CONVERT) {
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = stage_convert((int16_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu), calibType, c0, c1, fx);
        depth[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
        depth[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
  }
}
}  // namespace itm
extern "C" {

// ---- raw frames from the host --------------------------------------------------------------------------------------------------
// The reference's UpdateView starts with a synchronous copy of the raw image to the device (shortImage->SetFrom(rawDepthImage,
// CPU_TO_CUDA), DeviceSpecific/CUDA/ITMViewBuilder_CUDA.cu:53).  Here the uploads ride on a stream of their own, up to slots - 1
// frames ahead of the frame being fused, into a ring of device buffers; events tie the two streams together both ways: a slot is handed
// to the frame's stream only behind its copy, and overwritten only behind the work that read it.
struct itm_depth_stager {
  int w = 0, h = 0, slots = 0;
  int device = 0;
  hipStream_t copy = nullptr;
  std::vector<int16_t*> buf;
  std::vector<float*> depth;                  // per slot, once a conversion has been set: the float depth image of the slot's frame
  int calibType = -1; float c0 = 0, c1 = 0, fx = 0;
  std::vector<hipEvent_t> uploaded, consumed;
  // one thread may upload while another acquires / releases (a producer beside the frame loop): the counters both sides read are atomic
  std::unique_ptr<std::atomic<char>[]> consumedRecorded;
  std::atomic<unsigned long long> head{0}, tail{0};      // frames released / uploaded so far
  bool held = false;                          // a frame has been acquired and not yet released (consumer side only)
};

static void free_stager(itm_depth_stager* g) {
  if (!g) return;
  if (g->copy) { (void)hipStreamSynchronize(g->copy); (void)hipStreamDestroy(g->copy); }
  for (auto e : g->uploaded) if (e) (void)hipEventDestroy(e);
  for (auto e : g->consumed) if (e) (void)hipEventDestroy(e);
  for (auto b : g->buf) if (b) (void)hipFree(b);
  for (auto b : g->depth) if (b) (void)hipFree(b);
  delete g;
}

int itm_depth_stager_create(int w, int h, int slots, itm_depth_stager** out) {
  if (!out || w <= 0 || h <= 0 || slots < 2 || slots > 64) return set_error(ITM_ERR_INVALID, "bad argument");
  itm_depth_stager* g = new (std::nothrow) itm_depth_stager();
  if (!g) return set_error(ITM_ERR_DEVICE, "out of host memory");
  g->w = w; g->h = h; g->slots = slots;
  g->buf.assign(slots, nullptr); g->uploaded.assign(slots, nullptr); g->consumed.assign(slots, nullptr);
  g->consumedRecorded.reset(new (std::nothrow) std::atomic<char>[slots]);
  if (!g->consumedRecorded) { delete g; return set_error(ITM_ERR_DEVICE, "out of host memory"); }
  for (int i = 0; i < slots; ++i) g->consumedRecorded[i].store(0);
  hipError_t e = hipGetDevice(&g->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking);
  for (int i = 0; i < slots && e == hipSuccess; ++i) {
    e = hipMalloc((void**)&g->buf[i], (size_t)w * h * sizeof(int16_t));
    // DEVICE-scope release: producer and consumer of a slot are streams of this device.  The default (system-scope) release of
    // hipEventRecord writes back and invalidates the L2s -- recorded on the FRAME's stream once per frame (`consumed`), that made every
    // frame's kernels start on cold caches (the exchange found the same, exchange.hip)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->uploaded[i], hipEventDisableTiming | hipEventReleaseToDevice);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->consumed[i], hipEventDisableTiming | hipEventReleaseToDevice);
  }
  if (e != hipSuccess) { free_stager(g); return hip_fail(e, "depth stager", __FILE__, __LINE__); }
  *out = g;
  return ITM_OK;
}

int itm_depth_stager_destroy(itm_depth_stager* g) { free_stager(g); return ITM_OK; }

int itm_depth_stager_upload(itm_depth_stager* g, const int16_t* host) {
  if (!g || !host) return set_error(ITM_ERR_INVALID, "null argument");
  const unsigned long long tail = g->tail.load(std::memory_order_relaxed);
  if (tail - g->head.load(std::memory_order_acquire) >= (unsigned long long)g->slots) return set_error(ITM_ERR_INVALID, "every slot of the stager holds a frame that has not been released");
  const int b = (int)(tail % (unsigned long long)g->slots);
  if (g->consumedRecorded[b].load(std::memory_order_acquire)) {
    // the work that read the slot's previous frame: with the uploads a frame or two ahead it finished long ago, and a wait the copy
    // stream need not make is worth avoiding (a copy queued behind another queue's event costs the CALL ~100 us on this runtime)
    // A host that submits frames faster than the device fuses them arrives here before the slot's reader has run: it WAITS here, polling
    // (a query is a load; this is the back-pressure that keeps the host at most `slots - 1` frames ahead of the device, which never runs
    // dry: the frames in between are queued).  Handing the wait to the copy stream instead (hipStreamWaitEvent) was what round 4 did and
    // cost the CALL ~100 us every frame once the host ran ahead -- the whole difference between frames from host memory and frames in HBM.
    hipError_t q = hipEventQuery(g->consumed[b]);
    if (q == hipErrorNotReady) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0; q == hipErrorNotReady; ++spins) {
        if ((spins & 0xffu) == 0xffu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        __builtin_ia32_pause();
        q = hipEventQuery(g->consumed[b]);
      }
      if (q == hipErrorNotReady) { ITM_HIP(hipStreamWaitEvent(g->copy, g->consumed[b], 0)); q = hipSuccess; }      // (a device that is 2 s behind: let the stream sort it out)
      else if (q == hipSuccess) g->consumedRecorded[b].store(0, std::memory_order_relaxed);
    } else if (q == hipSuccess) g->consumedRecorded[b].store(0, std::memory_order_relaxed);
    if (q != hipSuccess) return hip_fail(q, "hipEventQuery(slot consumed)", __FILE__, __LINE__);
  }
  // Pinned host memory is mapped into the device's address space: a kernel on the copy stream reads the frame over PCIe itself.  (The
  // runtime's own asynchronous copy goes through the SDMA engine on a stream that runs nothing else -- measured on this box: ~100 us
  // per 600 KB frame, in the CALL when the queue is full -- and through a blit kernel only on a stream that is busy with kernels.)
  const size_t bytes = (size_t)g->w * g->h * sizeof(int16_t);
  void* mapped = nullptr;
  if (hipHostGetDevicePointer(&mapped, (void*)host, 0) == hipSuccess && mapped && (bytes % 16) == 0 && ((uintptr_t)mapped % 16) == 0) {
    const size_t n = bytes / 16;
    if (g->calibType >= 0) stage_copy_kernel<true><<<kStageWorkgroups, 256, 0, g->copy>>>((const uint4*)mapped, (uint4*)g->buf[b], n, (float4*)g->depth[b], g->calibType, g->c0, g->c1, g->fx);
    else stage_copy_kernel<false><<<kStageWorkgroups, 256, 0, g->copy>>>((const uint4*)mapped, (uint4*)g->buf[b], n, nullptr, 0, 0.0f, 0.0f, 0.0f);
    ITM_LAUNCH_CHECK();
  } else {
    (void)hipGetLastError();                  // pageable memory: the runtime's staged copy
    ITM_HIP(hipMemcpyAsync(g->buf[b], host, bytes, hipMemcpyHostToDevice, g->copy));
    if (g->calibType >= 0) {
      const int rc = g->calibType == 0 ? itm_convert_disparity(g->buf[b], g->depth[b], g->w, g->h, g->c0, g->c1, g->fx, (itm_stream)g->copy)
                                       : itm_convert_depth_affine(g->buf[b], g->depth[b], g->w, g->h, g->c0, g->c1, (itm_stream)g->copy);
      if (rc) return rc;
    }
  }
  ITM_HIP(hipEventRecord(g->uploaded[b], g->copy));
  g->tail.store(tail + 1, std::memory_order_release);
  return ITM_OK;
}

// How many uploaded frames have not been acquired yet, and whether the copy stream still has to READ a host buffer: *busy == 0 means
// every pinned buffer handed to itm_depth_stager_upload so far may be rewritten (hipEventQuery of the newest upload, no waiting).
int itm_depth_stager_pending(itm_depth_stager* g, int* waiting, int* busy) {
  if (!g) return set_error(ITM_ERR_INVALID, "null argument");
  const unsigned long long tail = g->tail.load(std::memory_order_acquire), head = g->head.load(std::memory_order_acquire);
  if (waiting) *waiting = (int)(tail - head) - (g->held ? 1 : 0);
  if (busy) {
    *busy = 0;
    if (tail > 0) {
      const hipError_t q = hipEventQuery(g->uploaded[(int)((tail - 1) % (unsigned long long)g->slots)]);
      if (q == hipErrorNotReady) *busy = 1;
      else if (q != hipSuccess) return hip_fail(q, "hipEventQuery(upload)", __FILE__, __LINE__);
    }
  }
  return ITM_OK;
}

// From the next upload on the copy also CONVERTS: every slot gets a float image, written by the copy kernel itself with the operations of
// itm_update_view's first step (calibType 0: disparity, c0 / c1 / fx; 1: affine, c0 * raw + c1).  For a view without bilateral filter and
// noise model that image IS the view's depth: no conversion launch on the frame's stream.  Call before the first upload (or with the ring
// empty): frames already uploaded have no float image.
int itm_depth_stager_set_conversion(itm_depth_stager* g, int calibType, float c0, float c1, float fx) {
  if (!g || (calibType != 0 && calibType != 1)) return set_error(ITM_ERR_INVALID, "bad argument");
  if (g->tail.load(std::memory_order_acquire) != g->head.load(std::memory_order_acquire)) return set_error(ITM_ERR_INVALID, "frames are waiting in the ring");
  if (g->depth.empty()) {
    // all slots or none: a partly filled vector would pass for "allocated" on a retry, and the converting copy would write through a null slot
    std::vector<float*> imgs((size_t)g->slots, nullptr);
    for (int i = 0; i < g->slots; ++i) {
      const hipError_t e = hipMalloc((void**)&imgs[i], (size_t)g->w * g->h * sizeof(float));
      if (e != hipSuccess) {
        for (float* q : imgs) if (q) (void)hipFree(q);
        return hip_fail(e, "hipMalloc(stager float images)", __FILE__, __LINE__);
      }
    }
    g->depth = imgs;
  }
  g->calibType = calibType; g->c0 = c0; g->c1 = c1; g->fx = fx;
  return ITM_OK;
}

static int acquire_slot(itm_depth_stager* g, itm_stream stream, int* slot);
int itm_depth_stager_acquire_depth(itm_depth_stager* g, itm_stream stream, const int16_t** raw, const float** depth) {
  if (!g || !depth) return set_error(ITM_ERR_INVALID, "null argument");
  if (g->calibType < 0) return set_error(ITM_ERR_INVALID, "no conversion has been set (itm_depth_stager_set_conversion)");
  int b = 0;
  const int rc = acquire_slot(g, stream, &b);
  if (rc) return rc;
  if (raw) *raw = g->buf[b];
  *depth = g->depth[b];
  return ITM_OK;
}

int itm_depth_stager_acquire(itm_depth_stager* g, itm_stream stream, const int16_t** dev) {
  if (!g || !dev) return set_error(ITM_ERR_INVALID, "null argument");
  int b = 0;
  const int rc = acquire_slot(g, stream, &b);
  if (rc) return rc;
  *dev = g->buf[b];
  return ITM_OK;
}

static int acquire_slot(itm_depth_stager* g, itm_stream stream, int* slot) {
  if (g->held) return set_error(ITM_ERR_INVALID, "the previous frame has not been released");
  const unsigned long long head = g->head.load(std::memory_order_relaxed);
  if (head == g->tail.load(std::memory_order_acquire)) return set_error(ITM_ERR_INVALID, "no uploaded frame is waiting");
  const int b = (int)(head % (unsigned long long)g->slots);
  // with the uploads a frame ahead the copy has normally finished: then the frame's stream needs no dependency on the copy stream at all
  // (a cross-stream wait is ~3-4 us of stream time even when it is satisfied at once); a query is a host-side load
  // A host that runs ahead of the device gets here while the copy is still in flight (it was launched one submission ago and takes
  // ~25 us): it polls for a moment -- the device has frames queued, the host's wait costs it nothing -- rather than put a wait packet
  // on the frame's stream (5-6 us between icp_maps and the next frame's first kernel in rocprofv3's trace).  A copy that takes longer
  // than that (its slot's previous reader has not run yet) is left to the stream.
  hipError_t q = hipEventQuery(g->uploaded[b]);
  if (q == hipErrorNotReady) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; q == hipErrorNotReady; ++spins) {
      if ((spins & 0x3fu) == 0x3fu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 150e-6) break;
      __builtin_ia32_pause();
      q = hipEventQuery(g->uploaded[b]);
    }
    if (q == hipErrorNotReady) { ITM_HIP(hipStreamWaitEvent(as_stream(stream), g->uploaded[b], 0)); q = hipSuccess; }
  }
  if (q != hipSuccess) return hip_fail(q, "hipEventQuery(upload)", __FILE__, __LINE__);
  *slot = b;
  g->held = true;
  return ITM_OK;
}

int itm_depth_stager_release(itm_depth_stager* g, itm_stream stream) {
  if (!g) return set_error(ITM_ERR_INVALID, "null argument");
  if (!g->held) return set_error(ITM_ERR_INVALID, "no frame is held");
  const unsigned long long head = g->head.load(std::memory_order_relaxed);
  const int b = (int)(head % (unsigned long long)g->slots);
  // "everything submitted so far read it": engine calls that were only RECORDED with this slot's image as their view's depth (pending.hip:
  // a frame whose CreateICPMaps has not come) are launched first, so that the event below lies behind them
  {
    int rc = flush_overlapping(g->buf[b], (size_t)g->w * g->h * sizeof(int16_t), as_stream(stream));
    if (!rc && !g->depth.empty()) rc = flush_overlapping(g->depth[b], (size_t)g->w * g->h * sizeof(float), as_stream(stream));
    if (rc) return rc;
  }
  ITM_HIP(hipEventRecord(g->consumed[b], as_stream(stream)));
  g->consumedRecorded[b].store(1, std::memory_order_release);
  g->held = false;
  g->head.store(head + 1, std::memory_order_release);
  return ITM_OK;
}

}  // extern "C"
