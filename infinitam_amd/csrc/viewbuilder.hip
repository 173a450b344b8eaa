// viewbuilder.hip -- the device part of the view builder that precedes the path: 5x5 bilateral depth
// filter, normal + depth-uncertainty images, and the UpdateView sequence for a raw frame resident in HBM.
//
// Reference behaviour:
//   filterDepth / computeNormalAndWeight   DeviceAgnostic/ITMViewBuilder.h:30-117
//   DepthFiltering / ComputeNormalAndWeights / UpdateView   DeviceSpecific/CPU/ITMViewBuilder_CPU.cpp:14-63, 119-145
// Arithmetic follows the reference operation for operation (no contraction); the only non-IEEE functions are
// exp (filter weights) and acos (uncertainty), for which the device library and the host libm may differ in the last
// place -- parity tolerance 1e-6 relative for the filtered depth, exact for everything else that does not depend on them.
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
