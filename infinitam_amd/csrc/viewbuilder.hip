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

__global__ void __launch_bounds__(256) filter_depth_kernel(const float* __restrict__ in, float* __restrict__ out, int w, int h) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (lane & 15);
  const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
  if (x >= w || y >= h) return;
  float result = 0.0f;                                   // image_out->Clear()
  if (x >= 2 && x < w - 2 && y >= 2 && y < h - 2) {
    const float z = in[x + y * w];
    if (z < 0.0f) result = -1.0f;
    else {
      const float dzq = (z - 0.4f);
      const float sigma_z = 1.0f / (0.0012f + 0.0019f * dzq * dzq + 0.0001f / sqrtf(z) * 0.25f);
      const float msl = 1.2232f;
      float final_depth = 0.0f, w_sum = 0.0f;
      for (int i = -2; i <= 2; ++i)
        for (int j = -2; j <= 2; ++j) {
          const float tmpz = in[(x + j) + (y + i) * w];
          if (tmpz < 0.0f) continue;
          float dz = (tmpz - z); dz *= dz;
          const int a = (i < 0 ? -i : i) + (j < 0 ? -j : j);
          const float wgt = expf(-0.5f * ((float)a * msl * msl + dz * sigma_z * sigma_z));
          w_sum += wgt;
          final_depth += wgt * tmpz;
        }
      result = final_depth / w_sum;
    }
  }
  out[x + y * w] = result;
}

__global__ void __launch_bounds__(256) normal_weight_kernel(const float* __restrict__ depth, float4* __restrict__ normals, float* __restrict__ sigmaZ,
                                                            int w, int h, float ix, float iy, float iz, float iw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (lane & 15);
  const int y = blockIdx.y * 16 + wave * 4 + (lane >> 4);
  if (x < 2 || x >= w - 2 || y < 2 || y >= h - 2) return;     // the border is never written by the reference
  const int idx = x + y * w;
  const float z = depth[idx];
  // the reference only sets normal.w / sigmaZ for rejected pixels; the other components keep their old value
  if (z < 0.0f) { normals[idx].w = -1.0f; sigmaZ[idx] = -1.0f; return; }
  const float zxp = depth[(x + 1) + y * w], zyp = depth[x + (y + 1) * w], zxm = depth[(x - 1) + y * w], zym = depth[x + (y - 1) * w];
  if (zxp <= 0 || zyp <= 0 || zxm <= 0 || zym <= 0) { normals[idx].w = -1.0f; sigmaZ[idx] = -1.0f; return; }
  const float fx_ = (float)x, fy_ = (float)y;
  const float xp1x = zxp * ((fx_ + 1.0f) - iz) * ix, xp1y = zxp * (fy_ - iw) * iy;
  const float xm1x = zxm * ((fx_ - 1.0f) - iz) * ix, xm1y = zxm * (fy_ - iw) * iy;
  const float yp1x = zyp * (fx_ - iz) * ix, yp1y = zyp * ((fy_ + 1.0f) - iw) * iy;
  const float ym1x = zym * (fx_ - iz) * ix, ym1y = zym * ((fy_ - 1.0f) - iw) * iy;
  const float dxx = xp1x - xm1x, dxy = xp1y - xm1y, dxz = zxp - zxm;
  const float dyx = yp1x - ym1x, dyy = yp1y - ym1y, dyz = zyp - zym;
  float nx = (dxy * dyz - dxz * dyy);
  float ny = (dxz * dyx - dxx * dyz);
  float nz = (dxx * dyy - dxy * dyx);
  if (nx == 0.0f && ny == 0 && nz == 0) { normals[idx].w = -1.0f; sigmaZ[idx] = -1.0f; return; }
  const float norm = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
  nx *= norm; ny *= norm; nz *= norm;
  normals[idx] = make_float4(nx, ny, nz, 1.0f);
  const float PIf = 3.1415926535897932384626433832795f;
  const float theta = acosf(nz);
  const float theta_diff = theta / (PIf * 0.5f - theta);
  const float dzq = (z - 0.4f);
  sigmaZ[idx] = (0.0012f + 0.0019f * dzq * dzq + 0.0001f / sqrtf(z) * theta_diff * theta_diff);
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
