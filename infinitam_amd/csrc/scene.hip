// scene.hip -- scene / render-state lifetime, ResetScene, state transfer and the small helpers of
// the C-ABI (include/itm_hip.h).  Reference behaviour restated:
//   ResetScene               DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:24-45, :301-312
//   ITMScene / ITMLocalVBA   Objects/ITMScene.h:37-43, Objects/ITMLocalVBA.h:40-48
//   ITMRenderState(_VH)      Objects/ITMRenderState.h:51-75, Objects/ITMRenderState_VH.h:38-47
//   view builder conversions DeviceAgnostic/ITMViewBuilder.h:7-28
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <new>

#include "itm_internal.h"

namespace itm {

static thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
  return ITM_ERR_DEVICE;
}

hipEvent_t Profiler::get() {
  if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  // device-scope release: the default system-scope release of an event record writes back the L2s, which both lengthens the
  // bracketed interval and cools the caches of the launches that follow (HIP: "useful to obtain more precise timings")
  (void)hipEventCreateWithFlags(&e, hipEventReleaseToDevice);
  return e;
}
void Profiler::flush() {
  for (const Rec& r : pending) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.id] += ms; calls[r.id] += 1; }
    pool.push_back(r.a); pool.push_back(r.b);
  }
  pending.clear();
}

// ---- host matrix helpers ---------------------------------------------------------------------
// Matrix4::inv (ORUtils/Matrix.h:162-223): cofactors of the transposed matrix, then every element
// times 1/det.  Host code in this file is compiled with -ffp-contract=off as well.
static inline float tri(float a, float b, float c, float d, float e, float f) { return a * b + c * d + e * f; }

bool invert4(const float* m, float* o) {
  float s[16], t[12];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s[i + 4 * j] = m[i * 4 + j];
  t[0] = s[10] * s[15]; t[1] = s[11] * s[14]; t[2] = s[9] * s[15];  t[3] = s[11] * s[13];
  t[4] = s[9] * s[14];  t[5] = s[10] * s[13]; t[6] = s[8] * s[15];  t[7] = s[11] * s[12];
  t[8] = s[8] * s[14];  t[9] = s[10] * s[12]; t[10] = s[8] * s[13]; t[11] = s[9] * s[12];
  o[0] = tri(t[0], s[5], t[3], s[6], t[4], s[7]) - tri(t[1], s[5], t[2], s[6], t[5], s[7]);
  o[1] = tri(t[1], s[4], t[6], s[6], t[9], s[7]) - tri(t[0], s[4], t[7], s[6], t[8], s[7]);
  o[2] = tri(t[2], s[4], t[7], s[5], t[10], s[7]) - tri(t[3], s[4], t[6], s[5], t[11], s[7]);
  o[3] = tri(t[5], s[4], t[8], s[5], t[11], s[6]) - tri(t[4], s[4], t[9], s[5], t[10], s[6]);
  float det = s[0] * o[0] + s[1] * o[1] + s[2] * o[2] + s[3] * o[3];
  if (det == 0.0f) return false;
  o[4] = tri(t[1], s[1], t[2], s[2], t[5], s[3]) - tri(t[0], s[1], t[3], s[2], t[4], s[3]);
  o[5] = tri(t[0], s[0], t[7], s[2], t[8], s[3]) - tri(t[1], s[0], t[6], s[2], t[9], s[3]);
  o[6] = tri(t[3], s[0], t[6], s[1], t[11], s[3]) - tri(t[2], s[0], t[7], s[1], t[10], s[3]);
  o[7] = tri(t[4], s[0], t[9], s[1], t[10], s[2]) - tri(t[5], s[0], t[8], s[1], t[11], s[2]);
  t[0] = s[2] * s[7]; t[1] = s[3] * s[6]; t[2] = s[1] * s[7];  t[3] = s[3] * s[5];
  t[4] = s[1] * s[6]; t[5] = s[2] * s[5]; t[6] = s[0] * s[7];  t[7] = s[3] * s[4];
  t[8] = s[0] * s[6]; t[9] = s[2] * s[4]; t[10] = s[0] * s[5]; t[11] = s[1] * s[4];
  o[8] = tri(t[0], s[13], t[3], s[14], t[4], s[15]) - tri(t[1], s[13], t[2], s[14], t[5], s[15]);
  o[9] = tri(t[1], s[12], t[6], s[14], t[9], s[15]) - tri(t[0], s[12], t[7], s[14], t[8], s[15]);
  o[10] = tri(t[2], s[12], t[7], s[13], t[10], s[15]) - tri(t[3], s[12], t[6], s[13], t[11], s[15]);
  o[11] = tri(t[5], s[12], t[8], s[13], t[11], s[14]) - tri(t[4], s[12], t[9], s[13], t[10], s[14]);
  o[12] = tri(t[2], s[10], t[5], s[11], t[1], s[9]) - tri(t[4], s[11], t[0], s[9], t[3], s[10]);
  o[13] = tri(t[8], s[11], t[0], s[8], t[7], s[10]) - tri(t[6], s[10], t[9], s[11], t[1], s[8]);
  o[14] = tri(t[6], s[9], t[11], s[11], t[3], s[8]) - tri(t[10], s[11], t[2], s[8], t[7], s[9]);
  o[15] = tri(t[10], s[10], t[4], s[8], t[9], s[9]) - tri(t[8], s[9], t[11], s[10], t[5], s[8]);
  float rdet = 1 / det;
  for (int i = 0; i < 16; ++i) o[i] *= rdet;
  return true;
}

// Matrix4 * Matrix4 (ORUtils/Matrix.h:102-108): element (col,row) accumulated from zero over k.
void matmul4(const float* lhs, const float* rhs, float* out) {
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += lhs[k * 4 + row] * rhs[col * 4 + k];
      out[col * 4 + row] = acc;
    }
}

// ---- fill kernels ----------------------------------------------------------------------------
template <class VX>
__global__ void __launch_bounds__(256) reset_voxels_kernel(void* vba, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  typename VX::Reg init = VX::init();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) VX::store(vba, i, init);
}

// VoxelS volumes are the large ones (dense 512^3): write 16 bytes per lane.
__global__ void __launch_bounds__(256) reset_voxels_s_x4_kernel(uint4* vba, size_t n4) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint4 v = make_uint4(32767u, 32767u, 32767u, 32767u);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) vba[i] = v;
}

__global__ void __launch_bounds__(256) reset_hash_kernel(uint4* hash, int nEntries, int32_t* excessList, int nExcess,
                                                         int32_t* allocList, int nBlocks, uint32_t* allocKey, uint32_t* headBits, int nHeadWords,
                                                         int32_t* chunkReq, int nChunkReq, SceneCounters* counters) {
  int stride = gridDim.x * blockDim.x;
  int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const uint4 empty = pack_entry(0, 0, 0, 0, -2);
  for (int i = i0; i < nEntries; i += stride) { hash[i] = empty; allocKey[i] = 0u; }
  for (int i = i0; i < nHeadWords; i += stride) headBits[i] = 0u;
  for (int i = i0; i < nExcess; i += stride) excessList[i] = i;
  for (int i = i0; i < nBlocks; i += stride) allocList[i] = i;
  for (int i = i0; i < nChunkReq; i += stride) chunkReq[i] = 0;
  if (i0 == 0) {
    counters->lastFreeBlockId = nBlocks - 1;
    counters->lastFreeExcessListId = nExcess - 1;
    counters->noAllocRequests = 0;
    counters->statusFlags = 0;
  }
}

// rebuilds the occupancy bitmap from the table (after an upload of hash entries)
__global__ void __launch_bounds__(256) head_bits_kernel(const uint4* __restrict__ hash, uint32_t* __restrict__ headBits, int nWords, int bucketNum) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nWords) return;
  uint32_t bits = 0;
  for (int k = 0; k < 32; ++k) {
    const int slot = w * 32 + k;
    if (slot < bucketNum && (int)hash[slot].w >= -1) bits |= 1u << k;      // -1: swapped out, its chain may still hold resident blocks
  }
  headBits[w] = bits;
}

// ---- the acceleration cubes (block directory, slot directory, sdf mirror; itm_types.h) -------------------------------------------
// Invariant: the only non-empty cells are those of table entries with ptr >= 0, at the scene's current origin.  So the cubes are
// emptied by visiting exactly those entries (unfill) -- before the table is reset or replaced, or before the origin moves -- and
// filled again from the table afterwards: O(allocated blocks) instead of an 18 GB memset per ResetScene / upload / move.

// FILL: records every entry with ptr >= 0 in the directory cubes; !FILL: empties those cells
template <bool FILL>
__global__ void __launch_bounds__(256) directory_fill_kernel(const uint4* __restrict__ hash, int nEntries, int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, AccelOrigin org) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nEntries) return;
  const HashEntry e = unpack_entry(hash[i]);
  if (e.ptr < 0) return;
  directory_insert(dirPtr, dirSlot, org, e.px, e.py, e.pz, FILL ? e.ptr : -1, FILL ? i : -1);
}

// FILL: the 512 sdf values of every allocated block inside the mirror cube, from the pool; !FILL: "no block here" in those cells.
// One workgroup per entry.
template <class VX, bool FILL>
__global__ void __launch_bounds__(256) mirror_fill_kernel(const uint4* __restrict__ hash, int nEntries, const void* __restrict__ vba, void* __restrict__ mirror, AccelOrigin org, size_t numVoxels) {
  using MC = MirrorCodec<VX::kShort>;
  for (int i = blockIdx.x; i < nEntries; i += gridDim.x) {
    const HashEntry e = unpack_entry(hash[i]);
    if (e.ptr < 0) continue;
    if (FILL && (size_t)e.ptr * kBlockVoxels + kBlockVoxels > numVoxels) continue;      // an uploaded table may hold anything
    // (FILL maps the block's page if need be: one thread asks, the workgroup hears the answer)
    __shared__ size_t baseShared; __shared__ int okShared;
    __syncthreads();
    if (threadIdx.x == 0) { size_t b0 = 0; okShared = mirror_block_base<FILL>(org, e.px, e.py, e.pz, b0) ? 1 : 0; baseShared = b0; }
    __syncthreads();
    if (!okShared) continue;
    const size_t base = baseShared;
    for (int t = threadIdx.x; t < kBlockVoxels; t += 256) {
      typename MC::T v;
      if constexpr (FILL) v = MC::of(VX::load_raw_sdf(vba, (size_t)e.ptr * kBlockVoxels + t));
      else if constexpr (VX::kShort) v = (typename MC::T)-32768;
      else v = (typename MC::T)0xffffffffu;
      ((typename MC::T*)mirror)[base + mirror_block_lin((uint32_t)t)] = v;
    }
  }
}

#ifndef ITM_MIRROR_FLOAT_TYPES
#define ITM_MIRROR_FLOAT_TYPES 0
#endif
static bool mirror_is_float(const itm_scene* s) { return s->cfg.voxelType == ITM_VOXEL_F || s->cfg.voxelType == ITM_VOXEL_F_RGB; }
// side of the mirror's cube in blocks, and half of it: the DENSE form's is chosen per scene (itm_types.h, mirror_dense_bits)
static int mirror_side(const itm_scene* s) { return s->org.mMaxPages < 0 ? (1 << mirror_dense_bits(s->org)) : kMirrorSide; }
static size_t mirror_dense_cells(int bits) { return (size_t)1 << (3 * bits); }
// every voxel of every page of the pool "no block here" (-32768 per short, all ones per float), no page handed out (scene creation only)
static hipError_t mirror_clear(itm_scene* s, hipStream_t st) {
  const size_t voxels = s->org.mMaxPages < 0 ? mirror_dense_cells(mirror_dense_bits(s->org)) * 512 : (size_t)s->mirrorPages * kPageBlocks * 512;
  hipError_t e = mirror_is_float(s) ? hipMemsetAsync(s->sdfMirror, 0xff, voxels * 4, st) : hipMemsetD16Async((unsigned short*)s->sdfMirror, (unsigned short)0x8000, voxels, st);
  if (e == hipSuccess && s->org.mTable) e = hipMemsetAsync(s->org.mTable, 0xff, kMirrorTableCells * 4, st);
  if (e == hipSuccess && s->org.mPages) e = hipMemsetAsync(s->org.mPages, 0, 4, st);
  return e;
}

template <bool FILL>
static int mirror_pass(itm_scene* s, hipStream_t st) {
  if (!s->sdfMirror) return ITM_OK;
  int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    mirror_fill_kernel<VX, FILL><<<4096, 256, 0, st>>>(s->hash, s->noTotalEntries, s->vba, s->sdfMirror, s->org, s->numVoxels);
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  if (!FILL && s->org.mTable) {
    // every mapped page is all "absent" again (the invariant of itm_types.h): the pool is as good as new
    ITM_HIP(hipMemsetAsync(s->org.mTable, 0xff, kMirrorTableCells * 4, st));
    ITM_HIP(hipMemsetAsync(s->org.mPages, 0, 4, st));
  }
  return ITM_OK;
}
template <bool FILL>
static int directory_pass(itm_scene* s, hipStream_t st) {
  if (!s->dirPtr) return ITM_OK;
  directory_fill_kernel<FILL><<<(s->noTotalEntries + 255) / 256, 256, 0, st>>>(s->hash, s->noTotalEntries, s->dirPtr, s->dirSlot, s->org);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

// the mirror's values again from the pool (the voxels were replaced, the table was not)
int rebuild_sdf_mirror(itm_scene* s, hipStream_t st) { return mirror_pass<true>(s, st); }

// empties the cubes: to be called while the table still holds the entries that filled them
int accel_unfill(itm_scene* s, hipStream_t st) {
  int rc = directory_pass<false>(s, st);
  if (rc) return rc;
  return mirror_pass<false>(s, st);
}

// occupancy bitmap AND both cubes, from the table (after the table was replaced: accel_unfill ran before the replacement)
int rebuild_head_bits(itm_scene* s, hipStream_t st) {
  if (!s->headBits) return ITM_OK;
  const int nWords = (s->cfg.bucketNum + 31) / 32;
  head_bits_kernel<<<(nWords + 255) / 256, 256, 0, st>>>(s->hash, s->headBits, nWords, s->cfg.bucketNum);
  ITM_LAUNCH_CHECK();
  int rc = directory_pass<true>(s, st);
  if (rc) return rc;
  return mirror_pass<true>(s, st);
}

// Where the cubes should lie for a camera at block `cb` looking along `dir`, reaching `reach` blocks: the frustum lies inside
// the ball of radius 0.7 reach around the point half a reach ahead of the camera.
static inline int round4(double v) { return (int)floor(v / 4.0 + 0.5) * 4; }
static bool cube_keeps(const int org[3], int half, const double m[3], double rho) {
  // the ball (m, rho) inside the cube -- or, for a cube smaller than the ball, its centre within a quarter of the cube's side of m
  const double slack = rho < half ? half - rho : half * 0.5;
  for (int k = 0; k < 3; ++k) if (fabs(m[k] - (org[k] + half)) > slack) return false;
  return true;
}

// Places (first frame) or re-places the cubes around the view of `invM` (camera -> world).  Moving a cube empties and refills
// it from the table on `st`: two passes over the table, a kilobyte per allocated block each.
int accel_place(itm_scene* s, const float* invM, hipStream_t st) {
  if (!s->dirPtr && !s->sdfMirror) return ITM_OK;
  const double bs = (double)s->prm.voxelSize * kBlockSide;
  if (!(bs > 0.0)) return ITM_OK;
  double cb[3], dir[3], len = 0.0;
  for (int k = 0; k < 3; ++k) { cb[k] = (double)invM[12 + k] / bs; dir[k] = invM[8 + k]; len += dir[k] * dir[k]; }
  len = sqrt(len);
  if (!(len > 0.0) || !std::isfinite(cb[0]) || !std::isfinite(cb[1]) || !std::isfinite(cb[2])) return ITM_OK;
  const double reach = fmin((double)s->prm.viewFrustum_max / bs + 2.0, 30000.0);
  double m[3];
  for (int k = 0; k < 3; ++k) m[k] = cb[k] + dir[k] / len * reach * 0.5;
  const double rho = 0.7 * reach;
  AccelOrigin o = s->org;
  const int dOrg[3] = {o.dx, o.dy, o.dz}, mOrg[3] = {o.mx, o.my, o.mz};
  const bool moveDir = !s->orgPlaced || !cube_keeps(dOrg, kDirHalf, m, rho);
  const int mHalf = mirror_side(s) / 2;
  const bool moveMir = !s->orgPlaced || !cube_keeps(mOrg, mHalf, m, rho);
  if (!moveDir && !moveMir) return ITM_OK;
  // a cube smaller than the view is centred further towards the camera (the near part of the frustum holds most rays' steps)
  auto centre = [&](int half, int k) { const double ahead = fmin(reach * 0.5, (double)half * 0.5); return round4(cb[k] + dir[k] / len * ahead); };
  if (moveDir) { o.dx = centre(kDirHalf, 0) - kDirHalf; o.dy = centre(kDirHalf, 1) - kDirHalf; o.dz = centre(kDirHalf, 2) - kDirHalf; }
  if (moveMir) { o.mx = centre(mHalf, 0) - mHalf; o.my = centre(mHalf, 1) - mHalf; o.mz = centre(mHalf, 2) - mHalf; }
  const bool first = !s->orgPlaced;
  s->orgPlaced = true;
  if (first) { s->org = o; return ITM_OK; }       // nothing allocated yet: the cubes are empty wherever they lie
  int rc = ITM_OK;
  if (moveDir) rc = directory_pass<false>(s, st);
  if (!rc && moveMir) rc = mirror_pass<false>(s, st);
  if (rc) return rc;
  s->org = o;
  ++s->accelMoves;
  if (moveDir) rc = directory_pass<true>(s, st);
  if (!rc && moveMir) rc = mirror_pass<true>(s, st);
  return rc;
}

// after an upload of the table: the cubes around the centre of the allocated blocks of `entries` (host copy of what was uploaded)
static void accel_place_for_table(itm_scene* s, const HashEntry* entries, size_t n) {
  long long lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
  bool any = false;
  for (size_t i = 0; i < n; ++i) {
    if (entries[i].ptr < 0) continue;
    const int c[3] = {entries[i].px, entries[i].py, entries[i].pz};
    for (int k = 0; k < 3; ++k) { if (c[k] < lo[k]) lo[k] = c[k]; if (c[k] > hi[k]) hi[k] = c[k]; }
    any = true;
  }
  if (!any) { s->orgPlaced = false; return; }
  const int c[3] = {round4((lo[0] + hi[0]) * 0.5), round4((lo[1] + hi[1]) * 0.5), round4((lo[2] + hi[2]) * 0.5)};
  s->org.dx = c[0] - kDirHalf; s->org.dy = c[1] - kDirHalf; s->org.dz = c[2] - kDirHalf;
  { const int mHalf = mirror_side(s) / 2; s->org.mx = c[0] - mHalf; s->org.my = c[1] - mHalf; s->org.mz = c[2] - mHalf; }
  s->orgPlaced = true;
}

__global__ void reset_dense_kernel(int32_t* allocList, SceneCounters* counters) {
  allocList[0] = 0;
  counters->lastFreeBlockId = 0;
  counters->lastFreeExcessListId = 0;
  counters->noAllocRequests = 0;
  counters->statusFlags = 0;
}

__global__ void __launch_bounds__(256) fill_range_kernel(float2* img, int n, float a, float b) {
  int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) img[i] = make_float2(a, b);
}

// convertDepthAffineToFloat (DeviceAgnostic/ITMViewBuilder.h:22-28)
__global__ void __launch_bounds__(256) depth_affine_kernel(const int16_t* raw, float* out, int n, float a, float b) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int16_t d = raw[i];
  out[i] = ((d <= 0) || (d > 32000)) ? -1.0f : (float)d * a + b;
}
// convertDisparityToDepth (DeviceAgnostic/ITMViewBuilder.h:7-20)
__global__ void __launch_bounds__(256) depth_disparity_kernel(const int16_t* raw, float* out, int n, float c0, float c1, float fx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = c0 - (float)raw[i];
  float depth;
  if (t == 0) depth = 0.0f; else depth = 8.0f * c1 * fx / t;
  out[i] = (depth > 0) ? depth : -1.0f;
}

__global__ void __launch_bounds__(256) div32767_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = VoxelS::to_float(in[i]);
}
// mode 1: a/b through the shared-reciprocal chain; mode 2: a/b with b a small integer through the
// refined reciprocal + 3-instruction quotient; mode 3: the refined reciprocal itself; mode 4: a/b with a
// host-rounded reciprocal in r (division by a constant)
__global__ void __launch_bounds__(256) divide_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ r, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 1) out[i] = div_by_rcp(a[i], b[i], refined_rcp(b[i]));
  else if (mode == 2) out[i] = div_markstein(a[i], b[i], refined_rcp(b[i]));
  else if (mode == 3) out[i] = refined_rcp(b[i]);
  else out[i] = div_markstein(a[i], b[i], r[i]);
}

__global__ void __launch_bounds__(256) export_record_kernel(const int32_t* ids, const RenderCounters* rc, Mat4 M, int maxIds, int32_t* dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int nv = rc->noVisibleEntries;
  if (i < 16) dst[i] = __float_as_int(M.m[i]);
  if (i == 16) dst[16] = nv;
  if (i < maxIds) dst[17 + i] = (i < nv) ? ids[i] : -1;
}

static void* buffer_of(const itm_scene* s, const itm_render_state* rs, int which, size_t* bytes) {
  *bytes = 0;
  bool hashScene = s && s->cfg.indexType == ITM_INDEX_HASH;
  switch (which) {
    case ITM_BUF_HASH_ENTRIES: if (hashScene) { *bytes = (size_t)s->noTotalEntries * 16; return s->hash; } break;
    case ITM_BUF_EXCESS_LIST: if (hashScene) { *bytes = (size_t)s->cfg.excessNum * 4; return s->excessList; } break;
    case ITM_BUF_VOXEL_BLOCKS: if (s) { *bytes = s->numVoxels * s->voxBytes; return s->vba; } break;
    case ITM_BUF_ALLOCATION_LIST: if (s) { *bytes = (hashScene ? (size_t)s->cfg.localBlockNum : 1) * 4; return s->allocList; } break;
    case ITM_BUF_SWAP_STATES: if (hashScene && s->swapStates) { *bytes = (size_t)s->noTotalEntries; return s->swapStates; } break;
    default: break;
  }
  if (!rs) return nullptr;
  size_t P = (size_t)rs->w * rs->h;
  switch (which) {
    case ITM_BUF_VISIBLE_IDS: if (rs->hash) { *bytes = (size_t)rs->capIds * 4; return rs->visibleIds; } break;
    case ITM_BUF_VISIBLE_TYPE: if (rs->hash && rs->scene) { *bytes = (size_t)rs->scene->noTotalEntries; return rs->visibleType; } break;      // (an orphaned render state -- its scene was destroyed first -- no longer knows the table's size)
    case ITM_BUF_RANGE_IMAGE: *bytes = P * 8; return rs->range;
    case ITM_BUF_RAYCAST_RESULT: *bytes = P * 16; return rs->raycast;
    case ITM_BUF_RAYCAST_IMAGE: *bytes = P * 4; return rs->image;
    case ITM_BUF_FORWARD_PROJECTION: *bytes = P * 16; return rs->fwdProj;
    case ITM_BUF_MISSING_POINTS: *bytes = P * 4; return rs->missing;
    default: break;
  }
  return nullptr;
}

static std::atomic<int> g_liveHashScenes[64];
int live_hash_scenes(int device) { return (device >= 0 && device < 64) ? g_liveHashScenes[device].load() : 2; }

// Render states keep a back pointer to their scene and a scene keeps the list of its render states; either may be destroyed first
// (hosts with garbage-collected handles destroy in any order).  The list is touched at creation / destruction only.
static std::mutex g_rsRegistryMutex;

static void free_scene(itm_scene* s) {
  if (!s) return;
  if (s->deferredRs) forget_deferred(s->deferredRs);      // the scene they were recorded for is going away
  if (s->aheadRs) { s->aheadRs->ahead.valid = false; s->aheadRs = nullptr; }
  {
    // render states that outlive the scene: nothing recorded or announced on them survives it, and they no longer point at it
    std::lock_guard<std::mutex> lock(g_rsRegistryMutex);
    for (itm_render_state* r : s->renderStates) { forget_deferred(r); r->ahead.valid = false; r->scene = nullptr; }
    s->renderStates.clear();
  }
  if (s->fatalHost) (void)hipHostFree((void*)s->fatalHost);
  if (s->countedLive && s->device >= 0 && s->device < 64) g_liveHashScenes[s->device].fetch_sub(1);
  free_swap_state(s);
  if (s->prof) { s->prof->flush(); for (hipEvent_t e : s->prof->pool) (void)hipEventDestroy(e); delete s->prof; }
  (void)hipFree(s->hash); (void)hipFree(s->excessList); (void)hipFree(s->vba); (void)hipFree(s->allocList);
  (void)hipFree(s->counters); (void)hipFree(s->headBits); (void)hipFree(s->allocKey); (void)hipFree(s->chunkReq); (void)hipFree(s->chunkVis); (void)hipFree(s->chunkGran); (void)hipFree(s->chunkSweepDone); (void)hipFree(s->chunkKeptGran);
  (void)hipFree(s->dirPtr); (void)hipFree(s->dirSlot); (void)hipFree(s->sdfMirror); (void)hipFree(s->org.mTable); (void)hipFree(s->org.mPages); (void)hipFree(s->depthTiles);
  delete s;
}
static void free_rs(itm_render_state* r) {
  if (!r) return;
  // r->scene is null when the scene was destroyed first (free_scene above): then nothing is recorded on r any more and there is no
  // scene to tell.  Otherwise calls recorded on it still happen (pending.hip: the host made them, the reference would have executed
  // them; the recorded view's images must still be valid -- see itm_scene_set_deferred_fusion in the header).
  // (r->scene is read under the registry's lock: free_scene clears it under the same lock, possibly on another thread)
  bool haveScene;
  { std::lock_guard<std::mutex> lock(g_rsRegistryMutex); haveScene = r->scene != nullptr; }
  if (haveScene) (void)flush_deferred(r);
  {
    std::lock_guard<std::mutex> lock(g_rsRegistryMutex);
    itm_scene* s = const_cast<itm_scene*>(r->scene);
    if (s) {
      if (s->aheadRs == r) s->aheadRs = nullptr;
      for (size_t i = 0; i < s->renderStates.size(); ++i)
        if (s->renderStates[i] == r) { s->renderStates[i] = s->renderStates.back(); s->renderStates.pop_back(); break; }
    }
  }
  forget_deferred(r);
  (void)hipFree(r->range); (void)hipFree(r->raycast); (void)hipFree(r->fwdProj); (void)hipFree(r->missing);
  (void)hipFree(r->image); (void)hipFree(r->visibleIds); (void)hipFree(r->visibleType); (void)hipFree(r->counters);
  (void)hipFree(r->projBuf); (void)hipFree(r->rangePartials); (void)hipFree(r->pixScratch); (void)hipFree(r->pixChunk); (void)hipFree(r->viewFlags); (void)hipFree(r->viewChunkVis);
  if (r->sideStream) { (void)hipStreamSynchronize(r->sideStream); (void)hipStreamDestroy(r->sideStream); }
  if (r->listReady) (void)hipEventDestroy(r->listReady);
  if (r->projectionDone) (void)hipEventDestroy(r->projectionDone);
  delete r;
}

}  // namespace itm

using namespace itm;

extern "C" {

const char* itm_version(void) { return "itm-hip 0.1 (gfx950)"; }
const char* itm_last_error(void) { return g_last_error.c_str(); }
int itm_uses_device_memory(void) { return 1; }
size_t itm_voxel_size_bytes(int t) {
  switch (t) {
    case ITM_VOXEL_S: return VoxelS::kBytes;
    case ITM_VOXEL_F: return VoxelF::kBytes;
    case ITM_VOXEL_S_RGB: return VoxelSRgb::kBytes;
    case ITM_VOXEL_F_RGB: return VoxelFRgb::kBytes;
  }
  return 0;
}

int itm_dev_malloc(void** p, size_t n) {
  if (!p) return set_error(ITM_ERR_INVALID, "null pointer");
  ITM_HIP(hipMalloc(p, n ? n : 1));
  return ITM_OK;
}
int itm_dev_free(void* p) { ITM_HIP(hipFree(p)); return ITM_OK; }
// page-locked host memory, mapped into the device's address space: what asynchronous uploads (itm_depth_stager) need
int itm_host_malloc(void** p, size_t n) {
  if (!p) return set_error(ITM_ERR_INVALID, "null pointer");
  ITM_HIP(hipHostMalloc(p, n ? n : 1, hipHostMallocDefault));
  return ITM_OK;
}
int itm_host_free(void* p) { ITM_HIP(hipHostFree(p)); return ITM_OK; }
int itm_host_register(void* p, size_t n) {
  if (!p || !n) return set_error(ITM_ERR_INVALID, "null pointer / empty range");
  const hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return ITM_ALREADY_REGISTERED; }      // usable, but not the caller's to unregister
  if (e != hipSuccess) return hip_fail(e, "hipHostRegister", __FILE__, __LINE__);
  return ITM_OK;
}
int itm_host_unregister(void* p) {
  if (!p) return ITM_OK;
  const hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) { (void)hipGetLastError(); return set_error(ITM_ERR_INVALID, "the range was not registered"); }
  return ITM_OK;
}
int itm_memcpy_h2d(void* d, const void* s, size_t n, itm_stream st) {
  { const int rc = flush_overlapping(d, n, as_stream(st)); if (rc) return rc; }     // recorded engine calls that read the target (pending.hip)
  ITM_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, as_stream(st)));
  return ITM_OK;
}
int itm_memcpy_d2h(void* d, const void* s, size_t n, itm_stream st) {
  ITM_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, as_stream(st)));
  return ITM_OK;
}
int itm_stream_synchronize(itm_stream st) {
  { const int rc = flush_overlapping(nullptr, 0, as_stream(st)); if (rc) return rc; }   // engine calls recorded on this stream are launched first
  ITM_HIP(hipStreamSynchronize(as_stream(st)));
  return ITM_OK;
}
int itm_set_device(int d) { ITM_HIP(hipSetDevice(d)); return ITM_OK; }

int itm_scene_create(const itm_scene_config* cfg_in, const itm_scene_params* prm, itm_scene** out) {
  if (!cfg_in || !prm || !out) return set_error(ITM_ERR_INVALID, "null argument");
  itm_scene_config cfg = *cfg_in;
  if (cfg.bucketNum == 0) cfg.bucketNum = ITM_DEFAULT_BUCKET_NUM;
  if (cfg.excessNum == 0) cfg.excessNum = ITM_DEFAULT_EXCESS_NUM;
  if (cfg.localBlockNum == 0) cfg.localBlockNum = ITM_DEFAULT_LOCAL_BLOCK_NUM;
  if (cfg.denseSize[0] == 0 && cfg.denseSize[1] == 0 && cfg.denseSize[2] == 0) {
    cfg.denseSize[0] = cfg.denseSize[1] = cfg.denseSize[2] = 512;
    if (!cfg.denseOffsetSet) { cfg.denseOffset[0] = -256; cfg.denseOffset[1] = -256; cfg.denseOffset[2] = 0; }
  }
  cfg.denseOffsetSet = 1;
  if (cfg.maxRenderingBlocks == 0) cfg.maxRenderingBlocks = ITM_MAX_RENDERING_BLOCKS;
  if (cfg.maxRenderingBlocks < 0) return set_error(ITM_ERR_INVALID, "maxRenderingBlocks must be positive");
  size_t vb = itm_voxel_size_bytes(cfg.voxelType);
  if (!vb) return set_error(ITM_ERR_INVALID, "unknown voxel type");
  if (cfg.indexType != ITM_INDEX_HASH && cfg.indexType != ITM_INDEX_DENSE) return set_error(ITM_ERR_INVALID, "unknown index type");
  if (cfg.indexType == ITM_INDEX_HASH) {
    if (cfg.bucketNum <= 0 || (cfg.bucketNum & (cfg.bucketNum - 1))) return set_error(ITM_ERR_INVALID, "bucketNum must be a power of two");
    if (cfg.excessNum <= 0 || cfg.localBlockNum <= 0) return set_error(ITM_ERR_INVALID, "pool sizes must be positive");
    if ((cfg.bucketNum + cfg.excessNum) % 8 != 0) return set_error(ITM_ERR_INVALID, "bucketNum + excessNum must be a multiple of 8");
  } else {
    if (cfg.denseSize[0] <= 0 || cfg.denseSize[1] <= 0 || cfg.denseSize[2] <= 0) return set_error(ITM_ERR_INVALID, "dense size must be positive");
    // (the ray march indexes a dense volume with 32 bits, 0xffffffff = "no voxel": raycast_device.h dense_lin)
    if ((unsigned long long)cfg.denseSize[0] * (unsigned long long)cfg.denseSize[1] * (unsigned long long)cfg.denseSize[2] >= 0xffffffffull)
      return set_error(ITM_ERR_INVALID, "dense volume too large (2^32 - 1 voxels or more)");
  }
  if (prm->voxelSize <= 0 || prm->mu <= 0 || prm->maxW <= 0 || prm->maxW > 255) return set_error(ITM_ERR_INVALID, "bad scene parameters");

  itm_scene* s = new (std::nothrow) itm_scene();
  if (!s) return set_error(ITM_ERR_DEVICE, "out of host memory");
  s->cfg = cfg; s->prm = *prm; s->voxBytes = vb;
  (void)hipGetDevice(&s->device);
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
  if (cfg.indexType == ITM_INDEX_HASH) {
    s->noTotalEntries = cfg.bucketNum + cfg.excessNum;
    s->numChunks = (s->noTotalEntries + kSweepChunk - 1) / kSweepChunk;
    s->numVoxels = (size_t)cfg.localBlockNum * kBlockVoxels;
    alloc((void**)&s->hash, (size_t)s->noTotalEntries * 16);
    alloc((void**)&s->excessList, (size_t)cfg.excessNum * 4);
    alloc((void**)&s->allocList, (size_t)cfg.localBlockNum * 4);
    alloc((void**)&s->allocKey, (size_t)s->noTotalEntries * 4);
    alloc((void**)&s->headBits, (size_t)(cfg.bucketNum + 31) / 32 * 4);
    alloc((void**)&s->chunkReq, (size_t)s->numChunks * 2 * 2 * 4);
    alloc((void**)&s->chunkVis, (size_t)s->numChunks * 4);
    alloc((void**)&s->chunkGran, (size_t)s->numChunks * 8);
    alloc((void**)&s->chunkSweepDone, (size_t)s->numChunks * 4);
    alloc((void**)&s->chunkKeptGran, (size_t)s->numChunks * (kSweepChunk / 32) * 8);
  } else {
    s->numVoxels = (size_t)cfg.denseSize[0] * cfg.denseSize[1] * cfg.denseSize[2];
    alloc((void**)&s->allocList, 4);
  }
  alloc(&s->vba, s->numVoxels * vb + 16);
  alloc((void**)&s->counters, sizeof(SceneCounters));
  if (e != hipSuccess) { free_scene(s); return hip_fail(e, "hipMalloc(scene)", __FILE__, __LINE__); }
  // The directories are accelerators like the mirror below: a device without a gigabyte to spare runs without them (every look-up then
  // walks the table, as the reference does) instead of failing to create the scene
  if (cfg.indexType == ITM_INDEX_HASH && !getenv("ITM_NO_ACCELERATION_CUBES")) {
    if (hipMalloc((void**)&s->dirPtr, kDirCells * 4) != hipSuccess) { s->dirPtr = nullptr; (void)hipGetLastError(); }
    else if (hipMalloc((void**)&s->dirSlot, kDirCells * 4) != hipSuccess) { (void)hipFree(s->dirPtr); s->dirPtr = nullptr; s->dirSlot = nullptr; (void)hipGetLastError(); }
  }
  // The sdf mirror is an accelerator, in one of two forms (itm_types.h): DENSE -- the whole cube of 256^3 blocks, 17 GB, the faster one
  // (ray cast 38.3 us against 42.8-43.4 on BASELINE configs[1]) -- while the device has three times that to spare, i.e. for the first
  // handful of scenes of a 288 GB device; PAGED -- a 768 MB pool of 4 MB pages behind a 16 KB table, O(touched pages) -- for every
  // scene after that, and whenever ITM_MIRROR=paged is in the environment (ITM_MIRROR=dense insists on the cube, =off on none;
  // ITM_MIRROR_PAGES sizes the pool).  Silently left out when not even the pool fits -- and only for the short voxel types: for the
  // float types it was measured on BASELINE configs[4], the ray cast gains less than the integration pays for the extra 4-byte stores
  // (ITM_MIRROR_FLOAT_TYPES=1 builds it in).
  if (cfg.indexType == ITM_INDEX_HASH && s->dirPtr && !g_debug_no_sdf_mirror && (ITM_MIRROR_FLOAT_TYPES || !mirror_is_float(s))) {
    const char* mode = getenv("ITM_MIRROR");
    const size_t voxBytes = mirror_is_float(s) ? 4 : 2;
    // the dense cube's side: the smallest power of two of blocks that is at least 1.25 x the view's reach (viewFrustum_max in blocks + 2:
    // accel_place keeps the frustum's ball inside, or the cube's centre within a quarter of its side of the ball's) -- 128 for the
    // reference's 3 m frustum at 4 mm voxels (2.1 GB), 256 at 2 mm (17 GB); ITM_MIRROR_BITS = 6 .. 8 overrides (measurements)
    int denseBits = kMirrorBits;
    {
      const double bs = (double)s->prm.voxelSize * kBlockSide;
      const double reach = bs > 0.0 ? (double)s->prm.viewFrustum_max / bs + 2.0 : 1e9;
      denseBits = 6;
      while (denseBits < kMirrorBits && (double)(1 << denseBits) < 1.25 * reach) ++denseBits;
      if (const char* e = getenv("ITM_MIRROR_BITS")) { const int v = atoi(e); if (v >= 5 && v <= kMirrorBits) denseBits = v; }
    }
    const size_t pageBytes = (size_t)kPageBlocks * 512 * voxBytes, denseBytes = mirror_dense_cells(denseBits) * 512 * voxBytes;
    int pages = (int)(((size_t)768 << 20) / pageBytes);
    if (const char* e = getenv("ITM_MIRROR_PAGES")) { const int v = atoi(e); if (v > 0) pages = v; }
    size_t freeB = 0, totalB = 0;
    const bool haveInfo = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
    const bool off = mode && !strcmp(mode, "off");
    bool dense = !off && haveInfo && ((mode && !strcmp(mode, "dense")) ? freeB > denseBytes + ((size_t)1 << 30) : (!(mode && !strcmp(mode, "paged")) && freeB > 3 * denseBytes));
    const bool paged = !off && !dense && haveInfo && freeB > 3 * (size_t)pages * pageBytes;
    auto drop = [&]() {
      (void)hipGetLastError();
      (void)hipFree(s->sdfMirror); (void)hipFree(s->org.mTable); (void)hipFree(s->org.mPages);
      s->sdfMirror = nullptr; s->org.mTable = nullptr; s->org.mPages = nullptr; s->org.mMaxPages = 0; s->mirrorPages = 0;
    };
    if (dense) {
      s->org.mMaxPages = -denseBits;
      { const int half = 1 << (denseBits - 1); s->org.mx = -half; s->org.my = -half; s->org.mz = -half + half / 2; }      // (until the first view places it)
      if (hipMalloc(&s->sdfMirror, denseBytes) != hipSuccess || mirror_clear(s, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) drop();
    } else if (paged) {
      s->mirrorPages = pages;
      s->org.mMaxPages = pages;
      if (hipMalloc(&s->sdfMirror, (size_t)pages * pageBytes) != hipSuccess || hipMalloc((void**)&s->org.mTable, kMirrorTableCells * 4) != hipSuccess ||
          hipMalloc((void**)&s->org.mPages, 4) != hipSuccess || mirror_clear(s, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) drop();
    }
  }
  if (cfg.indexType == ITM_INDEX_HASH && s->device >= 0 && s->device < 64) { g_liveHashScenes[s->device].fetch_add(1); s->countedLive = true; }
  e = hipMemset(s->counters, 0, sizeof(SceneCounters));
  if (e == hipSuccess && s->allocKey) e = hipMemset(s->allocKey, 0, (size_t)s->noTotalEntries * 4);
  if (e == hipSuccess && s->headBits) e = hipMemset(s->headBits, 0, (size_t)(cfg.bucketNum + 31) / 32 * 4);
  if (e == hipSuccess && s->hash) e = hipMemset(s->hash, 0, (size_t)s->noTotalEntries * 16);
  if (e == hipSuccess && s->chunkReq) e = hipMemset(s->chunkReq, 0, (size_t)s->numChunks * 16);
  if (e == hipSuccess && s->chunkGran) e = hipMemset(s->chunkGran, 0, (size_t)s->numChunks * 8);
  if (e == hipSuccess && s->chunkSweepDone) e = hipMemset(s->chunkSweepDone, 0, (size_t)s->numChunks * 4);
  if (e == hipSuccess && s->chunkKeptGran) e = hipMemset(s->chunkKeptGran, 0, (size_t)s->numChunks * (kSweepChunk / 32) * 8);
  if (e == hipSuccess && s->dirPtr) e = hipMemset(s->dirPtr, 0xff, kDirCells * 4);
  if (e == hipSuccess && s->dirSlot) e = hipMemset(s->dirSlot, 0xff, kDirCells * 4);
  if (e != hipSuccess) { free_scene(s); return hip_fail(e, "hipMemset(scene)", __FILE__, __LINE__); }
  if (cfg.useSwapping && cfg.indexType == ITM_INDEX_HASH) {
    const int rc = create_swap_state(s);
    if (rc) { free_scene(s); return rc; }
  }
  // the host-visible status word (itm_internal.h); a device that cannot map host memory runs without it
  {
    void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      memset(h, 0, 64);
      s->fatalHost = (volatile int32_t*)h; s->fatalDev = (int32_t*)d;
    } else { if (h) (void)hipHostFree(h); (void)hipGetLastError(); }
  }
  s->deferredFusion = deferred_fusion_default();
  *out = s;
  return ITM_OK;
}

int itm_scene_destroy(itm_scene* s) { free_scene(s); return ITM_OK; }

int itm_stream_create(itm_stream* out) {
  if (!out) return set_error(ITM_ERR_INVALID, "null argument");
  hipStream_t st = nullptr;
  ITM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *out = (itm_stream)st;
  return ITM_OK;
}
int itm_stream_destroy(itm_stream stream) {
  if (!stream) return ITM_OK;
  ITM_HIP(hipStreamDestroy(as_stream(stream)));
  return ITM_OK;
}

int itm_scene_accel_info(const itm_scene* s, itm_accel_info* out) {
  if (!s || !out) return set_error(ITM_ERR_INVALID, "null argument");
  memset(out, 0, sizeof *out);
  out->directory_bytes = s->dirPtr ? (int64_t)(kDirCells * 4) : 0;
  out->slot_directory_bytes = s->dirSlot ? (int64_t)(kDirCells * 4) : 0;
  out->mirror_bytes = !s->sdfMirror ? 0 : s->org.mMaxPages < 0 ? (int64_t)(mirror_dense_cells(mirror_dense_bits(s->org)) * 512 * (mirror_is_float(s) ? 4 : 2))
                                                                  : (int64_t)((size_t)s->mirrorPages * kPageBlocks * 512 * (mirror_is_float(s) ? 4 : 2) + kMirrorTableCells * 4);
  out->mirror_pages = s->mirrorPages;          // 0 for the dense form
  out->mirror_pages_mapped = 0;
  if (s->sdfMirror && s->org.mPages) { int n = 0; if (hipMemcpy(&n, s->org.mPages, 4, hipMemcpyDeviceToHost) == hipSuccess) out->mirror_pages_mapped = n < s->mirrorPages ? n : s->mirrorPages; else (void)hipGetLastError(); }
  out->origin_directory[0] = s->org.dx; out->origin_directory[1] = s->org.dy; out->origin_directory[2] = s->org.dz;
  out->origin_mirror[0] = s->org.mx; out->origin_mirror[1] = s->org.my; out->origin_mirror[2] = s->org.mz;
  out->placed = s->orgPlaced ? 1 : 0;
  out->moves = s->accelMoves;
  return ITM_OK;
}

int itm_scene_get_config(const itm_scene* s, itm_scene_config* c, itm_scene_params* p) {
  if (!s) return set_error(ITM_ERR_INVALID, "null scene");
  if (c) *c = s->cfg;
  if (p) *p = s->prm;
  return ITM_OK;
}

int itm_reset_scene(itm_scene* s, itm_stream stream) {
  if (!s) return set_error(ITM_ERR_INVALID, "null scene");
  hipStream_t st = as_stream(stream);
  // (no fatal-status check: this is the call that clears it)
  if (s->deferredRs) { const int rc = flush_deferred(s->deferredRs); if (rc) return rc; }
  if (s->aheadRs) { const int rc = cancel_ahead(s, s->aheadRs, st); if (rc) return rc; }     // requests issued ahead die with the table
  if (s->fatalHost && *s->fatalHost) { ITM_HIP(hipStreamSynchronize(st)); *s->fatalHost = 0; }   // nothing still running may raise it again
  const int grid = 2048;
  if (s->cfg.voxelType == ITM_VOXEL_S && (s->numVoxels % 4) == 0) {
    reset_voxels_s_x4_kernel<<<grid, 256, 0, st>>>((uint4*)s->vba, s->numVoxels / 4);
  } else {
    int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
      using VX = decltype(vx);
      reset_voxels_kernel<VX><<<grid, 256, 0, st>>>(s->vba, s->numVoxels);
      return ITM_OK;
    });
    if (rc) return rc;
  }
  ITM_LAUNCH_CHECK();
  if (s->cfg.indexType == ITM_INDEX_HASH) {
    // the cubes are emptied through the table that filled them (a few MB of stores instead of an 18 GB memset), then the table is reset;
    // the next frame places them around its camera
    { int rc = accel_unfill(s, st); if (rc) return rc; }
    s->orgPlaced = false;
    ++s->tableEpoch;
    reset_hash_kernel<<<1024, 256, 0, st>>>(s->hash, s->noTotalEntries, s->excessList, s->cfg.excessNum, s->allocList,
                                            s->cfg.localBlockNum, s->allocKey, s->headBits, (s->cfg.bucketNum + 31) / 32, s->chunkReq, s->numChunks * 4, s->counters);
  } else {
    reset_dense_kernel<<<1, 1, 0, st>>>(s->allocList, s->counters);
  }
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_render_state_create(const itm_scene* s, int w, int h, itm_render_state** out) {
  if (!s || !out || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  if ((long long)w * h >= (1ll << 24)) return set_error(ITM_ERR_INVALID, "image too large");
  itm_render_state* r = new (std::nothrow) itm_render_state();
  if (!r) return set_error(ITM_ERR_DEVICE, "out of host memory");
  r->scene = s; r->w = w; r->h = h;
  r->hash = s->cfg.indexType == ITM_INDEX_HASH;
  r->capIds = r->hash ? s->cfg.localBlockNum : 0;
  size_t P = (size_t)w * h;
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
  alloc((void**)&r->range, P * 8);
  alloc((void**)&r->raycast, P * 16);
  alloc((void**)&r->fwdProj, P * 16);
  alloc((void**)&r->missing, P * 4);
  alloc((void**)&r->image, P * 4);
  alloc((void**)&r->counters, sizeof(RenderCounters));
  alloc((void**)&r->pixScratch, P * 4);
  alloc((void**)&r->pixChunk, ((P + kSweepChunk - 1) / kSweepChunk) * 4 + 16);
  if (r->hash) {
    alloc((void**)&r->visibleIds, (size_t)r->capIds * 4);
    alloc((void**)&r->visibleType, (size_t)s->noTotalEntries);
    alloc((void**)&r->projBuf, (size_t)r->capIds * 32);
    if ((size_t)((w + 7) / 8) * ((h + 7) / 8) * 8 <= 150 * 1024) alloc((void**)&r->rangePartials, (size_t)64 * ((w + 7) / 8) * ((h + 7) / 8) * 8);   // room for up to 64 partial images
  }
  if (e != hipSuccess) { free_rs(r); return hip_fail(e, "hipMalloc(render state)", __FILE__, __LINE__); }
  // MemoryBlock<T> storage is zero-initialised in the reference (ORUtils/MemoryBlock.h Clear on allocate)
  e = hipMemset(r->counters, 0, sizeof(RenderCounters));
  if (e == hipSuccess) e = hipMemset(r->raycast, 0, P * 16);
  if (e == hipSuccess) e = hipMemset(r->fwdProj, 0, P * 16);
  if (e == hipSuccess) e = hipMemset(r->missing, 0, P * 4);
  if (e == hipSuccess) e = hipMemset(r->image, 0, P * 4);
  if (e == hipSuccess && r->hash) e = hipMemset(r->visibleIds, 0, (size_t)r->capIds * 4);
  if (e == hipSuccess && r->hash) e = hipMemset(r->visibleType, 0, (size_t)s->noTotalEntries);
  if (e != hipSuccess) { free_rs(r); return hip_fail(e, "hipMemset(render state)", __FILE__, __LINE__); }
  fill_range_kernel<<<256, 256, 0, 0>>>(r->range, (int)P, s->prm.viewFrustum_min, s->prm.viewFrustum_max);
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { free_rs(r); return hip_fail(e, "init range image", __FILE__, __LINE__); }
  { std::lock_guard<std::mutex> lock(g_rsRegistryMutex); s->renderStates.push_back(r); }
  *out = r;
  return ITM_OK;
}
int itm_render_state_destroy(itm_render_state* r) { free_rs(r); return ITM_OK; }

int itm_debug_div32767(const float* in, float* out, int n, itm_stream st) {
  if (!in || !out || n < 0) return set_error(ITM_ERR_INVALID, "bad argument");
  div32767_kernel<<<(n + 255) / 256, 256, 0, as_stream(st)>>>(in, out, n);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_debug_divide(int mode, const float* a, const float* b, const float* r, float* out, int n, itm_stream st) {
  if (!a || !b || !out || n < 0 || mode < 1 || mode > 4 || (mode == 4 && !r)) return set_error(ITM_ERR_INVALID, "bad argument");
  divide_kernel<<<(n + 255) / 256, 256, 0, as_stream(st)>>>(mode, a, b, r, out, n);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_convert_depth_affine(const int16_t* raw, float* out, int w, int h, float a, float b, itm_stream st) {
  if (!raw || !out || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  int n = w * h;
  { const int rc = flush_overlapping(out, (size_t)n * 4, as_stream(st)); if (rc) return rc; }
  depth_affine_kernel<<<(n + 255) / 256, 256, 0, as_stream(st)>>>(raw, out, n, a, b);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}
int itm_convert_disparity(const int16_t* raw, float* out, int w, int h, float c0, float c1, float fx, itm_stream st) {
  if (!raw || !out || w <= 0 || h <= 0) return set_error(ITM_ERR_INVALID, "bad argument");
  int n = w * h;
  { const int rc = flush_overlapping(out, (size_t)n * 4, as_stream(st)); if (rc) return rc; }
  depth_disparity_kernel<<<(n + 255) / 256, 256, 0, as_stream(st)>>>(raw, out, n, c0, c1, fx);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_profile_enable(itm_scene* s, uint32_t mask) {
  if (!s) return set_error(ITM_ERR_INVALID, "null scene");
  if (!s->prof) s->prof = new Profiler();
  s->prof->mask = mask;
  return ITM_OK;
}
int itm_profile_sample(itm_scene* s, int every) {
  if (!s || every < 1) return set_error(ITM_ERR_INVALID, "profile_sample: null scene or every < 1");
  if (!s->prof) s->prof = new Profiler();
  s->prof->every = every;
  for (int i = 0; i < ITM_TK_COUNT; ++i) s->prof->tick[i] = 0;
  return ITM_OK;
}
// n empty brackets (an event pair with nothing between) on `stream`, accumulated in slot ITM_TK_EMPTY: what an event pair adds to
// the interval it brackets on this machine and queue -- the part of a timed kernel's figure that is not the kernel
int itm_profile_calibrate(itm_scene* s, int n, itm_stream stream) {
  if (!s || n < 1) return set_error(ITM_ERR_INVALID, "profile_calibrate: null scene or n < 1");
  if (!s->prof) s->prof = new Profiler();
  Profiler* p = s->prof;
  const uint32_t mask = p->mask; const int every = p->every;
  p->mask |= 1u << ITM_TK_EMPTY; p->every = 1;
  for (int i = 0; i < n; ++i) { KernelTimer tk(s, ITM_TK_EMPTY, as_stream(stream)); }
  p->mask = mask; p->every = every;
  return ITM_OK;
}
int itm_profile_read(itm_scene* s, itm_profile* out, int reset) {
  if (!s || !out) return set_error(ITM_ERR_INVALID, "null argument");
  memset(out, 0, sizeof *out);
  if (!s->prof) return ITM_OK;
  s->prof->flush();
  for (int i = 0; i < ITM_TK_COUNT; ++i) { out->calls[i] = s->prof->calls[i]; out->total_ms[i] = s->prof->total_ms[i]; }
  if (reset) for (int i = 0; i < ITM_TK_COUNT; ++i) { s->prof->calls[i] = 0; s->prof->total_ms[i] = 0; }
  return ITM_OK;
}

int itm_get_counters(const itm_scene* s, const itm_render_state* rs, itm_counters* out, itm_stream stream) {
  if (!out) return set_error(ITM_ERR_INVALID, "null argument");
  memset(out, 0, sizeof *out);
  hipStream_t st = as_stream(stream);
  // recorded calls first; a fatal status is reported AFTER the counters have been read (they say what happened)
  if (s && s->deferredRs) { const int frc = flush_deferred(s->deferredRs); if (frc) return frc; }
  if (rs && rs->deferred.stage) { const int frc = flush_deferred(const_cast<itm_render_state*>(rs)); if (frc) return frc; }
  SceneCounters sc{}; RenderCounters rc{};
  if (s) ITM_HIP(hipMemcpyAsync(&sc, s->counters, sizeof sc, hipMemcpyDeviceToHost, st));
  if (rs) ITM_HIP(hipMemcpyAsync(&rc, rs->counters, sizeof rc, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  out->lastFreeBlockId = sc.lastFreeBlockId;
  out->lastFreeExcessListId = sc.lastFreeExcessListId;
  out->noAllocRequests = sc.noAllocRequests;
  out->statusFlags = sc.statusFlags;
  out->noVisibleEntries = rc.noVisibleEntries;
  out->noFwdProjMissingPoints = rc.noFwdProjMissingPoints;
  out->noTotalPoints = rc.noTotalPoints;
  out->noRenderingBlocks = (rc.renderingBlocksAccepted >= 0) ? rc.renderingBlocksAccepted : rc.noRenderingBlocks;
  return enter_scene(s ? s : (rs ? rs->scene : nullptr), nullptr);          // ITM_ERR_DEVICE once the scene has raised a fatal status
}

int itm_set_counters(itm_scene* s, itm_render_state* rs, const itm_counters* in, itm_stream stream) {
  if (!in) return set_error(ITM_ERR_INVALID, "null argument");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  if (rs) rs->denseRangeReady = false;
  hipStream_t st = as_stream(stream);
  if (s) {
    SceneCounters sc{};
    ITM_HIP(hipMemcpyAsync(&sc, s->counters, sizeof sc, hipMemcpyDeviceToHost, st));
    ITM_HIP(hipStreamSynchronize(st));
    sc.lastFreeBlockId = in->lastFreeBlockId;
    sc.lastFreeExcessListId = in->lastFreeExcessListId;
    ITM_HIP(hipMemcpyAsync(s->counters, &sc, sizeof sc, hipMemcpyHostToDevice, st));
    ITM_HIP(hipStreamSynchronize(st));
  }
  if (rs) {
    RenderCounters rc{};
    ITM_HIP(hipMemcpyAsync(&rc, rs->counters, sizeof rc, hipMemcpyDeviceToHost, st));
    ITM_HIP(hipStreamSynchronize(st));
    rc.noVisibleEntries = in->noVisibleEntries;
    rc.rawVisibleCount = in->noVisibleEntries;       // (a restored list is taken as complete)
    rs->listCoherent = false;
    ITM_HIP(hipMemcpyAsync(rs->counters, &rc, sizeof rc, hipMemcpyHostToDevice, st));
    ITM_HIP(hipStreamSynchronize(st));
  }
  return ITM_OK;
}

size_t itm_buffer_bytes(const itm_scene* s, const itm_render_state* rs, int which) {
  size_t b; buffer_of(s, rs, which, &b); return b;
}
void* itm_buffer_ptr(const itm_scene* s, const itm_render_state* rs, int which) {
  if (enter_scene(s, rs)) return nullptr;       // (what is recorded now is launched; calls recorded LATER are the holder's to itm_flush)
  size_t b; return buffer_of(s, rs, which, &b);
}
int itm_download(const itm_scene* s, const itm_render_state* rs, int which, void* dst, size_t bytes, itm_stream stream) {
  size_t b; void* p = buffer_of(s, rs, which, &b);
  if (!p && bytes == 0) return ITM_OK;
  if (!p || !dst || bytes > b) return set_error(ITM_ERR_INVALID, "bad buffer / size");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  if (which == ITM_BUF_VISIBLE_TYPE && refuse_while_ahead(s, rs, "download of the visible types")) return ITM_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  ITM_HIP(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  return enter_scene(s, rs);           // a fatal status raised by the work that has just drained
}
int itm_upload(itm_scene* s, itm_render_state* rs, int which, const void* src, size_t bytes, itm_stream stream) {
  size_t b; void* p = buffer_of(s, rs, which, &b);
  if (!p || !src || bytes > b) return set_error(ITM_ERR_INVALID, "bad buffer / size");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  if ((which == ITM_BUF_VISIBLE_IDS || which == ITM_BUF_VISIBLE_TYPE) && refuse_while_ahead(s, rs, "upload of the visible list")) return ITM_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  if (which == ITM_BUF_HASH_ENTRIES) { int rc = accel_unfill(s, st); if (rc) return rc; ++s->tableEpoch; }      // while the table still holds what filled the cubes
  ITM_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, st));
  if (which == ITM_BUF_HASH_ENTRIES) {
    if (bytes < b) {          // a partial upload: the rest of the table stays; place by the whole table as it now is
      std::vector<HashEntry> all(b / sizeof(HashEntry));
      ITM_HIP(hipMemcpyAsync(all.data(), p, b, hipMemcpyDeviceToHost, st));
      ITM_HIP(hipStreamSynchronize(st));
      accel_place_for_table(s, all.data(), all.size());
    } else accel_place_for_table(s, (const HashEntry*)src, bytes / sizeof(HashEntry));
    int rc = rebuild_head_bits(s, st); if (rc) return rc;
  }
  if (which == ITM_BUF_VOXEL_BLOCKS) { int rc = rebuild_sdf_mirror(s, st); if (rc) return rc; }   // the table is the same, the values are new
  if (rs && (which == ITM_BUF_VISIBLE_IDS || which == ITM_BUF_VISIBLE_TYPE)) rs->listCoherent = false;
  if (rs) rs->denseRangeReady = false;
  ITM_HIP(hipStreamSynchronize(st));
  return ITM_OK;
}

int itm_export_visible_record(const itm_render_state* rs, const float M_d[16], int max_ids, void* dst, itm_stream stream) {
  if (!rs || !rs->hash || !dst || !M_d || max_ids < 0) return set_error(ITM_ERR_INVALID, "bad argument");
  if (!rs->scene) return set_error(ITM_ERR_INVALID, "the render state's scene has been destroyed");
  { const int rc = enter_scene(rs->scene, rs); if (rc) return rc; }
  Mat4 M; memcpy(M.m, M_d, 64);
  int n = (max_ids > 17 ? max_ids : 17);
  export_record_kernel<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(rs->visibleIds, rs->counters, M, max_ids, (int32_t*)dst);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

}  // extern "C"
