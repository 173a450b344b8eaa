// itm_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Single-threaded CPU restatement of the InfiniTAM allocate / integrate / raycast path, written
// from the reference's behaviour (not its text) so that the HIP kernels can be checked against
// something that runs everywhere (the reference itself cannot travel to the GPU box).
// It implements the same C-ABI as include/itm_hip.h with the prefix `itmo_` and HOST pointers.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (infinitam_amd/, libitmhip.so) never includes, links or calls anything in oracle/.
//
// Parity status: PINNED.  The reference ships no tests or golden vectors, so
// this restatement is pinned against the reference's own CPU engines compiled from
// /root/reference by oracle/Makefile (target `ref`, output oracle/_ref/libitm_ref.so) --
// tests/test_oracle_vs_reference.py compares them bit-for-bit where /root/reference exists --
// and against fixtures generated from that build (tests/golden/, tests/golden/make_golden.py).
//
// Build: g++ -O2 -ffp-contract=off (no -march=native, no fast-math, no OpenMP): float results
// must not depend on contraction.  All arithmetic is fp32 in the operation order documented in
// SURVEY.md Appendix A; each function cites the reference lines it follows
// (paths relative to /root/reference/InfiniTAM/ITMLib).

#define ITM_FN(name) itmo_##name
#include "../include/itm_hip.h"
#include "../include/itm_debug.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }

// ------------------------------------------------------------------------------------------
// POD layouts (Utils/ITMLibDefines.h:71-82, :100-199).  Padding bytes are zeroed here; the
// reference leaves them uninitialised, so comparisons are field-wise.
// ------------------------------------------------------------------------------------------
struct HashEntry { int16_t px, py, pz, pad; int32_t offset; int32_t ptr; };
static_assert(sizeof(HashEntry) == 16, "ITMHashEntry is 16 bytes");

struct VoxS   { int16_t sdf; uint8_t w; uint8_t pad; };
struct VoxF   { float sdf; uint8_t w; uint8_t pad[3]; };
struct VoxSC  { int16_t sdf; uint8_t w; uint8_t clr[3]; uint8_t wc; uint8_t pad; };
struct VoxFC  { float sdf; uint8_t w; uint8_t clr[3]; uint8_t wc; uint8_t pad[3]; };
static_assert(sizeof(VoxS) == 4 && sizeof(VoxF) == 8 && sizeof(VoxSC) == 8 && sizeof(VoxFC) == 12, "voxel sizes");

// voxel codec traits: SDF_valueToFloat / SDF_floatToValue / SDF_initialValue / hasColorInformation
template <class V> struct Codec;
template <> struct Codec<VoxS> {
  static constexpr bool color = false;
  static float toF(float raw) { return raw / 32767.0f; }
  static int16_t toV(float f) { return (int16_t)(f * 32767.0f); }
  static VoxS init() { VoxS v; std::memset(&v, 0, sizeof v); v.sdf = 32767; return v; }
};
template <> struct Codec<VoxSC> {
  static constexpr bool color = true;
  static float toF(float raw) { return raw / 32767.0f; }
  static int16_t toV(float f) { return (int16_t)(f * 32767.0f); }
  static VoxSC init() { VoxSC v; std::memset(&v, 0, sizeof v); v.sdf = 32767; return v; }
};
template <> struct Codec<VoxF> {
  static constexpr bool color = false;
  static float toF(float raw) { return raw; }
  static float toV(float f) { return f; }
  static VoxF init() { VoxF v; std::memset(&v, 0, sizeof v); v.sdf = 1.0f; return v; }
};
template <> struct Codec<VoxFC> {
  static constexpr bool color = true;
  static float toF(float raw) { return raw; }
  static float toV(float f) { return f; }
  static VoxFC init() { VoxFC v; std::memset(&v, 0, sizeof v); v.sdf = 1.0f; return v; }
};

struct V2f { float x, y; };
struct V4f { float x, y, z, w; };
struct V3f { float x, y, z; };
struct V3i { int x, y, z; };

inline float fmin_ref(float a, float b) { return (a < b) ? a : b; }   // MIN  ORUtils/MathUtils.h:5-7
inline float fmax_ref(float a, float b) { return (a < b) ? b : a; }   // MAX  :9-11
inline float round_ref(float x) { return (x < 0) ? (x - 0.5f) : (x + 0.5f); }  // ROUND :21-23

// Matrix4 * Vector4, ORUtils/Matrix.h:115-122: left-to-right sum of four products per row.
inline V4f mul(const float* m, const V4f& v) {
  V4f r;
  r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
  r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
  r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
  r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
  return r;
}

// Matrix4 * Matrix4, ORUtils/Matrix.h:102-108: r(col,row) accumulated from zero over k.
void matmul4(const float* lhs, const float* rhs, float* out) {
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += lhs[k * 4 + row] * rhs[col * 4 + k];
      out[col * 4 + row] = acc;
    }
}

// Matrix4::inv, ORUtils/Matrix.h:162-223: cofactor expansion on the transposed matrix, every
// element finally multiplied by 1/det.  `p3` is the left-to-right sum of three products.
inline float p3(float a, float b, float c, float d, float e, float f) { return a * b + c * d + e * f; }
bool invert4(const float* m, float* dst) {
  float s[16], t[12];
  for (int i = 0; i < 4; ++i) { s[i] = m[i * 4]; s[i + 4] = m[i * 4 + 1]; s[i + 8] = m[i * 4 + 2]; s[i + 12] = m[i * 4 + 3]; }
  t[0] = s[10] * s[15]; t[1] = s[11] * s[14]; t[2] = s[9] * s[15];  t[3] = s[11] * s[13];
  t[4] = s[9] * s[14];  t[5] = s[10] * s[13]; t[6] = s[8] * s[15];  t[7] = s[11] * s[12];
  t[8] = s[8] * s[14];  t[9] = s[10] * s[12]; t[10] = s[8] * s[13]; t[11] = s[9] * s[12];
  dst[0] = p3(t[0], s[5], t[3], s[6], t[4], s[7]) - p3(t[1], s[5], t[2], s[6], t[5], s[7]);
  dst[1] = p3(t[1], s[4], t[6], s[6], t[9], s[7]) - p3(t[0], s[4], t[7], s[6], t[8], s[7]);
  dst[2] = p3(t[2], s[4], t[7], s[5], t[10], s[7]) - p3(t[3], s[4], t[6], s[5], t[11], s[7]);
  dst[3] = p3(t[5], s[4], t[8], s[5], t[11], s[6]) - p3(t[4], s[4], t[9], s[5], t[10], s[6]);
  float det = s[0] * dst[0] + s[1] * dst[1] + s[2] * dst[2] + s[3] * dst[3];
  if (det == 0.0f) return false;
  dst[4] = p3(t[1], s[1], t[2], s[2], t[5], s[3]) - p3(t[0], s[1], t[3], s[2], t[4], s[3]);
  dst[5] = p3(t[0], s[0], t[7], s[2], t[8], s[3]) - p3(t[1], s[0], t[6], s[2], t[9], s[3]);
  dst[6] = p3(t[3], s[0], t[6], s[1], t[11], s[3]) - p3(t[2], s[0], t[7], s[1], t[10], s[3]);
  dst[7] = p3(t[4], s[0], t[9], s[1], t[10], s[2]) - p3(t[5], s[0], t[8], s[1], t[11], s[2]);
  t[0] = s[2] * s[7]; t[1] = s[3] * s[6]; t[2] = s[1] * s[7];  t[3] = s[3] * s[5];
  t[4] = s[1] * s[6]; t[5] = s[2] * s[5]; t[6] = s[0] * s[7];  t[7] = s[3] * s[4];
  t[8] = s[0] * s[6]; t[9] = s[2] * s[4]; t[10] = s[0] * s[5]; t[11] = s[1] * s[4];
  dst[8] = p3(t[0], s[13], t[3], s[14], t[4], s[15]) - p3(t[1], s[13], t[2], s[14], t[5], s[15]);
  dst[9] = p3(t[1], s[12], t[6], s[14], t[9], s[15]) - p3(t[0], s[12], t[7], s[14], t[8], s[15]);
  dst[10] = p3(t[2], s[12], t[7], s[13], t[10], s[15]) - p3(t[3], s[12], t[6], s[13], t[11], s[15]);
  dst[11] = p3(t[5], s[12], t[8], s[13], t[11], s[14]) - p3(t[4], s[12], t[9], s[13], t[10], s[14]);
  dst[12] = p3(t[2], s[10], t[5], s[11], t[1], s[9]) - p3(t[4], s[11], t[0], s[9], t[3], s[10]);
  dst[13] = p3(t[8], s[11], t[0], s[8], t[7], s[10]) - p3(t[6], s[10], t[9], s[11], t[1], s[8]);
  dst[14] = p3(t[6], s[9], t[11], s[11], t[3], s[8]) - p3(t[10], s[11], t[2], s[8], t[7], s[9]);
  dst[15] = p3(t[10], s[10], t[4], s[8], t[9], s[9]) - p3(t[8], s[9], t[11], s[10], t[5], s[8]);
  float rdet = 1 / det;
  for (int i = 0; i < 16; ++i) dst[i] *= rdet;
  return true;
}

// hashIndex, DeviceAgnostic/ITMRepresentationAccess.h:8-10 (coordinates sign-extend to uint32)
inline int hash_index(int bx, int by, int bz, uint32_t mask) {
  return (int)((((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) & mask);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// opaque objects
// ------------------------------------------------------------------------------------------
struct itm_scene {
  itm_scene_config cfg;
  itm_scene_params prm;
  size_t voxBytes;
  int noTotalEntries;
  std::vector<HashEntry> hash;       // scene->index.GetEntries()
  std::vector<int32_t> excessList;   // GetExcessAllocationList()
  std::vector<uint8_t> vba;          // localVBA.GetVoxelBlocks()
  std::vector<int32_t> allocList;    // localVBA.GetAllocationList()
  int lastFreeBlockId = 0, lastFreeExcessListId = 0;
  // engine scratch (ITMSceneReconstructionEngine_CPU members)
  std::vector<uint8_t> allocType;
  std::vector<int16_t> blockCoords;  // 4 shorts per slot
  int noAllocRequests = 0;
  // ITMGlobalCache (Objects/ITMGlobalCache.h), scenes with useSwapping
  std::vector<uint8_t> swapStates;     // ITMHashSwapState::state per entry
  std::vector<uint8_t> hasStoredData;
  uint8_t* storedBlocks = nullptr;     // noTotalEntries x 512 voxels, calloc: untouched pages cost nothing
  ~itm_scene() { std::free(storedBlocks); }
  int transferBlockNum() const { return cfg.transferBlockNum > 0 ? cfg.transferBlockNum : 0x1000; }
  size_t numVoxels() const {
    return cfg.indexType == ITM_INDEX_HASH ? (size_t)cfg.localBlockNum * 512
                                           : (size_t)cfg.denseSize[0] * cfg.denseSize[1] * cfg.denseSize[2];
  }
};

struct itm_render_state {
  int w, h;
  bool hash;
  int capIds;
  int rawVisible = 0;      // visible slots found by the last list build, before the clamp to capIds
  std::vector<V2f> range;        // renderingRangeImage
  std::vector<V4f> raycast;      // raycastResult
  std::vector<V4f> fwdProj;      // forwardProjection
  std::vector<int32_t> missing;  // fwdProjMissingPoints
  std::vector<uint32_t> image;   // raycastImage (uchar4)
  std::vector<int32_t> visibleIds;
  std::vector<uint8_t> visibleType;
  int noVisibleEntries = 0, noFwdProjMissingPoints = 0, noTotalPoints = 0, noRenderingBlocks = 0;
};

#include "../infinitam_amd/csrc/mc_tables.h"   // marching-cubes case table: DATA generated from the reference
struct itm_mesh {
  const itm_scene* scene;
  uint32_t maxTriangles, noTotalTriangles;
  std::vector<float> tri;   // ITMMesh::Triangle: 9 floats each
};

namespace {

// ------------------------------------------------------------------------------------------
// voxel access: readVoxel for the hash (DeviceAgnostic/ITMRepresentationAccess.h:85-119, block
// decomposition :12-20, IndexCache Objects/ITMVoxelBlockHash.h:27-33) and for the dense array
// (:61-77, :129-135).
// ------------------------------------------------------------------------------------------
struct BlockCache { int bx = 0x7fffffff, by = 0x7fffffff, bz = 0x7fffffff; int base = -1; };

// Work counters (the "algorithmic bytes" of DESIGN.md are priced from these); read through
// itmo_debug_stats, which is not part of the shared ABI.
struct Stats {
  long long rays, ray_hits, ray_steps, max_ray_steps;
  long long nearest_reads, nearest_misses, trilinear_reads, voxel_reads, hash_probes;
  long long alloc_pixels, alloc_steps, alloc_probes;
  long long fuse_blocks, fuse_voxels_visited, fuse_voxels_updated;
};
Stats g_stats;
// The OpenMP build (libitm_oracle_omp.so, TIMING ONLY: bench.py's all-cores CPU baseline) parallelises the loops the
// reference parallelises under WITH_OPENMP (ITMSceneReconstructionEngine_CPU.cpp:80,164,348; ITMVisualisationEngine_CPU.cpp:
// 168,211,283); its work counters are compiled out (shared counters would serialise the threads) and, like the
// reference's OpenMP allocation loop, its request pass is racy -- parity tests always use the single-thread build.
#ifdef _OPENMP
#define ITMO_STAT(x) ((void)0)
#define ITMO_PARALLEL_FOR _Pragma("omp parallel for schedule(dynamic, 64)")
#define ITMO_PARALLEL_FOR_STATIC _Pragma("omp parallel for schedule(static)")
#else
#define ITMO_STAT(x) (x)
#define ITMO_PARALLEL_FOR
#define ITMO_PARALLEL_FOR_STATIC
#endif
int* g_rayTrace = nullptr;  // optional per-ray census, 8 ints: steps, band (trilinear) steps, not-found steps, unit steps on a single-voxel value,
                            // unit steps on a trilinear value, steps on a value of exactly 1, longest not-found run, steps after that run; width in g_rayTraceW
int g_rayTraceW = 0;

inline int floor_div8(int p) { return ((p < 0) ? p - 7 : p) / 8; }

template <class V>
struct Reader {
  const itm_scene* sc;
  const V* vox;
  bool dense;
  uint32_t mask;
  explicit Reader(const itm_scene* s) : sc(s), vox((const V*)s->vba.data()), dense(s->cfg.indexType == ITM_INDEX_DENSE), mask((uint32_t)s->cfg.bucketNum - 1u) {}

  V read(int px, int py, int pz, bool& found, BlockCache& cache) const {
    if (dense) {
      int qx = px - sc->cfg.denseOffset[0], qy = py - sc->cfg.denseOffset[1], qz = pz - sc->cfg.denseOffset[2];
      const int* sz = sc->cfg.denseSize;
      if (qx < 0 || qx >= sz[0] || qy < 0 || qy >= sz[1] || qz < 0 || qz >= sz[2]) { found = false; return Codec<V>::init(); }
      found = true;
      return vox[qx + qy * sz[0] + qz * sz[0] * sz[1]];
    }
    int bx = floor_div8(px), by = floor_div8(py), bz = floor_div8(pz);
    // the reference's expression, kept in int arithmetic (equals the in-block linear index)
    int lin = px + (py - bx) * 8 + (pz - by) * 64 - bz * 512;
    if (bx == cache.bx && by == cache.by && bz == cache.bz) { found = true; return vox[cache.base + lin]; }
    int idx = hash_index(bx, by, bz, mask);
    for (;;) {
      const HashEntry& e = sc->hash[idx];
      ITMO_STAT(++g_stats.hash_probes);
      if (e.px == bx && e.py == by && e.pz == bz && e.ptr >= 0) {
        found = true;
        cache.bx = bx; cache.by = by; cache.bz = bz; cache.base = e.ptr * 512;
        return vox[cache.base + lin];
      }
      if (e.offset < 1) break;
      idx = sc->cfg.bucketNum + e.offset - 1;
    }
    found = false;
    return Codec<V>::init();
  }
  V read(int px, int py, int pz, bool& found) const { BlockCache c; return read(px, py, pz, found, c); }

  // readFromSDF_float_uninterpolated :153-159
  float nearest(const V3f& p, bool& found, BlockCache& cache) const {
    V v = read((int)round_ref(p.x), (int)round_ref(p.y), (int)round_ref(p.z), found, cache);
    ITMO_STAT(++g_stats.nearest_reads); if (!found) ITMO_STAT(++g_stats.nearest_misses); else ITMO_STAT(++g_stats.voxel_reads);
    return Codec<V>::toF((float)v.sdf);
  }
  // readFromSDF_float_interpolated :161-185 (raw values blended, then converted; found := true)
  float trilinear(const V3f& p, bool& found, BlockCache& cache) const {
    float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    float cx = p.x - fx, cy = p.y - fy, cz = p.z - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float v1, v2, r1, r2;
    v1 = (float)read(ix, iy, iz, found, cache).sdf;
    v2 = (float)read(ix + 1, iy, iz, found, cache).sdf;
    r1 = (1.0f - cx) * v1 + cx * v2;
    v1 = (float)read(ix, iy + 1, iz, found, cache).sdf;
    v2 = (float)read(ix + 1, iy + 1, iz, found, cache).sdf;
    r1 = (1.0f - cy) * r1 + cy * ((1.0f - cx) * v1 + cx * v2);
    v1 = (float)read(ix, iy, iz + 1, found, cache).sdf;
    v2 = (float)read(ix + 1, iy, iz + 1, found, cache).sdf;
    r2 = (1.0f - cx) * v1 + cx * v2;
    v1 = (float)read(ix, iy + 1, iz + 1, found, cache).sdf;
    v2 = (float)read(ix + 1, iy + 1, iz + 1, found, cache).sdf;
    r2 = (1.0f - cy) * r2 + cy * ((1.0f - cx) * v1 + cx * v2);
    found = true;
    ITMO_STAT(++g_stats.trilinear_reads); ITMO_STAT(g_stats.voxel_reads += 8);
    return Codec<V>::toF((1.0f - cz) * r1 + cz * r2);
  }
};

// ------------------------------------------------------------------------------------------
// ResetScene   DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:24-45 / :301-312
// ------------------------------------------------------------------------------------------
template <class V>
void reset_scene_t(itm_scene* s) {
  V* vox = (V*)s->vba.data();
  V init = Codec<V>::init();
  size_t n = s->numVoxels();
  for (size_t i = 0; i < n; ++i) vox[i] = init;
  int nb = (s->cfg.indexType == ITM_INDEX_HASH) ? s->cfg.localBlockNum : 1;
  for (int i = 0; i < nb; ++i) s->allocList[i] = i;
  s->lastFreeBlockId = nb - 1;
  if (s->cfg.indexType == ITM_INDEX_HASH) {
    HashEntry e; std::memset(&e, 0, sizeof e); e.ptr = -2;
    for (int i = 0; i < s->noTotalEntries; ++i) s->hash[i] = e;
    for (int i = 0; i < s->cfg.excessNum; ++i) s->excessList[i] = i;
    s->lastFreeExcessListId = s->cfg.excessNum - 1;
  }
}

// ------------------------------------------------------------------------------------------
// per-voxel fusion
//   computeUpdatedVoxelDepthInfo  DeviceAgnostic/ITMSceneReconstructionEngine.h:9-56
//   computeUpdatedVoxelColorInfo  :58-100 ; gate :121-139 ; interpolateBilinear
//   DeviceAgnostic/ITMPixelUtils.h:11-39
// ------------------------------------------------------------------------------------------
struct FuseCtx {
  float M_d[16], M_rgb[16];
  float fx, fy, cx, cy;          // depth intrinsics
  float fxc, fyc, cxc, cyc;      // rgb intrinsics
  float mu; int maxW;
  const float* depth; const uint8_t* rgb;
  int w, h, wc, hc;
  bool stopAtMax;
};

template <class V>
inline float fuse_depth(V& vox, const V4f& pm, const FuseCtx& c) {
  V4f pc = mul(c.M_d, pm);
  if (pc.z <= 0) return -1;
  float u = c.fx * pc.x / pc.z + c.cx;
  float v = c.fy * pc.y / pc.z + c.cy;
  if ((u < 1) || (u > c.w - 2) || (v < 1) || (v > c.h - 2)) return -1;
  float dm = c.depth[(int)(u + 0.5f) + (int)(v + 0.5f) * c.w];
  if (dm <= 0.0) return -1;
  float eta = dm - pc.z;
  if (eta < -c.mu) return eta;
  float oldF = Codec<V>::toF((float)vox.sdf);
  int oldW = vox.w;
  float newF = fmin_ref(1.0f, eta / c.mu);
  int newW = 1;
  newF = oldW * oldF + newW * newF;
  newW = oldW + newW;
  newF /= newW;
  newW = (newW < c.maxW) ? newW : c.maxW;
  vox.sdf = Codec<V>::toV(newF);
  vox.w = (uint8_t)newW;
  ITMO_STAT(++g_stats.fuse_voxels_updated);
  return eta;
}

template <class V>
inline void fuse_colour(V& vox, const V4f& pm, const FuseCtx& c) {
  float oldW = (float)vox.wc;
  float oc[3] = {(float)vox.clr[0] / 255.0f, (float)vox.clr[1] / 255.0f, (float)vox.clr[2] / 255.0f};
  V4f pc = mul(c.M_rgb, pm);
  float u = c.fxc * pc.x / pc.z + c.cxc;
  float v = c.fyc * pc.y / pc.z + c.cyc;
  if ((u < 1) || (u > c.wc - 2) || (v < 1) || (v > c.hc - 2)) return;
  // bilinear tap pattern: b/c/d are only read when their weight can be non-zero
  int px = (int)std::floor(u), py = (int)std::floor(v);
  float dx = u - (float)px, dy = v - (float)py;
  const uint8_t* A = c.rgb + 4 * ((size_t)px + (size_t)py * c.wc);
  uint8_t zero[4] = {0, 0, 0, 0};
  const uint8_t* B = zero; const uint8_t* C = zero; const uint8_t* D = zero;
  if (dx != 0) B = c.rgb + 4 * ((size_t)(px + 1) + (size_t)py * c.wc);
  if (dy != 0) C = c.rgb + 4 * ((size_t)px + (size_t)(py + 1) * c.wc);
  if (dx != 0 && dy != 0) D = c.rgb + 4 * ((size_t)(px + 1) + (size_t)(py + 1) * c.wc);
  float meas[3];
  for (int k = 0; k < 3; ++k) {
    float r = ((float)A[k] * (1.0f - dx) * (1.0f - dy) + (float)B[k] * dx * (1.0f - dy) +
               (float)C[k] * (1.0f - dx) * dy + (float)D[k] * dx * dy);
    meas[k] = r / 255.0f;
  }
  float newW = 1;
  float nc[3];
  for (int k = 0; k < 3; ++k) nc[k] = oc[k] * oldW + meas[k] * newW;
  newW = oldW + newW;
  for (int k = 0; k < 3; ++k) nc[k] /= newW;
  uint8_t maxWu = (uint8_t)c.maxW;
  newW = (newW < maxWu) ? newW : (float)maxWu;
  for (int k = 0; k < 3; ++k) {
    int vi = (int)round_ref(nc[k] * 255.0f);
    int lo = (vi < 255) ? vi : 255;           // MIN(255, vi)
    vox.clr[k] = (uint8_t)((0 < lo) ? lo : 0);  // MAX(0, .)
  }
  vox.wc = (uint8_t)newW;
}

template <class V, bool C = Codec<V>::color> struct Fuse;
template <class V> struct Fuse<V, false> {
  static void run(V& vox, const V4f& pm, const FuseCtx& c) { fuse_depth(vox, pm, c); }
};
template <class V> struct Fuse<V, true> {
  static void run(V& vox, const V4f& pm, const FuseCtx& c) {
    float eta = fuse_depth(vox, pm, c);
    if ((eta > c.mu) || (std::fabs(eta / c.mu) > 0.25f)) return;
    fuse_colour(vox, pm, c);
  }
};

void make_fuse_ctx(const itm_scene* s, const itm_view* v, FuseCtx& c) {
  std::memcpy(c.M_d, v->M_d, sizeof c.M_d);
  matmul4(v->rgb_to_depth_inv, v->M_d, c.M_rgb);   // calib_inv * M_d  (_CPU.cpp:61)
  c.fx = v->intr_d[0]; c.fy = v->intr_d[1]; c.cx = v->intr_d[2]; c.cy = v->intr_d[3];
  c.fxc = v->intr_rgb[0]; c.fyc = v->intr_rgb[1]; c.cxc = v->intr_rgb[2]; c.cyc = v->intr_rgb[3];
  c.mu = s->prm.mu; c.maxW = s->prm.maxW;
  c.depth = v->depth; c.rgb = v->rgb;
  c.w = v->w; c.h = v->h; c.wc = v->w_rgb; c.hc = v->h_rgb;
  c.stopAtMax = s->prm.stopIntegratingAtMaxW != 0;
}

// IntegrateIntoScene  hash: _CPU.cpp:47-114 ; dense: :319-369
template <class V>
void integrate_t(itm_scene* s, const itm_view* view, itm_render_state* rs) {
  FuseCtx c; make_fuse_ctx(s, view, c);
  V* vox = (V*)s->vba.data();
  float vs = s->prm.voxelSize;
  if (s->cfg.indexType == ITM_INDEX_HASH) {
    ITMO_PARALLEL_FOR
    for (int e = 0; e < rs->noVisibleEntries; ++e) {
      const HashEntry& he = s->hash[rs->visibleIds[e]];
      if (he.ptr < 0) continue;
      int gx = he.px * 8, gy = he.py * 8, gz = he.pz * 8;
      V* blk = vox + (size_t)he.ptr * 512;
      ITMO_STAT(++g_stats.fuse_blocks); ITMO_STAT(g_stats.fuse_voxels_visited += 512);
      for (int z = 0; z < 8; ++z) for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
        int loc = x + y * 8 + z * 64;
        if (c.stopAtMax && blk[loc].w == c.maxW) continue;
        V4f pm = {(float)(gx + x) * vs, (float)(gy + y) * vs, (float)(gz + z) * vs, 1.0f};
        Fuse<V>::run(blk[loc], pm, c);
      }
    }
  } else {
    const int* sz = s->cfg.denseSize; const int* off = s->cfg.denseOffset;
    size_t n = s->numVoxels();
    ITMO_PARALLEL_FOR_STATIC
    for (size_t loc = 0; loc < n; ++loc) {
      int z = (int)(loc / ((size_t)sz[0] * sz[1]));
      int tmp = (int)(loc - (size_t)z * sz[0] * sz[1]);
      int y = tmp / sz[0];
      int x = tmp - y * sz[0];
      if (c.stopAtMax && vox[loc].w == c.maxW) continue;
      V4f pm = {(float)(x + off[0]) * vs, (float)(y + off[1]) * vs, (float)(z + off[2]) * vs, 1.0f};
      Fuse<V>::run(vox[loc], pm, c);
    }
  }
}

// ------------------------------------------------------------------------------------------
// block frustum test: checkBlockVisibility<false> / checkPointVisibility<false>
// DeviceAgnostic/ITMSceneReconstructionEngine.h:243-342.  Corners are visited through
// incremental +-factor updates (the accumulated rounding is part of the behaviour).
// ------------------------------------------------------------------------------------------
inline bool corner_visible(const V4f& p, const float* M, const float* intr, int w, int h) {
  V4f q = mul(M, p);
  if (q.z < 1e-10f) return false;
  float u = intr[0] * q.x / q.z + intr[2];
  float v = intr[1] * q.y / q.z + intr[3];
  return (u >= 0 && u < w && v >= 0 && v < h);
}
bool block_visible(const HashEntry& e, const float* M, const float* intr, float voxelSize, int w, int h) {
  float f = (float)ITM_SDF_BLOCK_SIZE * voxelSize;
  V4f p = {(float)e.px * f, (float)e.py * f, (float)e.pz * f, 1.0f};
  if (corner_visible(p, M, intr, w, h)) return true;   // 0 0 0
  p.z += f; if (corner_visible(p, M, intr, w, h)) return true;   // 0 0 1
  p.y += f; if (corner_visible(p, M, intr, w, h)) return true;   // 0 1 1
  p.x += f; if (corner_visible(p, M, intr, w, h)) return true;   // 1 1 1
  p.z -= f; if (corner_visible(p, M, intr, w, h)) return true;   // 1 1 0
  p.y -= f; if (corner_visible(p, M, intr, w, h)) return true;   // 1 0 0
  p.x -= f; p.y += f; if (corner_visible(p, M, intr, w, h)) return true;   // 0 1 0
  p.x += f; p.y -= f; p.z += f; if (corner_visible(p, M, intr, w, h)) return true;   // 1 0 1
  return false;
}

// checkBlockVisibility<true> (:277-342 with checkPointVisibility<true> :243-275): a corner outside the image but inside the image
// enlarged by an eighth on every side makes the block "visible enlarged"; the walk over the corners ends at the first corner
// that is inside the image proper
bool block_visible_enlarged(const HashEntry& e, const float* M, const float* intr, float voxelSize, int w, int h) {
  float f = (float)ITM_SDF_BLOCK_SIZE * voxelSize;
  V4f p = {(float)e.px * f, (float)e.py * f, (float)e.pz * f, 1.0f};
  bool enlarged = false;
  auto corner = [&](const V4f& c) -> bool {      // returns isVisible
    V4f q = mul(M, c);
    if (q.z < 1e-10f) return false;
    float u = intr[0] * q.x / q.z + intr[2];
    float v = intr[1] * q.y / q.z + intr[3];
    if (u >= 0 && u < w && v >= 0 && v < h) { enlarged = true; return true; }
    const int lx = -w / 8, hx = w + w / 8, ly = -h / 8, hy = h + h / 8;
    if (u >= lx && u < hx && v >= ly && v < hy) enlarged = true;
    return false;
  };
  if (corner(p)) return true;
  p.z += f; if (corner(p)) return true;
  p.y += f; if (corner(p)) return true;
  p.x += f; if (corner(p)) return true;
  p.z -= f; if (corner(p)) return true;
  p.y -= f; if (corner(p)) return true;
  p.x -= f; p.y += f; if (corner(p)) return true;
  p.x += f; p.y -= f; p.z += f; if (corner(p)) return true;
  return enlarged;
}

// ------------------------------------------------------------------------------------------
// AllocateSceneFromDepth (hash)  _CPU.cpp:116-291 with buildHashAllocAndVisibleTypePP
// DeviceAgnostic/ITMSceneReconstructionEngine.h:141-241 (useSwapping == false)
// ------------------------------------------------------------------------------------------
void allocate_hash(itm_scene* s, const itm_view* view, itm_render_state* rs, bool onlyUpdateVisibleList) {
  const int W = view->w, H = view->h;
  const float vs = s->prm.voxelSize, mu = s->prm.mu;
  float invM[16];
  invert4(view->M_d, invM);
  const float ifx = 1.0f / view->intr_d[0], ify = 1.0f / view->intr_d[1];
  const float cx = view->intr_d[2], cy = view->intr_d[3];
  const float oneOverBlock = 1.0f / (vs * ITM_SDF_BLOCK_SIZE);
  const float vfmin = s->prm.viewFrustum_min, vfmax = s->prm.viewFrustum_max;
  const uint32_t mask = (uint32_t)s->cfg.bucketNum - 1u;
  const int BN = s->cfg.bucketNum;
  HashEntry* table = s->hash.data();
  uint8_t* visT = rs->visibleType.data();
  uint8_t* allocT = s->allocType.data();
  int16_t* coords = s->blockCoords.data();

  int lastFreeVBA = s->lastFreeBlockId, lastFreeExcess = s->lastFreeExcessListId;
  std::memset(allocT, 0, (size_t)s->noTotalEntries);
  // (_CPU.cpp:160-161 marks the previous LIST.  When more slots were visible than the list holds the reference has written past its
  // list, SURVEY section 7 trap 6; product and oracle clamp list and count, and then every slot that was visible -- listed or not --
  // counts as "visible in the previous frame": outside the reference's defined behaviour, the same rule on both sides)
  if (rs->rawVisible > rs->capIds) { for (int t = 0; t < s->noTotalEntries; ++t) if (visT[t]) visT[t] = 3; }
  else for (int i = 0; i < rs->noVisibleEntries; ++i) visT[rs->visibleIds[i]] = 3;

  ITMO_PARALLEL_FOR
  for (int loc = 0; loc < W * H; ++loc) {
    int y = loc / W, x = loc - y * W;
    float d = view->depth[x + y * W];
    if (d <= 0 || (d - mu) < 0 || (d - mu) < vfmin || (d + mu) > vfmax) continue;
    V3f p; p.z = d; p.x = p.z * (((float)x - cx) * ifx); p.y = p.z * (((float)y - cy) * ify);
    float norm = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
    V4f a = {p.x * (1.0f - mu / norm), p.y * (1.0f - mu / norm), p.z * (1.0f - mu / norm), 1.0f};
    V4f ta = mul(invM, a);
    V3f pt = {ta.x * oneOverBlock, ta.y * oneOverBlock, ta.z * oneOverBlock};
    V4f b = {p.x * (1.0f + mu / norm), p.y * (1.0f + mu / norm), p.z * (1.0f + mu / norm), 1.0f};
    V4f tb = mul(invM, b);
    V3f pe = {tb.x * oneOverBlock, tb.y * oneOverBlock, tb.z * oneOverBlock};
    V3f dir = {pe.x - pt.x, pe.y - pt.y, pe.z - pt.z};
    norm = std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    int noSteps = (int)std::ceil(2.0f * norm);
    float div = (float)(noSteps - 1);
    dir.x /= div; dir.y /= div; dir.z /= div;
    ITMO_STAT(++g_stats.alloc_pixels); ITMO_STAT(g_stats.alloc_steps += noSteps);
    for (int i = 0; i < noSteps; ++i) {
      int16_t bx = (int16_t)std::floor(pt.x), by = (int16_t)std::floor(pt.y), bz = (int16_t)std::floor(pt.z);
      int idx = hash_index(bx, by, bz, mask);
      bool isFound = false;
      HashEntry he = table[idx];
      ITMO_STAT(++g_stats.alloc_probes);
      if (he.px == bx && he.py == by && he.pz == bz && he.ptr >= -1) {
        visT[idx] = (he.ptr == -1) ? 2 : 1;
        isFound = true;
      }
      if (!isFound) {
        bool isExcess = false;
        if (he.ptr >= -1) {
          while (he.offset >= 1) {
            idx = BN + he.offset - 1;
            he = table[idx];
            ITMO_STAT(++g_stats.alloc_probes);
            if (he.px == bx && he.py == by && he.pz == bz && he.ptr >= -1) {
              visT[idx] = (he.ptr == -1) ? 2 : 1;
              isFound = true;
              break;
            }
          }
          isExcess = true;
        }
        if (!isFound) {
          allocT[idx] = isExcess ? 2 : 1;
          if (!isExcess) visT[idx] = 1;
          coords[idx * 4 + 0] = bx; coords[idx * 4 + 1] = by; coords[idx * 4 + 2] = bz; coords[idx * 4 + 3] = 1;
        }
      }
      pt.x += dir.x; pt.y += dir.y; pt.z += dir.z;
    }
  }

  int requests = 0;
  if (!onlyUpdateVisibleList) {
    for (int t = 0; t < s->noTotalEntries; ++t) {
      uint8_t type = allocT[t];
      if (type == 1) {
        ++requests;
        int vbaIdx = lastFreeVBA; lastFreeVBA--;
        if (vbaIdx >= 0) {
          HashEntry e; std::memset(&e, 0, sizeof e);
          e.px = coords[t * 4]; e.py = coords[t * 4 + 1]; e.pz = coords[t * 4 + 2];
          e.ptr = s->allocList[vbaIdx]; e.offset = 0;
          table[t] = e;
        }
      } else if (type == 2) {
        ++requests;
        int vbaIdx = lastFreeVBA; lastFreeVBA--;
        int exlIdx = lastFreeExcess; lastFreeExcess--;
        if (vbaIdx >= 0 && exlIdx >= 0) {
          HashEntry e; std::memset(&e, 0, sizeof e);
          e.px = coords[t * 4]; e.py = coords[t * 4 + 1]; e.pz = coords[t * 4 + 2];
          e.ptr = s->allocList[vbaIdx]; e.offset = 0;
          int exlOffset = s->excessList[exlIdx];
          table[t].offset = exlOffset + 1;
          table[BN + exlOffset] = e;
          visT[BN + exlOffset] = 1;
        }
      }
    }
  }
  s->noAllocRequests = requests;

  int nv = 0;
  for (int t = 0; t < s->noTotalEntries; ++t) {
    uint8_t vt = visT[t];
    if (vt == 3) {
      if (s->cfg.useSwapping) { if (!block_visible_enlarged(table[t], view->M_d, view->intr_d, vs, W, H)) vt = 0; }
      else if (!block_visible(table[t], view->M_d, view->intr_d, vs, W, H)) vt = 0;
      visT[t] = vt;
    }
    if (s->cfg.useSwapping) { if (vt > 0 && s->swapStates[t] != 2) s->swapStates[t] = 1; }     // _CPU.cpp:250-253
    if (vt > 0) {
      // the reference writes past visibleEntryIDs[SDF_LOCAL_BLOCK_NUM] here; this ABI clamps
      if (nv < rs->capIds) rs->visibleIds[nv] = t;
      nv++;
    }
  }
  // reallocate deleted ones from previous swap operation (_CPU.cpp:271-285)
  if (s->cfg.useSwapping) {
    for (int t = 0; t < s->noTotalEntries; ++t) {
      if (visT[t] > 0 && table[t].ptr == -1) {
        int vbaIdx = lastFreeVBA; lastFreeVBA--;
        if (vbaIdx >= 0) table[t].ptr = s->allocList[vbaIdx];
      }
    }
  }
  rs->noVisibleEntries = (nv < rs->capIds) ? nv : rs->capIds;
  rs->rawVisible = nv;
  s->lastFreeBlockId = lastFreeVBA;
  s->lastFreeExcessListId = lastFreeExcess;
}

// ------------------------------------------------------------------------------------------
// ITMSwappingEngine_CPU (DeviceSpecific/CPU/ITMSwappingEngine_CPU.cpp:21-168) with CombineVoxelInformation
// (DeviceAgnostic/ITMSwappingEngine.h:7-69)
// ------------------------------------------------------------------------------------------
inline uint8_t to_uchar_ref(float x) {            // TO_UCHAR3 per component: round half away from zero, then clamp (ORUtils/Vector.h:245-247, MathUtils.h:21-23)
  int v = (int)round_ref(x);
  v = (v < 255) ? v : 255;
  return (uint8_t)((0 < v) ? v : 0);
}
inline void combine_colour(const VoxS&, VoxS&, int) {}
inline void combine_colour(const VoxF&, VoxF&, int) {}
template <class V>
inline void combine_colour(const V& src, V& dst, int maxW) {     // combineVoxelColorInformation
  int newW = dst.wc, oldW = src.wc;
  float newC[3], oldC[3];
  for (int k = 0; k < 3; ++k) { newC[k] = (float)dst.clr[k] / 255.0f; oldC[k] = (float)src.clr[k] / 255.0f; }
  if (oldW == 0) return;
  for (int k = 0; k < 3; ++k) newC[k] = oldC[k] * (float)oldW + newC[k] * (float)newW;
  newW = oldW + newW;
  for (int k = 0; k < 3; ++k) newC[k] /= (float)newW;
  newW = std::min(newW, maxW);
  for (int k = 0; k < 3; ++k) dst.clr[k] = to_uchar_ref(newC[k] * 255.0f);
  dst.wc = (uint8_t)newW;
}
template <class V>
void combine_voxel(const V& src, V& dst, int maxW) {              // CombineVoxelInformation<hasColor, TVoxel>::compute
  int newW = dst.w, oldW = src.w;
  float newF = Codec<V>::toF(dst.sdf), oldF = Codec<V>::toF(src.sdf);
  if (oldW != 0) {                                                // combineVoxelDepthInformation returns here; the colour part still runs
    newF = oldW * oldF + newW * newF;
    newW = oldW + newW;
    newF /= newW;
    newW = std::min(newW, maxW);
    dst.w = (uint8_t)newW;
    dst.sdf = Codec<V>::toV(newF);
  }
  combine_colour(src, dst, maxW);
}
template <class V>
void swap_integrate_t(itm_scene* s) {
  const int N = s->noTotalEntries, cap = s->transferBlockNum();
  std::vector<int> needed;
  for (int e = 0; e < N && (int)needed.size() < cap; ++e) if (s->swapStates[e] == 1) needed.push_back(e);
  V* vba = (V*)s->vba.data();
  for (int id : needed) {
    if (s->hasStoredData[id] && s->hash[id].ptr >= 0) {          // (the reference dereferences ptr unchecked; a state-1 entry without a block cannot be combined)
      const V* src = (const V*)s->storedBlocks + (size_t)id * 512;
      V* dst = vba + (size_t)s->hash[id].ptr * 512;
      for (int v = 0; v < 512; ++v) combine_voxel(src[v], dst[v], s->prm.maxW);
    }
    s->swapStates[id] = 2;
  }
}
template <class V>
void swap_save_t(itm_scene* s, itm_render_state* rs) {
  const int N = s->noTotalEntries, cap = s->transferBlockNum();
  V* vba = (V*)s->vba.data();
  // (an exhausted pool leaves the counter below -1, where the reference writes in front of the list: counted as -1 = "list empty",
  // as the product does -- outside the reference's defined behaviour)
  int noNeeded = 0, noAllocated = s->lastFreeBlockId;
  for (int e = 0; e < N; ++e) {
    if (noNeeded >= cap) break;
    const int localPtr = s->hash[e].ptr;
    if (s->swapStates[e] == 2 && localPtr >= 0 && rs->visibleType[e] == 0) {
      V* loc = vba + (size_t)localPtr * 512;
      s->hasStoredData[e] = 1;
      std::memcpy(s->storedBlocks + (size_t)e * 512 * sizeof(V), loc, 512 * sizeof(V));
      s->swapStates[e] = 0;
      if (noAllocated < -1) noAllocated = -1;                    // (a call without candidates leaves the counter as it found it)
      const int vbaIdx = noAllocated;
      if (vbaIdx < s->cfg.bucketNum - 1 && vbaIdx + 1 < s->cfg.localBlockNum) {   // (sic: SDF_BUCKET_NUM, ITMSwappingEngine_CPU.cpp:146; the second bound only guards an inconsistent uploaded counter)
        noAllocated++;
        s->allocList[vbaIdx + 1] = localPtr;
        s->hash[e].ptr = -1;
        for (int i = 0; i < 512; ++i) loc[i] = Codec<V>::init();
      }
      noNeeded++;
    }
  }
  s->lastFreeBlockId = noAllocated;
}

// FindVisibleBlocks  DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:39-77
void find_visible_hash(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs) {
  int nv = 0;
  for (int t = 0; t < s->noTotalEntries; ++t) {
    const HashEntry& e = s->hash[t];
    bool vis = false;
    if (e.ptr >= 0) vis = block_visible(e, M, intr, s->prm.voxelSize, rs->w, rs->h);
    if (vis) { if (nv < rs->capIds) rs->visibleIds[nv] = t; nv++; }
  }
  rs->noVisibleEntries = (nv < rs->capIds) ? nv : rs->capIds;
  rs->rawVisible = nv;
}

// ------------------------------------------------------------------------------------------
// CreateExpectedDepths  ITMVisualisationEngine_CPU.cpp:79-91 (dense) / :93-152 (hash) with
// ProjectSingleBlock / CreateRenderingBlocks  DeviceAgnostic/ITMVisualisationEngine.h:28-90
// ------------------------------------------------------------------------------------------
struct RBlock { int16_t ulx, uly, lrx, lry; float z0, z1; };

bool project_block(const HashEntry& e, const float* M, const float* intr, int W, int H, float voxelSize,
                   int& ulx, int& uly, int& lrx, int& lry, float& z0, float& z1) {
  ulx = W / 8; uly = H / 8; lrx = -1; lry = -1; z0 = 999999.9f; z1 = 0.05f;
  for (int corner = 0; corner < 8; ++corner) {
    int16_t tx = (int16_t)(e.px + ((corner & 1) ? 1 : 0));
    int16_t ty = (int16_t)(e.py + ((corner & 2) ? 1 : 0));
    int16_t tz = (int16_t)(e.pz + ((corner & 4) ? 1 : 0));
    V4f p = {(float)tx * (float)ITM_SDF_BLOCK_SIZE * voxelSize, (float)ty * (float)ITM_SDF_BLOCK_SIZE * voxelSize,
             (float)tz * (float)ITM_SDF_BLOCK_SIZE * voxelSize, 1.0f};
    V4f q = mul(M, p);
    if (q.z < 1e-6) continue;   // double literal in the reference
    float u = (intr[0] * q.x / q.z + intr[2]) / 8;
    float v = (intr[1] * q.y / q.z + intr[3]) / 8;
    if (ulx > std::floor(u)) ulx = (int)std::floor(u);
    if (lrx < std::ceil(u)) lrx = (int)std::ceil(u);
    if (uly > std::floor(v)) uly = (int)std::floor(v);
    if (lry < std::ceil(v)) lry = (int)std::ceil(v);
    if (z0 > q.z) z0 = q.z;
    if (z1 < q.z) z1 = q.z;
  }
  if (ulx < 0) ulx = 0;
  if (uly < 0) uly = 0;
  if (lrx >= W) lrx = W - 1;
  if (lry >= H) lry = H - 1;
  if (ulx > lrx) return false;
  if (uly > lry) return false;
  if (z0 < 0.05f) z0 = 0.05f;
  if (z1 < 0.05f) return false;
  return true;
}

void expected_depths(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs) {
  const int W = rs->w, H = rs->h;
  if (s->cfg.indexType == ITM_INDEX_DENSE) {
    for (int i = 0; i < W * H; ++i) { rs->range[i].x = 0.2f; rs->range[i].y = 3.0f; }
    return;
  }
  for (int i = 0; i < W * H; ++i) { rs->range[i].x = 999999.9f; rs->range[i].y = 0.05f; }
  std::vector<RBlock> blocks;
  int count = 0;
  for (int b = 0; b < rs->noVisibleEntries; ++b) {
    const HashEntry& e = s->hash[rs->visibleIds[b]];
    if (e.ptr < 0) continue;
    int ulx, uly, lrx, lry; float z0, z1;
    if (!project_block(e, M, intr, W, H, s->prm.voxelSize, ulx, uly, lrx, lry, z0, z1)) continue;
    int nx = (int)std::ceil((float)(lrx - ulx + 1) / 16.0f);
    int ny = (int)std::ceil((float)(lry - uly + 1) / 16.0f);
    if (count + nx * ny >= s->cfg.maxRenderingBlocks) continue;
    count += nx * ny;
    for (int by = 0; by < ny; ++by) for (int bx = 0; bx < nx; ++bx) {
      RBlock r;
      r.ulx = (int16_t)(ulx + bx * 16); r.uly = (int16_t)(uly + by * 16);
      r.lrx = (int16_t)(ulx + (bx + 1) * 16 - 1); r.lry = (int16_t)(uly + (by + 1) * 16 - 1);
      if (r.lrx > lrx) r.lrx = (int16_t)lrx;
      if (r.lry > lry) r.lry = (int16_t)lry;
      r.z0 = z0; r.z1 = z1;
      blocks.push_back(r);
    }
  }
  rs->noRenderingBlocks = count;
  for (const RBlock& r : blocks)
    for (int y = r.uly; y <= r.lry; ++y) for (int x = r.ulx; x <= r.lrx; ++x) {
      V2f& px = rs->range[x + y * W];
      if (px.x > r.z0) px.x = r.z0;
      if (px.y < r.z1) px.y = r.z1;
    }
}

// ------------------------------------------------------------------------------------------
// castRay  DeviceAgnostic/ITMVisualisationEngine.h:92-158 ; GenericRaycast  _CPU.cpp:154-188
// ------------------------------------------------------------------------------------------
template <class V>
bool cast_ray(V4f& out, int x, int y, const Reader<V>& rd, const float* invM, float ifx, float ify, float cx, float cy,
              float oneOverVoxel, float mu, const V2f& mm) {
  float sdf = 1.0f;
  float stepScale = mu * oneOverVoxel;
  V4f pc;
  pc.z = mm.x; pc.x = pc.z * (((float)x - cx) * ifx); pc.y = pc.z * (((float)y - cy) * ify); pc.w = 1.0f;
  float acc = 0; acc += pc.x * pc.x; acc += pc.y * pc.y; acc += pc.z * pc.z;   // dot(), ORUtils/Vector.h:811-816
  float total = std::sqrt(acc) * oneOverVoxel;
  V4f t = mul(invM, pc);
  V3f S = {t.x * oneOverVoxel, t.y * oneOverVoxel, t.z * oneOverVoxel};
  pc.z = mm.y; pc.x = pc.z * (((float)x - cx) * ifx); pc.y = pc.z * (((float)y - cy) * ify); pc.w = 1.0f;
  acc = 0; acc += pc.x * pc.x; acc += pc.y * pc.y; acc += pc.z * pc.z;
  float totalMax = std::sqrt(acc) * oneOverVoxel;
  t = mul(invM, pc);
  V3f E = {t.x * oneOverVoxel, t.y * oneOverVoxel, t.z * oneOverVoxel};
  V3f dir = {E.x - S.x, E.y - S.y, E.z - S.z};
  float dn = 1.0f / std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  dir.x *= dn; dir.y *= dn; dir.z *= dn;
  V3f pt = S;
  BlockCache cache;
  bool found;
  float step;
  long long steps = 0;
  int bandSteps = 0, missSteps = 0, unitNear = 0, unitBand = 0, farSteps = 0, missRun = 0, longestMissRun = 0, afterRun = 0;
  while (total < totalMax) {
    ++steps;
    sdf = rd.nearest(pt, found, cache);
    if (!found) {
      step = (float)ITM_SDF_BLOCK_SIZE; ++missSteps;
      if (++missRun > longestMissRun) { longestMissRun = missRun; afterRun = 0; }
    } else {
      missRun = 0; ++afterRun;
      bool band = false;
      if (sdf == 1.0f) ++farSteps;
      if ((sdf <= 0.1f) && (sdf >= -0.5f)) { sdf = rd.trilinear(pt, found, cache); ++bandSteps; band = true; }
      if (sdf <= 0.0f) break;
      step = fmax_ref(sdf * stepScale, 1.0f);
      if (step == 1.0f) { if (band) ++unitBand; else ++unitNear; }
    }
    pt.x += step * dir.x; pt.y += step * dir.y; pt.z += step * dir.z;
    total += step;
  }
  ITMO_STAT(++g_stats.rays); ITMO_STAT(g_stats.ray_steps += steps);
  if (g_rayTrace) { int* r = g_rayTrace + 8 * (x + y * g_rayTraceW); r[0] = (int)steps; r[1] = bandSteps; r[2] = missSteps; r[3] = unitNear; r[4] = unitBand; r[5] = farSteps; r[6] = longestMissRun; r[7] = afterRun; }
  ITMO_STAT(g_stats.max_ray_steps = (steps > g_stats.max_ray_steps) ? steps : g_stats.max_ray_steps);
  bool hit;
  if (sdf <= 0.0f) {
    ITMO_STAT(++g_stats.ray_hits);
    step = sdf * stepScale;
    pt.x += step * dir.x; pt.y += step * dir.y; pt.z += step * dir.z;
    sdf = rd.trilinear(pt, found, cache);
    step = sdf * stepScale;
    pt.x += step * dir.x; pt.y += step * dir.y; pt.z += step * dir.z;
    hit = true;
  } else hit = false;
  out.x = pt.x; out.y = pt.y; out.z = pt.z; out.w = hit ? 1.0f : 0.0f;
  return hit;
}

template <class V>
void raycast_t(const itm_scene* s, itm_render_state* rs, const float* invM, const float* intr, V4f* dst) {
  Reader<V> rd(s);
  const int W = rs->w, H = rs->h;
  float ifx = 1.0f / intr[0], ify = 1.0f / intr[1];
  float oov = 1.0f / s->prm.voxelSize;
  ITMO_PARALLEL_FOR
  for (int loc = 0; loc < W * H; ++loc) {
    int y = loc / W, x = loc - y * W;
    int loc2 = (int)std::floor((float)x / 8) + (int)std::floor((float)y / 8) * W;
    cast_ray<V>(dst[loc], x, y, rd, invM, ifx, ify, intr[2], intr[3], oov, s->prm.mu, rs->range[loc2]);
  }
}

// ------------------------------------------------------------------------------------------
// shading helpers  DeviceAgnostic/ITMVisualisationEngine.h:175-279
// ------------------------------------------------------------------------------------------
inline uint32_t grey_px(float angle) {   // drawPixelGrey :256-260
  float o = (0.8f * angle + 0.2f) * 255.0f;
  uint32_t g = (uint8_t)o;
  return g | (g << 8) | (g << 16) | (g << 24);
}

// computeNormalAndAngle<useSmoothing=true> from the ray-hit map :191-254
bool normal_from_map(const V4f* rays, int x, int y, int W, int H, float voxelSize, const V3f& L, V3f& n, float& angle) {
  if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) return false;
  V4f xp = rays[(x + 2) + y * W], yp = rays[x + (y + 2) * W];
  V4f xm = rays[(x - 2) + y * W], ym = rays[x + (y - 2) * W];
  V4f dx = {0, 0, 0, 0}, dy = {0, 0, 0, 0};
  bool plus1 = false;
  if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) plus1 = true;
  else {
    dx = {xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w};
    dy = {yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w};
    float l = fmax_ref(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z, dy.x * dy.x + dy.y * dy.y + dy.z * dy.z);
    if (l * voxelSize * voxelSize > (0.15f * 0.15f)) plus1 = true;
  }
  if (plus1) {
    xp = rays[(x + 1) + y * W]; yp = rays[x + (y + 1) * W];
    xm = rays[(x - 1) + y * W]; ym = rays[x + (y - 1) * W];
    dx = {xp.x - xm.x, xp.y - xm.y, xp.z - xm.z, xp.w - xm.w};
    dy = {yp.x - ym.x, yp.y - ym.y, yp.z - ym.z, yp.w - ym.w};
    if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) return false;
  }
  n.x = -(dx.y * dy.z - dx.z * dy.y);
  n.y = -(dx.z * dy.x - dx.x * dy.z);
  n.z = -(dx.x * dy.y - dx.y * dy.x);
  float sc = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
  n.x *= sc; n.y *= sc; n.z *= sc;
  angle = n.x * L.x + n.y * L.y + n.z * L.z;
  return angle > 0.0;
}

// computeSingleNormalFromSDF  DeviceAgnostic/ITMRepresentationAccess.h:224-337 (32 uncached reads)
template <class V>
V3f sdf_gradient(const Reader<V>& rd, const V3f& p) {
  bool f;
  float flx = std::floor(p.x), fly = std::floor(p.y), flz = std::floor(p.z);
  float cx = p.x - flx, cy = p.y - fly, cz = p.z - flz;
  int ix = (int)flx, iy = (int)fly, iz = (int)flz;
  float nx = 1.0f - cx, ny = 1.0f - cy, nz = 1.0f - cz;
  auto R = [&](int dx, int dy, int dz) { return (float)rd.read(ix + dx, iy + dy, iz + dz, f).sdf; };
  float f000 = R(0, 0, 0), f100 = R(1, 0, 0), f010 = R(0, 1, 0), f110 = R(1, 1, 0);
  float f001 = R(0, 0, 1), f101 = R(1, 0, 1), f011 = R(0, 1, 1), f111 = R(1, 1, 1);
  V3f g; float p1, p2, v1, a, b, c, d;
  // x
  p1 = f000 * ny * nz + f010 * cy * nz + f001 * ny * cz + f011 * cy * cz;
  a = R(-1, 0, 0); b = R(-1, 1, 0); c = R(-1, 0, 1); d = R(-1, 1, 1);
  p2 = a * ny * nz + b * cy * nz + c * ny * cz + d * cy * cz;
  v1 = p1 * cx + p2 * nx;
  p1 = f100 * ny * nz + f110 * cy * nz + f101 * ny * cz + f111 * cy * cz;
  a = R(2, 0, 0); b = R(2, 1, 0); c = R(2, 0, 1); d = R(2, 1, 1);
  p2 = a * ny * nz + b * cy * nz + c * ny * cz + d * cy * cz;
  g.x = Codec<V>::toF(p1 * nx + p2 * cx - v1);
  // y
  p1 = f000 * nx * nz + f100 * cx * nz + f001 * nx * cz + f101 * cx * cz;
  a = R(0, -1, 0); b = R(1, -1, 0); c = R(0, -1, 1); d = R(1, -1, 1);
  p2 = a * nx * nz + b * cx * nz + c * nx * cz + d * cx * cz;
  v1 = p1 * cy + p2 * ny;
  p1 = f010 * nx * nz + f110 * cx * nz + f011 * nx * cz + f111 * cx * cz;
  a = R(0, 2, 0); b = R(1, 2, 0); c = R(0, 2, 1); d = R(1, 2, 1);
  p2 = a * nx * nz + b * cx * nz + c * nx * cz + d * cx * cz;
  g.y = Codec<V>::toF(p1 * ny + p2 * cy - v1);
  // z
  p1 = f000 * nx * ny + f100 * cx * ny + f010 * nx * cy + f110 * cx * cy;
  a = R(0, 0, -1); b = R(1, 0, -1); c = R(0, 1, -1); d = R(1, 1, -1);
  p2 = a * nx * ny + b * cx * ny + c * nx * cy + d * cx * cy;
  v1 = p1 * cz + p2 * nz;
  p1 = f001 * nx * ny + f101 * cx * ny + f011 * nx * cy + f111 * cx * cy;
  a = R(0, 0, 2); b = R(1, 0, 2); c = R(0, 1, 2); d = R(1, 1, 2);
  p2 = a * nx * ny + b * cx * ny + c * nx * cy + d * cx * cy;
  g.z = Codec<V>::toF(p1 * nz + p2 * cz - v1);
  return g;
}

// computeNormalAndAngle<TVoxel,TIndex> (SDF gradient variant) :175-189
template <class V>
bool normal_from_sdf(const Reader<V>& rd, const V3f& p, const V3f& L, V3f& n, float& angle) {
  n = sdf_gradient<V>(rd, p);
  float sc = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
  n.x *= sc; n.y *= sc; n.z *= sc;
  angle = n.x * L.x + n.y * L.y + n.z * L.z;
  return angle > 0.0;
}

// readFromSDF_color4u_interpolated  DeviceAgnostic/ITMRepresentationAccess.h:187-222
template <class V, bool C = Codec<V>::color> struct Colour;
template <class V> struct Colour<V, false> {
  static V4f at(const Reader<V>&, const V3f&) { return V4f{0, 0, 0, 0}; }
};
template <class V> struct Colour<V, true> {
  static V4f at(const Reader<V>& rd, const V3f& p) {
    BlockCache cache; bool f;
    float flx = std::floor(p.x), fly = std::floor(p.y), flz = std::floor(p.z);
    float cx = p.x - flx, cy = p.y - fly, cz = p.z - flz;
    int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    float r[3] = {0, 0, 0};
    auto add = [&](int dx, int dy, int dz, float wgt) {
      V v = rd.read(ix + dx, iy + dy, iz + dz, f, cache);
      for (int k = 0; k < 3; ++k) r[k] += wgt * (float)v.clr[k];
    };
    add(0, 0, 0, (1.0f - cx) * (1.0f - cy) * (1.0f - cz));
    add(1, 0, 0, (cx) * (1.0f - cy) * (1.0f - cz));
    add(0, 1, 0, (1.0f - cx) * (cy) * (1.0f - cz));
    add(1, 1, 0, (cx) * (cy) * (1.0f - cz));
    add(0, 0, 1, (1.0f - cx) * (1.0f - cy) * cz);
    add(1, 0, 1, (cx) * (1.0f - cy) * cz);
    add(0, 1, 1, (1.0f - cx) * (cy)*cz);
    add(1, 1, 1, (cx) * (cy)*cz);
    return V4f{r[0] / 255.0f, r[1] / 255.0f, r[2] / 255.0f, 255.0f / 255.0f};
  }
};

// RenderImage_common  _CPU.cpp:190-240 with processPixelGrey/Colour/Normal :368-409
template <class V>
void render_image_t(const itm_scene* s, itm_render_state* rs, const float* M, const float* intr, uint32_t* out, int type) {
  float invM[16]; invert4(M, invM);
  raycast_t<V>(s, rs, invM, intr, rs->raycast.data());
  V3f L = {-invM[8], -invM[9], -invM[10]};
  Reader<V> rd(s);
  if (type == ITM_RENDER_COLOUR_FROM_VOLUME && !Codec<V>::color) type = ITM_RENDER_SHADED_GREYSCALE;
  ITMO_PARALLEL_FOR
  for (int loc = 0; loc < rs->w * rs->h; ++loc) {
    V4f r = rs->raycast[loc];
    V3f p = {r.x, r.y, r.z};
    bool found = r.w > 0;
    V3f n; float angle = 0;
    if (found) found = normal_from_sdf<V>(rd, p, L, n, angle);
    if (!found) { out[loc] = 0; continue; }
    if (type == ITM_RENDER_COLOUR_FROM_VOLUME) {
      V4f c = Colour<V>::at(rd, p);
      uint32_t px = (uint32_t)(uint8_t)(c.x * 255.0f) | ((uint32_t)(uint8_t)(c.y * 255.0f) << 8) |
                    ((uint32_t)(uint8_t)(c.z * 255.0f) << 16) | (255u << 24);
      out[loc] = px;
    } else if (type == ITM_RENDER_COLOUR_FROM_NORMAL) {
      // drawPixelNormal :262-267 writes r,g,b only; the alpha byte keeps its previous value
      uint32_t prev = out[loc] & 0xff000000u;
      uint32_t px = (uint32_t)(uint8_t)((0.3f + (-n.x + 1.0f) * 0.35f) * 255.0f) |
                    ((uint32_t)(uint8_t)((0.3f + (-n.y + 1.0f) * 0.35f) * 255.0f) << 8) |
                    ((uint32_t)(uint8_t)((0.3f + (-n.z + 1.0f) * 0.35f) * 255.0f) << 16) | prev;
      out[loc] = px;
    } else {
      out[loc] = grey_px(angle);
    }
  }
}

// CreateICPMaps_common  _CPU.cpp:266-287 with processPixelICP<true> :314-349
template <class V>
void icp_maps_t(const itm_scene* s, const itm_view* view, itm_render_state* rs, V4f* points, V4f* normals) {
  float invM[16]; invert4(view->M_d, invM);
  raycast_t<V>(s, rs, invM, view->intr_d, rs->raycast.data());
  V3f L = {-invM[8], -invM[9], -invM[10]};
  const int W = rs->w, H = rs->h;
  const float vs = s->prm.voxelSize;
  const V4f* rays = rs->raycast.data();
  ITMO_PARALLEL_FOR
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    int loc = x + y * W;
    V4f r = rays[loc];
    bool found = r.w > 0.0f;
    V3f n; float angle = 0;
    if (found) found = normal_from_map(rays, x, y, W, H, vs, L, n, angle);
    if (found) {
      rs->image[loc] = grey_px(angle);
      points[loc] = V4f{r.x * vs, r.y * vs, r.z * vs, 1.0f};
      normals[loc] = V4f{n.x, n.y, n.z, 0.0f};
    } else {
      points[loc] = V4f{0, 0, 0, -1.0f};
      normals[loc] = V4f{0, 0, 0, -1.0f};
      rs->image[loc] = 0;
    }
  }
}

// ForwardRender_common  _CPU.cpp:289-354 ; forwardProjectPixel  DeviceAgnostic/ITMVisualisationEngine.h:160-173
template <class V>
void forward_render_t(const itm_scene* s, const itm_view* view, itm_render_state* rs) {
  const int W = rs->w, H = rs->h;
  float invM[16]; invert4(view->M_d, invM);
  const float* M = view->M_d; const float* ip = view->intr_d;
  float ifx = 1.0f / ip[0], ify = 1.0f / ip[1];
  V3f L = {-invM[8], -invM[9], -invM[10]};
  const float vs = s->prm.voxelSize;
  Reader<V> rd(s);
  std::memset(rs->fwdProj.data(), 0, rs->fwdProj.size() * sizeof(V4f));
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    int loc = x + y * W;
    V4f px = rs->raycast[loc];
    V4f p = {px.x * vs, px.y * vs, px.z * vs, 1};
    V4f q = mul(M, p);
    float u = ip[0] * q.x / q.z + ip[2];
    float v = ip[1] * q.y / q.z + ip[3];
    if ((u < 0) || (u > W - 1) || (v < 0) || (v > H - 1)) continue;
    int locNew = (int)(u + 0.5f) + (int)(v + 0.5f) * W;
    if (locNew >= 0) rs->fwdProj[locNew] = px;
  }
  int nMissing = 0;
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    int loc = x + y * W;
    int loc2 = (int)std::floor((float)x / 8) + (int)std::floor((float)y / 8) * W;
    V4f fp = rs->fwdProj[loc]; V2f mm = rs->range[loc2]; float d = view->depth[loc];
    if ((fp.w <= 0) && ((fp.x == 0 && fp.y == 0 && fp.z == 0) || (d >= 0)) && (mm.x < mm.y)) rs->missing[nMissing++] = loc;
  }
  rs->noFwdProjMissingPoints = nMissing;
  float oov = 1.0f / vs;
  for (int i = 0; i < nMissing; ++i) {
    int loc = rs->missing[i];
    int y = loc / W, x = loc - y * W;
    int loc2 = (int)std::floor((float)x / 8) + (int)std::floor((float)y / 8) * W;
    cast_ray<V>(rs->fwdProj[loc], x, y, rd, invM, ifx, ify, ip[2], ip[3], oov, s->prm.mu, rs->range[loc2]);
  }
  const V4f* rays = rs->fwdProj.data();
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    int loc = x + y * W;
    bool found = rays[loc].w > 0.0f;
    V3f n; float angle = 0;
    if (found) found = normal_from_map(rays, x, y, W, H, vs, L, n, angle);
    rs->image[loc] = found ? grey_px(angle) : 0u;
  }
}

// CreatePointCloud_common  _CPU.cpp:242-264 ; RenderPointCloud :424-462
template <class V>
void point_cloud_t(const itm_scene* s, const itm_view* view, itm_render_state* rs, bool skipPoints, V4f* locations, V4f* colours) {
  float invMd[16], invM[16];
  invert4(view->M_d, invMd);
  matmul4(invMd, view->rgb_to_depth, invM);   // pose_d->GetInvM() * calib
  raycast_t<V>(s, rs, invM, view->intr_rgb, rs->raycast.data());
  V3f L = {-invM[8], -invM[9], -invM[10]};
  Reader<V> rd(s);
  const int W = rs->w, H = rs->h; const float vs = s->prm.voxelSize;
  int total = 0;
  for (int y = 0, loc = 0; y < H; ++y) for (int x = 0; x < W; ++x, ++loc) {
    V4f r = rs->raycast[loc];
    V3f p = {r.x, r.y, r.z};
    bool found = r.w > 0;
    V3f n; float angle = 0;
    if (found) found = normal_from_sdf<V>(rd, p, L, n, angle);
    rs->image[loc] = found ? grey_px(angle) : 0u;
    if (skipPoints && ((x % 2 == 0) || (y % 2 == 0))) found = false;
    if (found) {
      V4f c = Colour<V>::at(rd, p);
      if (c.w > 0.0f) { c.x /= c.w; c.y /= c.w; c.z /= c.w; c.w = 1.0f; }
      colours[total] = c;
      locations[total] = V4f{p.x * vs, p.y * vs, p.z * vs, 1.0f};
      total++;
    }
  }
  rs->noTotalPoints = total;
}


// ------------------------------------------------------------------------------------------
// MeshScene   DeviceSpecific/CPU/ITMMeshingEngine_CPU.cpp:19-58 ; findPointNeighbors / sdfInterp / buildVertList
// DeviceAgnostic/ITMMeshingEngine.h:153-231 ; ITMMesh Objects/ITMMesh.h.  The case table is the data file generated from the
// reference (infinitam_amd/csrc/mc_tables.h); the sequential loop, the early exits and the full-buffer behaviour are restated.
// ------------------------------------------------------------------------------------------
template <class V>
void mesh_scene_t(const itm_scene* s, itm_mesh* m) {
  Reader<V> rd(s);
  const float factor = s->prm.voxelSize;
  std::fill(m->tri.begin(), m->tri.end(), 0.0f);
  uint32_t n = 0;
  static const int off[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
  static const int edgeEnds[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  for (int entry = 0; entry < s->noTotalEntries; ++entry) {
    const HashEntry& he = s->hash[entry];
    if (he.ptr < 0) continue;
    const int gx = he.px * 8, gy = he.py * 8, gz = he.pz * 8;
    for (int z = 0; z < 8; ++z) for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
      float p[8][3], v[8];
      bool ok = true;
      for (int k = 0; k < 8 && ok; ++k) {
        const int qx = gx + x + off[k][0], qy = gy + y + off[k][1], qz = gz + z + off[k][2];
        bool found;
        const V vox = rd.read(qx, qy, qz, found);
        p[k][0] = (float)qx; p[k][1] = (float)qy; p[k][2] = (float)qz;
        v[k] = Codec<V>::toF((float)vox.sdf);
        if (!found || v[k] == 1.0f) ok = false;
      }
      if (!ok) continue;
      int cube = 0;
      for (int k = 0; k < 8; ++k) if (v[k] < 0) cube |= 1 << k;
      int edges = 0;
      for (int e = 0; e < 12; ++e) if (((cube >> edgeEnds[e][0]) ^ (cube >> edgeEnds[e][1])) & 1) edges |= 1 << e;
      if (edges == 0) continue;
      float vert[12][3];
      for (int e = 0; e < 12; ++e) {
        if (!(edges & (1 << e))) continue;
        const float* p1 = p[edgeEnds[e][0]]; const float* p2 = p[edgeEnds[e][1]];
        const float v1 = v[edgeEnds[e][0]], v2 = v[edgeEnds[e][1]];
        const float* pick = nullptr;
        if (std::fabs(0.0f - v1) < 0.00001f) pick = p1;
        else if (std::fabs(0.0f - v2) < 0.00001f) pick = p2;
        else if (std::fabs(v1 - v2) < 0.00001f) pick = p1;
        if (pick) { vert[e][0] = pick[0]; vert[e][1] = pick[1]; vert[e][2] = pick[2]; }
        else {
          const float t = (0.0f - v1) / (v2 - v1);
          for (int c = 0; c < 3; ++c) vert[e][c] = p1[c] + t * (p2[c] - p1[c]);
        }
      }
      for (uint64_t l = itm::kTriangleCases[cube]; (l & 0xf) != 0xf; l >>= 12) {
        float* o = &m->tri[(size_t)n * 9];
        for (int k = 0; k < 3; ++k) {
          const int e = (int)((l >> (4 * k)) & 0xf);
          o[3 * k] = vert[e][0] * factor; o[3 * k + 1] = vert[e][1] * factor; o[3 * k + 2] = vert[e][2] * factor;
        }
        if (n < m->maxTriangles - 1) ++n;
      }
    }
  }
  m->noTotalTriangles = n;
}

template <class F>
int dispatch_voxel(int voxelType, F&& f) {
  switch (voxelType) {
    case ITM_VOXEL_S: f((VoxS*)nullptr); return ITM_OK;
    case ITM_VOXEL_F: f((VoxF*)nullptr); return ITM_OK;
    case ITM_VOXEL_S_RGB: f((VoxSC*)nullptr); return ITM_OK;
    case ITM_VOXEL_F_RGB: f((VoxFC*)nullptr); return ITM_OK;
  }
  return fail(ITM_ERR_INVALID, "unknown voxel type");
}
#define VOX_T typename std::remove_pointer<decltype(tag)>::type

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

const char* itmo_version(void) { return "itm-oracle 1 (cpu restatement, test infrastructure)"; }
const char* itmo_last_error(void) { return g_err.c_str(); }
int itmo_uses_device_memory(void) { return 0; }
size_t itmo_voxel_size_bytes(int t) {
  switch (t) { case ITM_VOXEL_S: return 4; case ITM_VOXEL_F: return 8; case ITM_VOXEL_S_RGB: return 8; case ITM_VOXEL_F_RGB: return 12; }
  return 0;
}

int itmo_dev_malloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? ITM_OK : fail(ITM_ERR_DEVICE, "malloc"); }
int itmo_dev_free(void* p) { std::free(p); return ITM_OK; }
int itmo_memcpy_h2d(void* d, const void* s, size_t n, itm_stream) { std::memcpy(d, s, n); return ITM_OK; }
int itmo_memcpy_d2h(void* d, const void* s, size_t n, itm_stream) { std::memcpy(d, s, n); return ITM_OK; }
int itmo_stream_synchronize(itm_stream) { return ITM_OK; }
int itmo_set_device(int) { return ITM_OK; }
int itmo_debug_set(int, int) { return ITM_OK; }
int itmo_debug_divide(int mode, const float* a, const float* b, const float*, float* out, int n, itm_stream) { for (int i = 0; i < n; ++i) out[i] = (mode == 3) ? 1.0f / b[i] : a[i] / b[i]; return ITM_OK; }
int itmo_debug_div32767(const float* in, float* out, int n, itm_stream) { for (int i = 0; i < n; ++i) out[i] = in[i] / 32767.0f; return ITM_OK; }
int itmo_profile_enable(itm_scene*, uint32_t) { return ITM_OK; }
int itmo_profile_sample(itm_scene*, int) { return ITM_OK; }
int itmo_profile_calibrate(itm_scene*, int, itm_stream) { return ITM_OK; }
int itmo_profile_read(itm_scene*, itm_profile* out, int) { if (out) std::memset(out, 0, sizeof *out); return ITM_OK; }

int itmo_scene_create(const itm_scene_config* cfg_in, const itm_scene_params* prm, itm_scene** out) {
  if (!cfg_in || !prm || !out) return fail(ITM_ERR_INVALID, "null argument");
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
  if (cfg.indexType == ITM_INDEX_HASH && (cfg.bucketNum + cfg.excessNum) % 8 != 0) return fail(ITM_ERR_INVALID, "bucketNum + excessNum must be a multiple of 8");
  if (cfg.bucketNum & (cfg.bucketNum - 1)) return fail(ITM_ERR_INVALID, "bucketNum must be a power of two");
  size_t vb = itmo_voxel_size_bytes(cfg.voxelType);
  if (!vb) return fail(ITM_ERR_INVALID, "unknown voxel type");
  if (cfg.indexType != ITM_INDEX_HASH && cfg.indexType != ITM_INDEX_DENSE) return fail(ITM_ERR_INVALID, "unknown index type");
  itm_scene* s = new (std::nothrow) itm_scene();
  if (!s) return fail(ITM_ERR_DEVICE, "out of memory");
  s->cfg = cfg; s->prm = *prm; s->voxBytes = vb;
  s->noTotalEntries = cfg.bucketNum + cfg.excessNum;
  s->vba.assign(s->numVoxels() * vb, 0);
  if (cfg.indexType == ITM_INDEX_HASH) {
    s->hash.assign(s->noTotalEntries, HashEntry{0, 0, 0, 0, 0, 0});
    s->excessList.assign(cfg.excessNum, 0);
    s->allocList.assign(cfg.localBlockNum, 0);
    s->allocType.assign(s->noTotalEntries, 0);
    s->blockCoords.assign((size_t)s->noTotalEntries * 4, 0);
    if (cfg.useSwapping) {
      s->swapStates.assign(s->noTotalEntries, 0);
      s->hasStoredData.assign(s->noTotalEntries, 0);
      s->storedBlocks = (uint8_t*)std::calloc((size_t)s->noTotalEntries * 512, vb);
      if (!s->storedBlocks) { delete s; return fail(ITM_ERR_DEVICE, "out of memory (global cache)"); }
    }
  } else {
    s->allocList.assign(1, 0);
  }
  *out = s;
  return ITM_OK;
}
int itmo_scene_destroy(itm_scene* s) { delete s; return ITM_OK; }
int itmo_scene_get_config(const itm_scene* s, itm_scene_config* c, itm_scene_params* p) {
  if (!s) return fail(ITM_ERR_INVALID, "null scene");
  if (c) *c = s->cfg;
  if (p) *p = s->prm;
  return ITM_OK;
}

int itmo_reset_scene(itm_scene* s, itm_stream) {
  if (!s) return fail(ITM_ERR_INVALID, "null scene");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { reset_scene_t<VOX_T>(s); });
}

int itmo_render_state_create(const itm_scene* s, int w, int h, itm_render_state** out) {
  if (!s || !out || w <= 0 || h <= 0) return fail(ITM_ERR_INVALID, "bad argument");
  itm_render_state* rs = new itm_render_state();
  rs->w = w; rs->h = h; rs->hash = s->cfg.indexType == ITM_INDEX_HASH;
  size_t P = (size_t)w * h;
  rs->range.assign(P, V2f{s->prm.viewFrustum_min, s->prm.viewFrustum_max});
  rs->raycast.assign(P, V4f{0, 0, 0, 0});
  rs->fwdProj.assign(P, V4f{0, 0, 0, 0});
  rs->missing.assign(P, 0);
  rs->image.assign(P, 0);
  rs->capIds = rs->hash ? s->cfg.localBlockNum : 0;
  if (rs->hash) {
    rs->visibleIds.assign(s->cfg.localBlockNum, 0);
    rs->visibleType.assign(s->noTotalEntries, 0);
  }
  *out = rs;
  return ITM_OK;
}
int itmo_render_state_destroy(itm_render_state* rs) { delete rs; return ITM_OK; }

int itmo_allocate_scene_from_depth(itm_scene* s, const itm_view* v, itm_render_state* rs, int onlyVis, itm_stream) {
  if (!s || !v || !rs) return fail(ITM_ERR_INVALID, "null argument");
  if (s->cfg.indexType == ITM_INDEX_DENSE) return ITM_OK;
  if (v->w != rs->w || v->h != rs->h) return fail(ITM_ERR_INVALID, "view / render state size mismatch");
  allocate_hash(s, v, rs, onlyVis != 0);
  return ITM_OK;
}

int itmo_integrate_into_scene(itm_scene* s, const itm_view* v, itm_render_state* rs, itm_stream) {
  if (!s || !v || !rs) return fail(ITM_ERR_INVALID, "null argument");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { integrate_t<VOX_T>(s, v, rs); });
}

// ITMSwappingEngine<TVoxel, ITMVoxelBlockHash>::IntegrateGlobalIntoLocal / SaveToGlobalMemory and the ITMGlobalCache accessors
int itmo_swap_integrate_global_into_local(itm_scene* s, itm_render_state* rs, itm_stream) {
  if (!s || !rs || !s->cfg.useSwapping) return fail(ITM_ERR_INVALID, "scene without swapping");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { swap_integrate_t<VOX_T>(s); });
}
int itmo_swap_save_to_global_memory(itm_scene* s, itm_render_state* rs, itm_stream) {
  if (!s || !rs || !s->cfg.useSwapping) return fail(ITM_ERR_INVALID, "scene without swapping");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { swap_save_t<VOX_T>(s, rs); });
}
int itmo_global_cache_get(const itm_scene* s, int entry, void* dst, int* has) {
  if (!s || !s->cfg.useSwapping || entry < 0 || entry >= s->noTotalEntries || !has) return fail(ITM_ERR_INVALID, "bad argument");
  *has = s->hasStoredData[entry];
  if (*has && dst) std::memcpy(dst, s->storedBlocks + (size_t)entry * 512 * s->voxBytes, 512 * s->voxBytes);
  return ITM_OK;
}
int itmo_global_cache_flags(const itm_scene* s, uint8_t* dst, size_t bytes) {
  if (!s || !s->cfg.useSwapping || !dst || bytes > (size_t)s->noTotalEntries) return fail(ITM_ERR_INVALID, "bad argument");
  std::memcpy(dst, s->hasStoredData.data(), bytes);
  return ITM_OK;
}

int itmo_find_visible_blocks(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream) {
  if (!s || !rs) return fail(ITM_ERR_INVALID, "null argument");
  if (s->cfg.indexType == ITM_INDEX_HASH) find_visible_hash(s, M, intr, rs);
  return ITM_OK;
}

int itmo_create_expected_depths(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream) {
  if (!s || !rs) return fail(ITM_ERR_INVALID, "null argument");
  expected_depths(s, M, intr, rs);
  return ITM_OK;
}

int itmo_render_image(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, uint8_t* out, int type, itm_stream) {
  if (!s || !rs) return fail(ITM_ERR_INVALID, "null argument");
  uint32_t* dst = out ? (uint32_t*)out : rs->image.data();
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { render_image_t<VOX_T>(s, rs, M, intr, dst, type); });
}

int itmo_find_surface(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream) {
  if (!s || !rs) return fail(ITM_ERR_INVALID, "null argument");
  float invM[16]; invert4(M, invM);
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { raycast_t<VOX_T>(s, rs, invM, intr, rs->raycast.data()); });
}

int itmo_create_point_cloud(const itm_scene* s, const itm_view* v, itm_render_state* rs, int skip, float* loc, float* col, itm_stream) {
  if (!s || !v || !rs || !loc || !col) return fail(ITM_ERR_INVALID, "null argument");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { point_cloud_t<VOX_T>(s, v, rs, skip != 0, (V4f*)loc, (V4f*)col); });
}

int itmo_create_icp_maps(const itm_scene* s, const itm_view* v, itm_render_state* rs, float* pts, float* nrm, itm_stream) {
  if (!s || !v || !rs || !pts || !nrm) return fail(ITM_ERR_INVALID, "null argument");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { icp_maps_t<VOX_T>(s, v, rs, (V4f*)pts, (V4f*)nrm); });
}

int itmo_forward_render(const itm_scene* s, const itm_view* v, itm_render_state* rs, itm_stream) {
  if (!s || !v || !rs) return fail(ITM_ERR_INVALID, "null argument");
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { forward_render_t<VOX_T>(s, v, rs); });
}

int itmo_process_frame(itm_scene* s, const itm_view* v, itm_render_state* rs, float* pts, float* nrm, itm_stream st) {
  int r;
  if ((r = itmo_allocate_scene_from_depth(s, v, rs, 0, st))) return r;
  if ((r = itmo_integrate_into_scene(s, v, rs, st))) return r;
  if ((r = itmo_create_expected_depths(s, v->M_d, v->intr_d, rs, st))) return r;
  return itmo_create_icp_maps(s, v, rs, pts, nrm, st);
}

// (the look-ahead of the product's entry point is a launch-level matter: the results are those of the frame alone)
int itmo_process_frame_ahead(itm_scene* s, const itm_view* v, const itm_view*, itm_render_state* rs, float* pts, float* nrm, itm_stream st) {
  return itmo_process_frame(s, v, rs, pts, nrm, st);
}
// (recording calls and requests issued ahead are launch-level matters of the product: nothing is ever pending here)
int itmo_flush(itm_scene*, itm_render_state*, itm_stream) { return ITM_OK; }
int itmo_cancel_ahead(itm_scene*, itm_render_state*, itm_stream) { return ITM_OK; }

// convertDepthAffineToFloat / convertDisparityToDepth  DeviceAgnostic/ITMViewBuilder.h:7-28
int itmo_convert_depth_affine(const int16_t* raw, float* out, int w, int h, float a, float b, itm_stream) {
  for (int i = 0; i < w * h; ++i) { int16_t d = raw[i]; out[i] = ((d <= 0) || (d > 32000)) ? -1.0f : (float)d * a + b; }
  return ITM_OK;
}
int itmo_convert_disparity(const int16_t* raw, float* out, int w, int h, float c0, float c1, float fx, itm_stream) {
  for (int i = 0; i < w * h; ++i) {
    float t = c0 - (float)raw[i];
    float depth;
    if (t == 0) depth = 0.0; else depth = 8.0f * c1 * fx / t;
    out[i] = (depth > 0) ? depth : -1.0f;
  }
  return ITM_OK;
}

// filterDepth / computeNormalAndWeight  DeviceAgnostic/ITMViewBuilder.h:30-117 ; DepthFiltering / ComputeNormalAndWeights /
// UpdateView  DeviceSpecific/CPU/ITMViewBuilder_CPU.cpp:14-63,119-145
int itmo_filter_depth(const float* in, float* out, int w, int h, itm_stream) {
  std::memset(out, 0, (size_t)w * h * 4);
  for (int y = 2; y < h - 2; ++y) for (int x = 2; x < w - 2; ++x) {
    float z = in[x + y * w];
    if (z < 0.0f) { out[x + y * w] = -1.0f; continue; }
    float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / std::sqrt(z) * 0.25f);
    float final_depth = 0.0f, w_sum = 0.0f;
    for (int i = -2; i <= 2; ++i) for (int j = -2; j <= 2; ++j) {
      float tmpz = in[(x + j) + (y + i) * w];
      if (tmpz < 0.0f) continue;
      float dz = (tmpz - z); dz *= dz;
      float wgt = std::exp(-0.5f * ((std::abs(i) + std::abs(j)) * 1.2232f * 1.2232f + dz * sigma_z * sigma_z));
      w_sum += wgt;
      final_depth += wgt * tmpz;
    }
    final_depth /= w_sum;
    out[x + y * w] = final_depth;
  }
  return ITM_OK;
}
int itmo_compute_normal_and_weights(const float* depth, float* normals, float* sigmaZ, int w, int h, const float intr[4], itm_stream) {
  const float PIf = float(3.1415926535897932384626433832795);
  for (int y = 2; y < h - 2; ++y) for (int x = 2; x < w - 2; ++x) {
    const int idx = x + y * w;
    float* n = normals + 4 * (size_t)idx;
    float z = depth[idx];
    if (z < 0.0f) { n[3] = -1.0f; sigmaZ[idx] = -1; continue; }
    float zxp = depth[(x + 1) + y * w], zyp = depth[x + (y + 1) * w], zxm = depth[(x - 1) + y * w], zym = depth[x + (y - 1) * w];
    if (zxp <= 0 || zyp <= 0 || zxm <= 0 || zym <= 0) { n[3] = -1.0f; sigmaZ[idx] = -1; continue; }
    float xp1x = zxp * ((x + 1.0f) - intr[2]) * intr[0], xp1y = zxp * (y - intr[3]) * intr[1];
    float xm1x = zxm * ((x - 1.0f) - intr[2]) * intr[0], xm1y = zxm * (y - intr[3]) * intr[1];
    float yp1x = zyp * (x - intr[2]) * intr[0], yp1y = zyp * ((y + 1.0f) - intr[3]) * intr[1];
    float ym1x = zym * (x - intr[2]) * intr[0], ym1y = zym * ((y - 1.0f) - intr[3]) * intr[1];
    float dxx = xp1x - xm1x, dxy = xp1y - xm1y, dxz = zxp - zxm;
    float dyx = yp1x - ym1x, dyy = yp1y - ym1y, dyz = zyp - zym;
    float nx = (dxy * dyz - dxz * dyy), ny = (dxz * dyx - dxx * dyz), nz = (dxx * dyy - dxy * dyx);
    if (nx == 0.0f && ny == 0 && nz == 0) { n[3] = -1.0f; sigmaZ[idx] = -1; continue; }
    float norm = 1.0f / std::sqrt(nx * nx + ny * ny + nz * nz);
    nx *= norm; ny *= norm; nz *= norm;
    n[0] = nx; n[1] = ny; n[2] = nz; n[3] = 1.0f;
    float theta = std::acos(nz);
    float theta_diff = theta / (PIf * 0.5f - theta);
    sigmaZ[idx] = (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / std::sqrt(z) * theta_diff * theta_diff);
  }
  return ITM_OK;
}
int itmo_update_view(const int16_t* raw, int w, int h, int calibType, float c0, float c1, const float intr_d[4], int useBilateralFilter,
                     int modelSensorNoise, float* depth_out, float* scratch, float* normals, float* sigmaZ, itm_stream st) {
  if (calibType == 0) itmo_convert_disparity(raw, depth_out, w, h, c0, c1, intr_d[0], st);
  else if (calibType == 1) itmo_convert_depth_affine(raw, depth_out, w, h, c0, c1, st);
  else return ITM_ERR_INVALID;
  if (useBilateralFilter) {
    itmo_filter_depth(depth_out, scratch, w, h, st); itmo_filter_depth(scratch, depth_out, w, h, st);
    itmo_filter_depth(depth_out, scratch, w, h, st); itmo_filter_depth(scratch, depth_out, w, h, st);
    itmo_filter_depth(depth_out, scratch, w, h, st);
    std::memcpy(depth_out, scratch, (size_t)w * h * 4);
  }
  if (modelSensorNoise) itmo_compute_normal_and_weights(depth_out, normals, sigmaZ, w, h, intr_d, st);
  return ITM_OK;
}

int itmo_get_counters(const itm_scene* s, const itm_render_state* rs, itm_counters* c, itm_stream) {
  if (!c) return fail(ITM_ERR_INVALID, "null argument");
  std::memset(c, 0, sizeof *c);
  if (s) { c->lastFreeBlockId = s->lastFreeBlockId; c->lastFreeExcessListId = s->lastFreeExcessListId; c->noAllocRequests = s->noAllocRequests; }
  if (rs) { c->noVisibleEntries = rs->noVisibleEntries; c->noFwdProjMissingPoints = rs->noFwdProjMissingPoints; c->noTotalPoints = rs->noTotalPoints; c->noRenderingBlocks = rs->noRenderingBlocks; }
  return ITM_OK;
}
int itmo_set_counters(itm_scene* s, itm_render_state* rs, const itm_counters* c, itm_stream) {
  if (!c) return fail(ITM_ERR_INVALID, "null argument");
  if (s) { s->lastFreeBlockId = c->lastFreeBlockId; s->lastFreeExcessListId = c->lastFreeExcessListId; }
  if (rs) { rs->noVisibleEntries = c->noVisibleEntries; rs->rawVisible = c->noVisibleEntries; }
  return ITM_OK;
}

static void* buf_of(const itm_scene* s, const itm_render_state* rs, int which, size_t* bytes) {
  *bytes = 0;
  switch (which) {
    case ITM_BUF_HASH_ENTRIES: if (s) { *bytes = s->hash.size() * sizeof(HashEntry); return (void*)s->hash.data(); } break;
    case ITM_BUF_EXCESS_LIST: if (s) { *bytes = s->excessList.size() * 4; return (void*)s->excessList.data(); } break;
    case ITM_BUF_VOXEL_BLOCKS: if (s) { *bytes = s->vba.size(); return (void*)s->vba.data(); } break;
    case ITM_BUF_ALLOCATION_LIST: if (s) { *bytes = s->allocList.size() * 4; return (void*)s->allocList.data(); } break;
    case ITM_BUF_VISIBLE_IDS: if (rs) { *bytes = rs->visibleIds.size() * 4; return (void*)rs->visibleIds.data(); } break;
    case ITM_BUF_VISIBLE_TYPE: if (rs) { *bytes = rs->visibleType.size(); return (void*)rs->visibleType.data(); } break;
    case ITM_BUF_RANGE_IMAGE: if (rs) { *bytes = rs->range.size() * sizeof(V2f); return (void*)rs->range.data(); } break;
    case ITM_BUF_RAYCAST_RESULT: if (rs) { *bytes = rs->raycast.size() * sizeof(V4f); return (void*)rs->raycast.data(); } break;
    case ITM_BUF_RAYCAST_IMAGE: if (rs) { *bytes = rs->image.size() * 4; return (void*)rs->image.data(); } break;
    case ITM_BUF_FORWARD_PROJECTION: if (rs) { *bytes = rs->fwdProj.size() * sizeof(V4f); return (void*)rs->fwdProj.data(); } break;
    case ITM_BUF_MISSING_POINTS: if (rs) { *bytes = rs->missing.size() * 4; return (void*)rs->missing.data(); } break;
    case ITM_BUF_SWAP_STATES: if (s && !s->swapStates.empty()) { *bytes = s->swapStates.size(); return (void*)s->swapStates.data(); } break;
  }
  return nullptr;
}
size_t itmo_buffer_bytes(const itm_scene* s, const itm_render_state* rs, int which) { size_t b; buf_of(s, rs, which, &b); return b; }
void* itmo_buffer_ptr(const itm_scene* s, const itm_render_state* rs, int which) { size_t b; return buf_of(s, rs, which, &b); }
int itmo_download(const itm_scene* s, const itm_render_state* rs, int which, void* dst, size_t bytes, itm_stream) {
  size_t b; void* p = buf_of(s, rs, which, &b);
  if (!p && b == 0 && bytes == 0) return ITM_OK;
  if (!p || bytes > b) return fail(ITM_ERR_INVALID, "bad buffer / size");
  std::memcpy(dst, p, bytes);
  return ITM_OK;
}
int itmo_upload(itm_scene* s, itm_render_state* rs, int which, const void* src, size_t bytes, itm_stream) {
  size_t b; void* p = buf_of(s, rs, which, &b);
  if (!p || bytes > b) return fail(ITM_ERR_INVALID, "bad buffer / size");
  std::memcpy(p, src, bytes);
  return ITM_OK;
}

// test-only: read (and optionally clear) the work counters; 15 int64 values
int itmo_debug_ray_trace(int* buf, int width) { g_rayTrace = buf; g_rayTraceW = width; return 0; }
int itmo_debug_stats(long long* out, int clear) {
  if (out) std::memcpy(out, &g_stats, sizeof g_stats);
  if (clear) std::memset(&g_stats, 0, sizeof g_stats);
  return (int)(sizeof g_stats / sizeof(long long));
}


int itmo_mesh_create(const itm_scene* s, uint32_t maxTriangles, itm_mesh** out) {
  if (!s || !out) return fail(ITM_ERR_INVALID, "null argument");
  itm_mesh* m = new itm_mesh();
  m->scene = s; m->noTotalTriangles = 0;
  m->maxTriangles = maxTriangles ? maxTriangles : (uint32_t)s->cfg.localBlockNum * 32u;
  if (m->maxTriangles < 2) { delete m; return fail(ITM_ERR_INVALID, "a mesh needs room for at least two triangles"); }
  m->tri.assign((size_t)m->maxTriangles * 9, 0.0f);
  *out = m;
  return ITM_OK;
}
int itmo_mesh_destroy(itm_mesh* m) { delete m; return ITM_OK; }
int itmo_mesh_scene(const itm_scene* s, itm_mesh* m, itm_stream) {
  if (!s || !m || m->scene != s) return fail(ITM_ERR_INVALID, "bad argument");
  if (s->cfg.indexType != ITM_INDEX_HASH) { std::fill(m->tri.begin(), m->tri.end(), 0.0f); m->noTotalTriangles = 0; return ITM_OK; }
  return dispatch_voxel(s->cfg.voxelType, [&](auto tag) { mesh_scene_t<VOX_T>(s, m); });
}
int itmo_mesh_info(const itm_mesh* m, uint32_t* n, uint32_t* cap, const float** tri, itm_stream) {
  if (!m) return fail(ITM_ERR_INVALID, "null mesh");
  if (n) *n = m->noTotalTriangles;
  if (cap) *cap = m->maxTriangles;
  if (tri) *tri = m->tri.data();
  return ITM_OK;
}
int itmo_mesh_download(const itm_mesh* m, float* dst, uint32_t capacity, uint32_t* n, itm_stream) {
  if (!m || !n) return fail(ITM_ERR_INVALID, "null argument");
  *n = m->noTotalTriangles;
  const uint32_t k = *n < capacity ? *n : capacity;
  if (k) std::memcpy(dst, m->tri.data(), (size_t)k * 36);
  return ITM_OK;
}
int itmo_mesh_write_obj(const itm_mesh* m, const char* path, itm_stream) {
  FILE* f = fopen(path, "w+");
  if (!f) return fail(ITM_ERR_INVALID, "cannot create file");
  const uint32_t n = m->noTotalTriangles;
  for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) fprintf(f, "v %f %f %f\n", m->tri[(size_t)i * 9 + 3 * k], m->tri[(size_t)i * 9 + 3 * k + 1], m->tri[(size_t)i * 9 + 3 * k + 2]);
  for (uint32_t i = 0; i < n; ++i) fprintf(f, "f %d %d %d\n", i * 3 + 2 + 1, i * 3 + 1 + 1, i * 3 + 0 + 1);
  fclose(f);
  return ITM_OK;
}
int itmo_mesh_write_stl(const itm_mesh* m, const char* path, itm_stream) {
  FILE* f = fopen(path, "wb+");
  if (!f) return fail(ITM_ERR_INVALID, "cannot create file");
  for (int i = 0; i < 80; ++i) fwrite(" ", 1, 1, f);
  const uint32_t n = m->noTotalTriangles;
  fwrite(&n, 4, 1, f);
  const float zero[3] = {0, 0, 0}; const short attr = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const float* t = &m->tri[(size_t)i * 9];
    fwrite(zero, 4, 3, f); fwrite(t + 6, 4, 3, f); fwrite(t + 3, 4, 3, f); fwrite(t, 4, 3, f); fwrite(&attr, 2, 1, f);
  }
  fclose(f);
  return ITM_OK;
}

int itmo_export_visible_record(const itm_render_state* rs, const float M_d[16], int max_ids, void* dst, itm_stream) {
  if (!rs || !dst || max_ids < 0) return fail(ITM_ERR_INVALID, "bad argument");
  std::memcpy(dst, M_d, 64);
  int32_t* d = (int32_t*)dst + 16;
  d[0] = rs->noVisibleEntries;
  for (int i = 0; i < max_ids; ++i) d[1 + i] = (i < rs->noVisibleEntries) ? rs->visibleIds[i] : -1;
  return ITM_OK;
}

}  // extern "C"

// ==========================================================================================
// ICP depth tracker: sequential restatement
//   filterSubsampleWithHoles          DeviceAgnostic/ITMLowLevelEngine.h:26-47
//   computePerPointGH_Depth(_Ab)      DeviceAgnostic/ITMDepthTracker.h:8-106
//   ITMDepthTracker_CPU::ComputeGandH DeviceSpecific/CPU/ITMDepthTracker_CPU.cpp:15-79
//   ITMDepthTracker::TrackCamera      Engine/ITMDepthTracker.cpp:79-200
//   ITMPose parameter conversions     Objects/ITMPose.cpp:84-236,305-326 ; ORUtils/Cholesky.h
// ==========================================================================================
namespace {

bool bilerp_holes(const V4f* src, float px, float py, int W, V4f& r) {
  int16_t ix = (int16_t)std::floor(px), iy = (int16_t)std::floor(py);
  float dx = px - (float)ix, dy = py - (float)iy;
  const V4f &a = src[ix + iy * W], &b = src[(ix + 1) + iy * W], &c = src[ix + (iy + 1) * W], &d = src[(ix + 1) + (iy + 1) * W];
  if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) { r = V4f{0, 0, 0, -1.0f}; return false; }
  r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
  r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
  r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
  r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
  return true;
}

void subsample_holes(const float* in, int wIn, int hIn, float* out) {
  int w = wIn / 2, h = hIn / 2;
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    int sx = x * 2, sy = y * 2;
    float acc = 0.0f, good = 0.0f, v;
    v = in[sx + sy * wIn]; if (v > 0.0f) { acc += v; good++; }
    v = in[(sx + 1) + sy * wIn]; if (v > 0.0f) { acc += v; good++; }
    v = in[sx + (sy + 1) * wIn]; if (v > 0.0f) { acc += v; good++; }
    v = in[(sx + 1) + (sy + 1) * wIn]; if (v > 0.0f) { acc += v; good++; }
    if (good > 0) acc /= good;
    out[x + y * w] = acc;
  }
}

long long g_ghEvaluations = 0;   // test-only counter (itmo_debug_gh_evaluations)
int g_and_h(const float* depth, int w, int h, const float* vi, const V4f* pts, const V4f* nrm, int sW, int sH, const float* si,
            const float* invPose, const float* scenePose, float distThresh, int type, itm_tracker_gh* out) {
  ++g_ghEvaluations;
  std::memset(out, 0, sizeof *out);
  if (type == ITM_TRACKER_ITERATION_NONE) return 0;
  const bool shortIt = type != ITM_TRACKER_ITERATION_BOTH;
  const int np = shortIt ? 3 : 6, nh = np * (np + 1) / 2;
  float sumH[21], sumN[6], sumF = 0.0f; int n = 0;
  for (int i = 0; i < 21; ++i) sumH[i] = 0; for (int i = 0; i < 6; ++i) sumN[i] = 0;
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    float d = depth[x + y * w];
    if (d <= 1e-8f) continue;
    V4f p3 = {d * (((float)x - vi[2]) / vi[0]), d * (((float)y - vi[3]) / vi[1]), d, 1.0f};
    V4f q = mul(invPose, p3); q.w = 1.0f;
    V4f rp = mul(scenePose, q);
    if (rp.z <= 0.0f) continue;
    float u = si[0] * rp.x / rp.z + si[2], v = si[1] * rp.y / rp.z + si[3];
    if (!((u >= 0.0f) && (u <= sW - 2) && (v >= 0.0f) && (v <= sH - 2))) continue;
    V4f cp; bilerp_holes(pts, u, v, sW, cp);
    if (cp.w < 0.0f) continue;
    float ex = cp.x - q.x, ey = cp.y - q.y, ez = cp.z - q.z;
    float dist = ex * ex + ey * ey + ez * ez;
    if (dist > distThresh) continue;
    V4f nn; bilerp_holes(nrm, u, v, sW, nn);
    float b = nn.x * ex + nn.y * ey + nn.z * ez;
    float A[6];
    if (type == ITM_TRACKER_ITERATION_TRANSLATION) { A[0] = nn.x; A[1] = nn.y; A[2] = nn.z; }
    else {
      A[0] = +q.z * nn.y - q.y * nn.z;
      A[1] = -q.z * nn.x + q.x * nn.z;
      A[2] = +q.y * nn.x - q.x * nn.y;
      if (!shortIt) { A[3] = nn.x; A[4] = nn.y; A[5] = nn.z; }
    }
    n++; sumF += b * b;
    for (int r = 0, k = 0; r < np; ++r) {
      sumN[r] += b * A[r];
      for (int c = 0; c <= r; ++c, ++k) sumH[k] += A[r] * A[c];
    }
  }
  for (int r = 0, k = 0; r < np; ++r) for (int c = 0; c <= r; ++c, ++k) out->hessian[r + c * 6] = sumH[k];
  for (int r = 0; r < np; ++r) for (int c = r + 1; c < np; ++c) out->hessian[r + c * 6] = out->hessian[c + r * 6];
  for (int r = 0; r < np; ++r) out->nabla[r] = sumN[r];
  out->noValidPoints = n;
  out->f = (n > 100) ? std::sqrt(sumF) / n : 1e5f;
  (void)nh;
  return n;
}

struct OPose {
  float t[3], r[3], M[16];
  static float dot(const float* a, const float* b) { float s = 0; for (int i = 0; i < 3; ++i) s += a[i] * b[i]; return s; }
  static void cross(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
  static void fromParams(const float* w, const float* t, float* R, float* T) {
    float one_6th = 1.0f / 6.0f, one_20th = 1.0f / 20.0f;
    float theta_sq = dot(w, w), theta = std::sqrt(theta_sq), A, B, cv[3];
    cross(w, t, cv);
    if (theta_sq < 1e-8f) { A = 1.0f - one_6th * theta_sq; B = 0.5f; for (int i = 0; i < 3; ++i) T[i] = t[i] + 0.5f * cv[i]; }
    else {
      float C;
      if (theta_sq < 1e-6f) { C = one_6th * (1.0f - one_20th * theta_sq); A = 1.0f - theta_sq * C; B = 0.5f - 0.25f * one_6th * theta_sq; }
      else { float it = 1.0f / theta; A = sinf(theta) * it; B = (1.0f - cosf(theta)) * (it * it); C = (1.0f - A) * (it * it); }
      float c2[3]; cross(w, cv, c2);
      for (int i = 0; i < 3; ++i) T[i] = t[i] + B * cv[i] + C * c2[i];
    }
    float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2], a, b;
    R[0] = 1.0f - B * (wy2 + wz2); R[4] = 1.0f - B * (wx2 + wz2); R[8] = 1.0f - B * (wx2 + wy2);
    a = A * w[2]; b = B * (w[0] * w[1]); R[0 + 3 * 1] = b - a; R[1 + 3 * 0] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]); R[0 + 3 * 2] = b + a; R[2 + 3 * 0] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]); R[1 + 3 * 2] = b - a; R[2 + 3 * 1] = b + a;
  }
  void modelViewFromParams() {
    float R[9], T[3]; fromParams(r, t, R, T);
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) M[rr + 4 * c] = R[rr + 3 * c];
    M[12] = T[0]; M[13] = T[1]; M[14] = T[2]; M[3] = M[7] = M[11] = 0.0f; M[15] = 1.0f;
  }
  void paramsFromModelView() {
    float R[9], T[3] = {M[12], M[13], M[14]}, rot[3];
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) R[rr + 3 * c] = M[rr + 4 * c];
    float cos_angle = (R[0] + R[4] + R[8] - 1.0f) * 0.5f;
    rot[0] = (R[2 + 3 * 1] - R[1 + 3 * 2]) * 0.5f; rot[1] = (R[0 + 3 * 2] - R[2 + 3 * 0]) * 0.5f; rot[2] = (R[1 + 3 * 0] - R[0 + 3 * 1]) * 0.5f;
    float sin_abs = std::sqrt(dot(rot, rot));
    if (cos_angle > M_SQRT1_2) { if (sin_abs) { float s = asinf(sin_abs) / sin_abs; rot[0] *= s; rot[1] *= s; rot[2] *= s; } }
    else if (cos_angle > -M_SQRT1_2) { float s = acosf(cos_angle) / sin_abs; rot[0] *= s; rot[1] *= s; rot[2] *= s; }
    else {
      float angle = (float)M_PI - asinf(sin_abs);
      float d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle, r2[3];
      if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) { r2[0] = d0; r2[1] = (R[1] + R[3]) * 0.5f; r2[2] = (R[6] + R[2]) * 0.5f; }
      else if (fabsf(d1) > fabsf(d2)) { r2[0] = (R[1] + R[3]) * 0.5f; r2[1] = d1; r2[2] = (R[5] + R[7]) * 0.5f; }
      else { r2[0] = (R[6] + R[2]) * 0.5f; r2[1] = (R[5] + R[7]) * 0.5f; r2[2] = d2; }
      if (dot(r2, rot) < 0.0f) { r2[0] *= -1.0f; r2[1] *= -1.0f; r2[2] *= -1.0f; }
      float len = std::sqrt(dot(r2, r2));
      if (len == 0) { r2[0] = r2[1] = r2[2] = 0; } else { r2[0] /= len; r2[1] /= len; r2[2] /= len; }
      rot[0] = angle * r2[0]; rot[1] = angle * r2[1]; rot[2] = angle * r2[2];
    }
    float shtot = 0.5f, theta = std::sqrt(dot(rot, rot));
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    float hw[3] = {rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f}, z[3] = {0, 0, 0}, HR[9], HT[3], rt[3];
    fromParams(hw, z, HR, HT);
    for (int i = 0; i < 3; ++i) rt[i] = HR[i] * T[0] + HR[i + 3] * T[1] + HR[i + 6] * T[2];
    if (theta > 0.001f) { float denom = dot(rot, rot); float prm = dot(T, rot) * (1 - 2 * shtot) / denom; for (int i = 0; i < 3; ++i) rt[i] -= rot[i] * prm; }
    else { float prm = dot(T, rot) / 24; for (int i = 0; i < 3; ++i) rt[i] -= rot[i] * prm; }
    for (int i = 0; i < 3; ++i) rt[i] /= 2 * shtot;
    for (int i = 0; i < 3; ++i) { r[i] = rot[i]; t[i] = rt[i]; }
  }
};

void chol_solve(const float* mat, int n, const float* v, float* result) {
  std::vector<float> ch(mat, mat + n * n), y(n);
  for (int c = 0; c < n; ++c) {
    float inv_diag = 1;
    for (int r = c; r < n; ++r) {
      float val = ch[c + r * n];
      for (int c2 = 0; c2 < c; ++c2) val -= ch[c + c2 * n] * ch[c2 + r * n];
      if (r == c) { ch[c + r * n] = val; inv_diag = 1.0f / val; } else { ch[r + c * n] = val; ch[c + r * n] = val * inv_diag; }
    }
  }
  for (int i = 0; i < n; ++i) { float val = v[i]; for (int j = 0; j < i; ++j) val -= ch[j + i * n] * y[j]; y[i] = val; }
  for (int i = 0; i < n; ++i) y[i] /= ch[i + i * n];
  for (int i = n - 1; i >= 0; --i) { float val = y[i]; for (int j = i + 1; j < n; ++j) val -= ch[i + j * n] * result[j]; result[i] = val; }
}

}  // namespace

extern "C" {

long long itmo_debug_gh_evaluations(int clear) { const long long v = g_ghEvaluations; if (clear) g_ghEvaluations = 0; return v; }
int itmo_filter_subsample_with_holes(const float* in, int w_in, int h_in, float* out, itm_stream) {
  if (!in || !out || w_in < 2 || h_in < 2) return fail(ITM_ERR_INVALID, "bad argument");
  subsample_holes(in, w_in, h_in, out);
  return ITM_OK;
}

int itmo_tracker_compute_g_and_h(const float* depth, int w, int h, const float vi[4], const float* pts, const float* nrm, int sW, int sH,
                                 const float si[4], const float invPose[16], const float scenePose[16], float distThresh, int type,
                                 itm_tracker_gh* out, itm_stream) {
  if (!depth || !pts || !nrm || !out) return fail(ITM_ERR_INVALID, "bad argument");
  g_and_h(depth, w, h, vi, (const V4f*)pts, (const V4f*)nrm, sW, sH, si, invPose, scenePose, distThresh, type, out);
  return ITM_OK;
}

int itmo_track_camera(const itm_tracker_config* cfg, const itm_view* view, const float* pts, const float* nrm, const float scenePose[16],
                      float M_out[16], itm_stream) {
  if (!cfg || !view || !pts || !nrm || !scenePose || !M_out) return fail(ITM_ERR_INVALID, "null argument");
  const int L = cfg->noHierarchyLevels;
  if (L < 1 || L > 8) return fail(ITM_ERR_INVALID, "noHierarchyLevels must be 1..8");
  std::vector<std::vector<float>> pyr(L);
  std::vector<const float*> dl(L); std::vector<int> wl(L), hl(L); std::vector<float> il(4 * L), dt(L); std::vector<int> its(L);
  dl[0] = view->depth; wl[0] = view->w; hl[0] = view->h;
  for (int k = 0; k < 4; ++k) il[k] = view->intr_d[k];
  for (int i = 1; i < L; ++i) {
    wl[i] = wl[i - 1] / 2; hl[i] = hl[i - 1] / 2;
    if (wl[i] < 1 || hl[i] < 1) return fail(ITM_ERR_INVALID, "image too small for the hierarchy");
    pyr[i].resize((size_t)wl[i] * hl[i]);
    subsample_holes(dl[i - 1], wl[i - 1], hl[i - 1], pyr[i].data());
    dl[i] = pyr[i].data();
    for (int k = 0; k < 4; ++k) il[4 * i + k] = il[4 * (i - 1) + k] * 0.5f;
  }
  its[0] = 2; for (int i = 1; i < L; ++i) its[i] = its[i - 1] + 2;
  float stepT = cfg->distThresh / L; dt[L - 1] = cfg->distThresh;
  for (int i = L - 2; i >= 0; --i) dt[i] = dt[i + 1] - stepT;
  OPose pose; std::memcpy(pose.M, view->M_d, 64); pose.paramsFromModelView();
  float Hg[36], Ng[6], A[36], step[6];
  std::memset(Hg, 0, sizeof Hg); std::memset(Ng, 0, sizeof Ng);
  for (int lev = L - 1; lev >= cfg->noICPRunTillLevel; --lev) {
    int type = cfg->trackingRegime[lev];
    if (type == ITM_TRACKER_ITERATION_NONE) continue;
    float inv[16]; invert4(pose.M, inv);
    OPose good = pose; float f_old = 1e20f, lambda = 1.0f;
    for (int it = 0; it < its[lev]; ++it) {
      itm_tracker_gh gh;
      int n = g_and_h(dl[lev], wl[lev], hl[lev], &il[4 * lev], (const V4f*)pts, (const V4f*)nrm, view->w, view->h, &il[0], inv, scenePose, dt[lev], type, &gh);
      if ((n <= 0) || (gh.f > f_old)) { pose = good; invert4(pose.M, inv); lambda *= 10.0f; }
      else {
        good = pose; f_old = gh.f;
        for (int i = 0; i < 36; ++i) Hg[i] = gh.hessian[i] / n;
        for (int i = 0; i < 6; ++i) Ng[i] = gh.nabla[i] / n;
        lambda /= 10.0f;
      }
      for (int i = 0; i < 36; ++i) A[i] = Hg[i];
      for (int i = 0; i < 6; ++i) A[i + i * 6] *= 1.0f + lambda;
      for (int i = 0; i < 6; ++i) step[i] = 0;
      if (type != ITM_TRACKER_ITERATION_BOTH) { float sm[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sm[r + c * 3] = A[r + c * 6]; chol_solve(sm, 3, Ng, step); }
      else chol_solve(A, 6, Ng, step);
      float s[6] = {0, 0, 0, 0, 0, 0};
      if (type == ITM_TRACKER_ITERATION_ROTATION) { s[0] = step[0]; s[1] = step[1]; s[2] = step[2]; }
      else if (type == ITM_TRACKER_ITERATION_TRANSLATION) { s[3] = step[0]; s[4] = step[1]; s[5] = step[2]; }
      else for (int i = 0; i < 6; ++i) s[i] = step[i];
      float Ti[16] = {1.0f, -s[2], s[1], 0.0f,  s[2], 1.0f, -s[0], 0.0f,  -s[1], s[0], 1.0f, 0.0f,  s[3], s[4], s[5], 1.0f};
      float ninv[16]; matmul4(Ti, inv, ninv);
      invert4(ninv, pose.M); pose.paramsFromModelView();
      pose.paramsFromModelView(); pose.modelViewFromParams();
      invert4(pose.M, inv);
      float len = 0.0f; for (int i = 0; i < 6; ++i) len += step[i] * step[i];
      if (std::sqrt(len) / 6 < cfg->terminationThreshold) break;
    }
  }
  std::memcpy(M_out, pose.M, 64);
  return ITM_OK;
}

}  // extern "C"
