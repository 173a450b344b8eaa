// meshing.hip -- marching-cubes mesh of the hash scene.
//
// Reference behaviour:
//   ITMMeshingEngine_CPU<TVoxel, ITMVoxelBlockHash>::MeshScene   DeviceSpecific/CPU/ITMMeshingEngine_CPU.cpp:19-58
//   findPointNeighbors / sdfInterp / buildVertList               DeviceAgnostic/ITMMeshingEngine.h:153-231
//   ITMMesh (triangle buffer, noMaxTriangles, WriteOBJ, WriteSTL) Objects/ITMMesh.h:14-124
//   (ITMPlainVoxelArray: MeshScene is empty in the reference, :70-72 -> no triangles here either)
//
// The reference walks the table slot by slot, the 512 voxels of every allocated block in z, y, x order, and appends the
// triangles of each cell to one array -- so the array's ORDER is defined, and so is what happens when it is full (the last
// slot keeps being overwritten, the count stops at noMaxTriangles - 1).  Both are reproduced exactly:
//
// MI355X design: three launches over a device-resident list of the allocated slots (ordered compaction, as the visible list).
//   1. mesh_cells_kernel<COUNT>: one 512-lane workgroup per allocated block.  The 9x9x9 SDF samples the block's cells touch
//      (its own 8^3 voxels plus one layer from the 7 neighbouring blocks, found through the block directory) are staged in
//      LDS once; every lane then classifies its cell from LDS (sign configuration -> triangle count) and a workgroup scan
//      gives the block's total.
//   2. mesh_scan_kernel: exclusive scan of the per-block totals (one workgroup; meshing is an export step, not a frame step).
//   3. mesh_cells_kernel<WRITE>: the same staging, then every lane interpolates its cell's vertices with the reference's float
//      operations and writes its triangles at base(block) + offset(cell), i.e. in the reference's order.
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include <mutex>

#include "itm_internal.h"
#include "mc_tables.h"
#include "shading_device.h"
#include "wave_utils.h"

struct itm_mesh {
  const itm_scene* scene = nullptr;
  uint32_t maxTriangles = 0;
  float* triangles = nullptr;        // ITMMesh::Triangle[maxTriangles]: 9 floats (p0, p1, p2)
  int32_t* slots = nullptr;          // allocated slots in ascending order
  int32_t* blockTriangles = nullptr; // per listed block: triangle count, then exclusive prefix
  uint8_t* flags = nullptr;          // per slot: allocated?
  int32_t* chunkCount = nullptr;
  itm::RenderCounters* listCounters = nullptr;   // noVisibleEntries = number of listed blocks
  uint32_t* totals = nullptr;        // [0] triangles generated, [1] noTotalTriangles (after the cap)
  int capBlocks = 0;
};

namespace itm {

__device__ __constant__ uint64_t d_triangleCases[256];

// which of the 12 edges a sign configuration crosses: an edge is crossed when its two corners have different signs
__device__ inline uint32_t crossed_edges(uint32_t cube) {
  uint32_t mask = 0;
#pragma unroll
  for (int e = 0; e < 12; ++e)
    if (((cube >> kCubeEdge[e][0]) ^ (cube >> kCubeEdge[e][1])) & 1u) mask |= 1u << e;
  return mask;
}

__global__ void __launch_bounds__(256) mesh_flag_kernel(const uint4* __restrict__ hash, int nEntries, uint8_t* __restrict__ flags, int32_t* __restrict__ chunkCount) {
  __shared__ int lds[4];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int slot0 = chunk * kSweepChunk + tid * 8;
  int n = 0;
  if (slot0 < nEntries) {
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((int)hash[slot0 + k].w >= 0) { w[k >> 2] |= 1u << ((k & 3) * 8); ++n; }
    *(uint2*)(flags + slot0) = make_uint2(w[0], w[1]);
  }
  const int sum = block_reduce_sum<4>(n, lds);
  if (tid == 0) chunkCount[chunk] = sum;
}

// sdfInterp (DeviceAgnostic/ITMMeshingEngine.h:194-201), per component; p1/p2 are integer-valued voxel coordinates
__device__ inline void edge_vertex(const float* p1, const float* p2, float v1, float v2, float* out) {
  if (fabsf(0.0f - v1) < 0.00001f) { out[0] = p1[0]; out[1] = p1[1]; out[2] = p1[2]; return; }
  if (fabsf(0.0f - v2) < 0.00001f) { out[0] = p2[0]; out[1] = p2[1]; out[2] = p2[2]; return; }
  if (fabsf(v1 - v2) < 0.00001f) { out[0] = p1[0]; out[1] = p1[1]; out[2] = p1[2]; return; }
  const float t = (0.0f - v1) / (v2 - v1);
  out[0] = p1[0] + t * (p2[0] - p1[0]);
  out[1] = p1[1] + t * (p2[1] - p1[1]);
  out[2] = p1[2] + t * (p2[2] - p1[2]);
}

// block base (voxel index of its first voxel) of block (bx, by, bz), or -1: directory where it covers, table walk elsewhere
__device__ inline int block_base(const VolumeView& vol, int bx, int by, int bz) {
  const uint32_t ux = (uint32_t)(bx - vol.org.dx), uy = (uint32_t)(by - vol.org.dy), uz = (uint32_t)(bz - vol.org.dz);
  if (vol.dirPtr && dir_covers(ux, uy, uz)) {
    const int ptr = vol.dirPtr[dir_cell(ux, uy, uz)];
    return ptr < 0 ? -1 : ptr * kBlockVoxels;
  }
  if ((int)(int16_t)bx != bx || (int)(int16_t)by != by || (int)(int16_t)bz != bz) return -1;   // beyond the table's short coordinates
  return resolve_block(vol, unpack_entry(vol.hash[hash_index(bx, by, bz, vol.mask)]), bx, by, bz);
}

template <class VX, bool WRITE>
__global__ void __launch_bounds__(512) mesh_cells_kernel(VolumeView vol, const int32_t* __restrict__ slots, const RenderCounters* __restrict__ lc,
                                                         int32_t* __restrict__ blockTriangles, float* __restrict__ triangles, uint32_t maxTriangles,
                                                         const uint32_t* __restrict__ totals, float factor) {
  __shared__ float sdf[9 * 9 * 9];          // SDF_valueToFloat of the samples; NaN marks "no voxel stored there"
  __shared__ int nbBase[8];
  __shared__ int scan[9];
  const int nBlocks = lc->noVisibleEntries;
  const int t = threadIdx.x;
  for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
    const HashEntry he = unpack_entry(vol.hash[slots[b]]);
    __syncthreads();                          // previous block's LDS contents are no longer needed
    if (t < 8) nbBase[t] = (t == 0) ? he.ptr * kBlockVoxels : block_base(vol, he.px + (t & 1), he.py + ((t >> 1) & 1), he.pz + (t >> 2));
    __syncthreads();
    for (int i = t; i < 729; i += 512) {
      const int x = i % 9, y = (i / 9) % 9, z = i / 81;
      const int base = nbBase[(x >> 3) | ((y >> 3) << 1) | ((z >> 3) << 2)];
      float v = __builtin_nanf("");
      if (base >= 0) v = VX::to_float(VX::load_raw_sdf(vol.vba, (size_t)(base + (x & 7) + ((y & 7) << 3) + ((z & 7) << 6))));
      sdf[i] = v;
    }
    __syncthreads();
    const int x = t & 7, y = (t >> 3) & 7, z = t >> 6;       // the reference's loop order: z outer, x inner == ascending t
    float val[8];
    bool ok = true;
    uint32_t cube = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = kCubeCorner[k];
      val[k] = sdf[(x + (c & 1)) + (y + ((c >> 1) & 1)) * 9 + (z + (c >> 2)) * 81];
      ok = ok && !(val[k] != val[k]) && !(val[k] == 1.0f);   // findPointNeighbors: every corner stored and != 1.0f
      if (val[k] < 0.0f) cube |= 1u << k;
    }
    int nTri = 0;
    uint64_t list = ~0ull;
    if (ok && cube != 0u && cube != 255u) {
      list = d_triangleCases[cube];
      for (uint64_t l = list; (l & 0xfull) != 0xfull; l >>= 12) ++nTri;
    }
    int total;
    const int offset = block_exclusive_scan<8>(nTri, scan, &total);
    if (!WRITE) {
      if (t == 0) blockTriangles[b] = total;
      continue;
    }
    if (nTri == 0) continue;
    // vertices on the crossed edges, in voxel units (global voxel coordinates as floats), then scaled by the voxel size
    const float gx = (float)(he.px * kBlockSide + x), gy = (float)(he.py * kBlockSide + y), gz = (float)(he.pz * kBlockSide + z);
    float corner[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = kCubeCorner[k];
      // (blockLocation + offset).toFloat(): integer sum converted, the same value as the float sum for these magnitudes
      corner[k][0] = (float)(he.px * kBlockSide + x + (c & 1)); corner[k][1] = (float)(he.py * kBlockSide + y + ((c >> 1) & 1));
      corner[k][2] = (float)(he.pz * kBlockSide + z + (c >> 2));
    }
    (void)gx; (void)gy; (void)gz;
    const uint32_t edges = crossed_edges(cube);
    float vert[12][3];
#pragma unroll
    for (int e = 0; e < 12; ++e)
      if (edges & (1u << e)) edge_vertex(corner[kCubeEdge[e][0]], corner[kCubeEdge[e][1]], val[kCubeEdge[e][0]], val[kCubeEdge[e][1]], vert[e]);
    const uint64_t generated = totals[0];
    uint64_t g = (uint64_t)(uint32_t)blockTriangles[b] + (uint64_t)offset;
    for (uint64_t l = list; (l & 0xfull) != 0xfull; l >>= 12, ++g) {
      // triangles[noTriangles] = ...; if (noTriangles < noMaxTriangles - 1) noTriangles++   (_CPU.cpp:48-52)
      uint64_t dst = g;
      if (g >= (uint64_t)maxTriangles - 1ull) {
        if (g != generated - 1ull) continue;            // only the last triangle generated survives in the last slot
        dst = (uint64_t)maxTriangles - 1ull;
      }
      float* o = triangles + dst * 9ull;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int e = (int)((l >> (4 * k)) & 0xfull);
        // a dynamically indexed private array would live in scratch: select through a switch-free chain over the 12 edges
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
#pragma unroll
        for (int q = 0; q < 12; ++q) if (q == e) { vx = vert[q][0]; vy = vert[q][1]; vz = vert[q][2]; }
        o[3 * k + 0] = vx * factor; o[3 * k + 1] = vy * factor; o[3 * k + 2] = vz * factor;
      }
    }
  }
}

// exclusive scan of the per-block triangle counts (in place) by one workgroup; totals[0] = sum, totals[1] = count after the cap
__global__ void __launch_bounds__(1024) mesh_scan_kernel(int32_t* __restrict__ blockTriangles, const RenderCounters* __restrict__ lc,
                                                         uint32_t* __restrict__ totals, uint32_t maxTriangles) {
  __shared__ int lds[17];
  __shared__ unsigned long long carry;
  const int n = lc->noVisibleEntries;
  if (threadIdx.x == 0) carry = 0ull;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < n) ? blockTriangles[i] : 0;
    int total;
    const int ex = block_exclusive_scan<16>(v, lds, &total);
    const unsigned long long c = carry;
    if (i < n) blockTriangles[i] = (int32_t)(uint32_t)(c + (unsigned long long)ex);   // < 2^32: at most 5 * 512 per block, 2^18 blocks
    __syncthreads();
    if (threadIdx.x == 0) carry = c + (unsigned long long)total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned long long g = carry;
    totals[0] = (uint32_t)g;
    totals[1] = (g < (unsigned long long)maxTriangles - 1ull) ? (uint32_t)g : maxTriangles - 1u;
  }
}

int launch_ordered_compaction(const uint8_t* flags, const int32_t* chunkCount, int nChunks, int nEntries, int32_t* ids, int cap, RenderCounters* rc, hipStream_t st);

static void free_mesh(itm_mesh* m) {
  if (!m) return;
  (void)hipFree(m->triangles); (void)hipFree(m->slots); (void)hipFree(m->blockTriangles); (void)hipFree(m->flags);
  (void)hipFree(m->chunkCount); (void)hipFree(m->listCounters); (void)hipFree(m->totals);
  delete m;
}

static int upload_tables() {
  // once per device, also when several host threads create meshes at the same time
  static std::mutex guard;
  static bool done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(guard);
  if (!done[dev]) {
    ITM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_triangleCases), kTriangleCases, sizeof(kTriangleCases)));
    done[dev] = true;
  }
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_mesh_create(const itm_scene* s, uint32_t maxTriangles, itm_mesh** out) {
  if (!s || !out) return set_error(ITM_ERR_INVALID, "null argument");
  itm_mesh* m = new (std::nothrow) itm_mesh();
  if (!m) return set_error(ITM_ERR_DEVICE, "out of host memory");
  m->scene = s;
  const bool hash = s->cfg.indexType == ITM_INDEX_HASH;
  m->maxTriangles = maxTriangles ? maxTriangles : (uint32_t)s->cfg.localBlockNum * 32u;   // ITMMesh::noMaxTriangles (Objects/ITMMesh.h:23)
  if (m->maxTriangles < 2) { delete m; return set_error(ITM_ERR_INVALID, "a mesh needs room for at least two triangles"); }
  m->capBlocks = hash ? s->cfg.localBlockNum : 0;
  hipError_t e = hipMalloc((void**)&m->triangles, (size_t)m->maxTriangles * 36);
  if (e == hipSuccess) e = hipMalloc((void**)&m->totals, 8);
  if (e == hipSuccess) e = hipMemset(m->totals, 0, 8);
  if (e == hipSuccess && hash) e = hipMalloc((void**)&m->slots, (size_t)m->capBlocks * 4);
  if (e == hipSuccess && hash) e = hipMalloc((void**)&m->blockTriangles, (size_t)m->capBlocks * 4);
  if (e == hipSuccess && hash) e = hipMalloc((void**)&m->flags, (size_t)s->numChunks * kSweepChunk);
  if (e == hipSuccess && hash) e = hipMalloc((void**)&m->chunkCount, (size_t)s->numChunks * 4);
  if (e == hipSuccess && hash) e = hipMalloc((void**)&m->listCounters, sizeof(RenderCounters));
  if (e == hipSuccess) e = hipMemset(m->triangles, 0, (size_t)m->maxTriangles * 36);   // MemoryBlock storage starts zeroed
  if (e != hipSuccess) { free_mesh(m); return hip_fail(e, "mesh buffers", __FILE__, __LINE__); }
  *out = m;
  return ITM_OK;
}

int itm_mesh_destroy(itm_mesh* m) { free_mesh(m); return ITM_OK; }

int itm_mesh_scene(const itm_scene* s, itm_mesh* m, itm_stream stream) {
  if (!s || !m) return set_error(ITM_ERR_INVALID, "null argument");
  if (m->scene != s) return set_error(ITM_ERR_INVALID, "mesh belongs to another scene");
  { const int rc = enter_scene(s, nullptr); if (rc) return rc; }
  hipStream_t st = as_stream(stream);
  // mesh->triangles->Clear()
  ITM_HIP(hipMemsetAsync(m->triangles, 0, (size_t)m->maxTriangles * 36, st));
  ITM_HIP(hipMemsetAsync(m->totals, 0, 8, st));
  if (s->cfg.indexType != ITM_INDEX_HASH) return ITM_OK;       // ITMPlainVoxelArray: empty in the reference
  int rc = upload_tables();
  if (rc) return rc;
  mesh_flag_kernel<<<s->numChunks, 256, 0, st>>>(s->hash, s->noTotalEntries, m->flags, m->chunkCount);
  if ((rc = launch_ordered_compaction(m->flags, m->chunkCount, s->numChunks, s->noTotalEntries, m->slots, m->capBlocks, m->listCounters, st))) return rc;
  const VolumeView vol = make_volume(s);
  const int grid = 256 * 4;
  rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    mesh_cells_kernel<VX, false><<<grid, 512, 0, st>>>(vol, m->slots, m->listCounters, m->blockTriangles, m->triangles, m->maxTriangles, m->totals, s->prm.voxelSize);
    mesh_scan_kernel<<<1, 1024, 0, st>>>(m->blockTriangles, m->listCounters, m->totals, m->maxTriangles);
    mesh_cells_kernel<VX, true><<<grid, 512, 0, st>>>(vol, m->slots, m->listCounters, m->blockTriangles, m->triangles, m->maxTriangles, m->totals, s->prm.voxelSize);
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

int itm_mesh_info(const itm_mesh* m, uint32_t* noTotalTriangles, uint32_t* noMaxTriangles, const float** triangles_dev, itm_stream stream) {
  if (!m) return set_error(ITM_ERR_INVALID, "null mesh");
  uint32_t t[2] = {0, 0};
  hipStream_t st = as_stream(stream);
  ITM_HIP(hipMemcpyAsync(t, m->totals, 8, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  if (noTotalTriangles) *noTotalTriangles = t[1];
  if (noMaxTriangles) *noMaxTriangles = m->maxTriangles;
  if (triangles_dev) *triangles_dev = m->triangles;
  return ITM_OK;
}

int itm_mesh_download(const itm_mesh* m, float* dst_host, uint32_t capacityTriangles, uint32_t* noTotalTriangles, itm_stream stream) {
  if (!m || !noTotalTriangles) return set_error(ITM_ERR_INVALID, "null argument");
  int rc = itm_mesh_info(m, noTotalTriangles, nullptr, nullptr, stream);
  if (rc) return rc;
  const uint32_t n = *noTotalTriangles < capacityTriangles ? *noTotalTriangles : capacityTriangles;
  if (n && !dst_host) return set_error(ITM_ERR_INVALID, "null destination");
  hipStream_t st = as_stream(stream);
  if (n) ITM_HIP(hipMemcpyAsync(dst_host, m->triangles, (size_t)n * 36, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  return ITM_OK;
}

// ITMMesh::WriteOBJ (Objects/ITMMesh.h:34-62): three "v" lines per triangle, then the faces with the winding reversed
int itm_mesh_write_obj(const itm_mesh* m, const char* path, itm_stream stream) {
  if (!m || !path) return set_error(ITM_ERR_INVALID, "null argument");
  uint32_t n = 0;
  int rc = itm_mesh_info(m, &n, nullptr, nullptr, stream);
  if (rc) return rc;
  std::vector<float> tri((size_t)n * 9);
  if ((rc = itm_mesh_download(m, tri.data(), n, &n, stream))) return rc;
  FILE* f = fopen(path, "w+");
  if (!f) return set_error(ITM_ERR_INVALID, std::string("cannot create ") + path);
  for (uint32_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) fprintf(f, "v %f %f %f\n", tri[(size_t)i * 9 + 3 * k], tri[(size_t)i * 9 + 3 * k + 1], tri[(size_t)i * 9 + 3 * k + 2]);
  for (uint32_t i = 0; i < n; ++i) fprintf(f, "f %d %d %d\n", i * 3 + 2 + 1, i * 3 + 1 + 1, i * 3 + 0 + 1);
  const bool ok = fclose(f) == 0;
  return ok ? ITM_OK : set_error(ITM_ERR_INVALID, std::string("short write to ") + path);
}

// ITMMesh::WriteSTL (:64-110): binary STL, 80 spaces of header, zero normals, vertices in the order p2, p1, p0
int itm_mesh_write_stl(const itm_mesh* m, const char* path, itm_stream stream) {
  if (!m || !path) return set_error(ITM_ERR_INVALID, "null argument");
  uint32_t n = 0;
  int rc = itm_mesh_info(m, &n, nullptr, nullptr, stream);
  if (rc) return rc;
  std::vector<float> tri((size_t)n * 9);
  if ((rc = itm_mesh_download(m, tri.data(), n, &n, stream))) return rc;
  FILE* f = fopen(path, "wb+");
  if (!f) return set_error(ITM_ERR_INVALID, std::string("cannot create ") + path);
  for (int i = 0; i < 80; ++i) fwrite(" ", 1, 1, f);
  fwrite(&n, 4, 1, f);
  const float zero[3] = {0.0f, 0.0f, 0.0f};
  const short attribute = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const float* t = &tri[(size_t)i * 9];
    fwrite(zero, 4, 3, f);
    fwrite(t + 6, 4, 3, f); fwrite(t + 3, 4, 3, f); fwrite(t, 4, 3, f);
    fwrite(&attribute, 2, 1, f);
  }
  const bool ok = fclose(f) == 0;
  return ok ? ITM_OK : set_error(ITM_ERR_INVALID, std::string("short write to ") + path);
}

}  // extern "C"
