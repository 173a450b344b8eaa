// swapping.hip -- host swapping of voxel blocks.
//
// Reference behaviour:
//   ITMSwappingEngine<TVoxel, ITMVoxelBlockHash>  Engine/ITMSwappingEngine.h:19-36, DeviceSpecific/CPU/ITMSwappingEngine_CPU.cpp:21-168
//   CombineVoxelInformation                        DeviceAgnostic/ITMSwappingEngine.h:7-69
//   ITMGlobalCache                                 Objects/ITMGlobalCache.h:17-135
//   hooks of AllocateSceneFromDepth                DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:250-253 (swap states),
//                                                  :271-285 (entries that were swapped out and are visible again get a voxel block)
//
// MI355X design.  The swap state of every entry (one byte) lives in HBM beside the table; the cache -- a block slot and a flag per
// table entry -- in host memory (calloc: only blocks that were ever swapped out are touched).  Both engine methods pick "the first
// SDF_TRANSFER_BLOCK_NUM entries in table order with property P": an ORDERED selection with a cap, done by one workgroup of 1024 lanes
// (contiguous ranges of the table per lane, one scan), since the rest of a call is a host round trip anyway -- as in the reference's
// CUDA twin, which copies the lists with cudaMemcpy.  The voxel work is one workgroup per selected block, one lane per voxel, with
// the reference's operation order (integer weights converted to float, IEEE divisions).  Everything that adds or removes a voxel
// block also maintains the block directory, the slot directory and the sdf mirror, so the ray caster never sees a stale cell:
// a swapped-out entry stays in the table with ptr == -1 and has NO cell (the request kernel then finds it through the table and
// gives it visible type 2, exactly as the reference does).
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "itm_internal.h"
#include "wave_utils.h"

struct SwapHost {
  int cap = 0x1000;
  uint8_t* hasStored = nullptr;        // bool[noTotalEntries]
  uint8_t* stored = nullptr;           // TVoxel[noTotalEntries * 512], calloc
  // transfer buffers (ITMGlobalCache::syncedVoxelBlocks / hasSyncedData / neededEntryIDs): device + pinned host
  void* xferBlocksDev = nullptr; void* xferBlocksHost = nullptr;
  uint8_t* xferFlagsDev = nullptr; uint8_t* xferFlagsHost = nullptr;
  int32_t* xferIdsDev = nullptr; int32_t* xferIdsHost = nullptr;     // [cap + 1]: ids, then the count
};

namespace itm {

constexpr int kSelThreads = 1024;

// mode 0: entries in state 1 (IntegrateGlobalIntoLocal / LoadFromGlobalMemory); mode 1: state 2, a block, not visible (SaveToGlobalMemory)
template <int MODE>
__device__ inline bool swap_selected(int e, const uint8_t* __restrict__ states, const uint4* __restrict__ hash, const uint8_t* __restrict__ visT) {
  if (MODE == 0) return states[e] == 1;
  return states[e] == 2 && (int)hash[e].w >= 0 && visT[e] == 0;
}

// ids[0 .. min(count, cap)) = the first selected entries in table order; ids[cap] = that number.  One workgroup.
template <int MODE>
__global__ void __launch_bounds__(kSelThreads) swap_select_kernel(const uint8_t* __restrict__ states, const uint4* __restrict__ hash, const uint8_t* __restrict__ visT,
                                                                  int nEntries, int cap, int32_t* __restrict__ ids) {
  __shared__ int lds[kSelThreads / 64 + 1];
  const int per = (nEntries + kSelThreads - 1) / kSelThreads;
  const int lo = min((int)threadIdx.x * per, nEntries), hi = min(lo + per, nEntries);
  int n = 0;
  for (int e = lo; e < hi; ++e) n += swap_selected<MODE>(e, states, hash, visT) ? 1 : 0;
  int total;
  int pos = block_exclusive_scan<kSelThreads / 64>(n, lds, &total);
  if (pos < cap && n) {
    for (int e = lo; e < hi && pos < cap; ++e)
      if (swap_selected<MODE>(e, states, hash, visT)) ids[pos++] = e;
  }
  if (threadIdx.x == 0) ids[cap] = total < cap ? total : cap;
}

// After AllocateSceneFromDepth's visible list (_CPU.cpp:250-253, :271-285): every visible entry whose newest data is not on the
// device is marked "to be combined" (state 1); visible entries without a voxel block (ptr == -1: swapped out earlier) get one, in
// table order, from the free list -- and a cell in the directories / the mirror.  One workgroup.
__global__ void __launch_bounds__(kSelThreads) swap_after_allocation_kernel(uint8_t* __restrict__ states, uint4* __restrict__ hash, const uint8_t* __restrict__ visT,
                                                                            int nEntries, const int32_t* __restrict__ allocList, SceneCounters* __restrict__ counters,
                                                                            int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, void* __restrict__ mirror,
                                                                            int mirrorFloat, AccelOrigin org) {
  __shared__ int lds[kSelThreads / 64 + 1];
  const int per = (nEntries + kSelThreads - 1) / kSelThreads;
  const int lo = min((int)threadIdx.x * per, nEntries), hi = min(lo + per, nEntries);
  int n = 0;
  for (int e = lo; e < hi; ++e) {
    if (visT[e] == 0) continue;
    if (states[e] != 2) states[e] = 1;
    n += ((int)hash[e].w == -1) ? 1 : 0;
  }
  int total;
  int rank = block_exclusive_scan<kSelThreads / 64>(n, lds, &total);
  const int lastFree = counters->lastFreeBlockId;
  if (n) {
    for (int e = lo; e < hi; ++e) {
      if (visT[e] == 0) continue;
      uint4 raw = hash[e];
      if ((int)raw.w != -1) continue;
      const int vbaIdx = lastFree - rank; ++rank;
      if (vbaIdx < 0) continue;
      const int ptr = allocList[vbaIdx];
      raw.w = (uint32_t)ptr;
      hash[e] = raw;
      const HashEntry he = unpack_entry(raw);
      directory_insert(dirPtr, dirSlot, org, he.px, he.py, he.pz, ptr, e);
      mirror_init_block(mirror, mirrorFloat != 0, org, he.px, he.py, he.pz);     // the block was reset when it was swapped out
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) counters->lastFreeBlockId = lastFree - total;
}

// ---- the voxel arithmetic of DeviceAgnostic/ITMSwappingEngine.h, on the register images of the codecs -------------------------------
__device__ inline uint32_t to_uchar_ref(float x) {      // TO_UCHAR3 per component: round half away from zero, then clamp
  int v = (int)((x < 0) ? (x - 0.5f) : (x + 0.5f));
  v = (v < 255) ? v : 255;
  return (uint32_t)((0 < v) ? v : 0);
}
template <class VX>
__device__ inline typename VX::Reg combine_voxel(typename VX::Reg src, typename VX::Reg dst, int maxW) {
  {
    int newW = VX::w_depth(dst);
    const int oldW = VX::w_depth(src);
    float newF = VX::kShort ? VX::raw_sdf(dst) / 32767.0f : VX::raw_sdf(dst);
    const float oldF = VX::kShort ? VX::raw_sdf(src) / 32767.0f : VX::raw_sdf(src);
    if (oldW != 0) {
      newF = (float)oldW * oldF + (float)newW * newF;
      newW = oldW + newW;
      newF /= (float)newW;
      newW = (newW < maxW) ? newW : maxW;
      dst = VX::with_depth(dst, newF, newW);
    }
  }
  if constexpr (VX::kColor) {
    int nc[3], oc[3], newW, oldW;
    VX::get_color(dst, nc, newW);
    VX::get_color(src, oc, oldW);
    if (oldW != 0) {
      float c[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float newC = (float)nc[k] / 255.0f, oldC = (float)oc[k] / 255.0f;
        c[k] = oldC * (float)oldW + newC * (float)newW;
      }
      newW = oldW + newW;
#pragma unroll
      for (int k = 0; k < 3; ++k) { c[k] /= (float)newW; nc[k] = (int)to_uchar_ref(c[k] * 255.0f); }
      newW = (newW < maxW) ? newW : maxW;
      dst = VX::with_color(dst, nc, newW & 0xff);
    }
  }
  return dst;
}

// IntegrateGlobalIntoLocal, device half: block i of the transfer buffer into the entry's voxel block; state -> 2
template <class VX>
__global__ void __launch_bounds__(512) swap_combine_kernel(const int32_t* __restrict__ ids, const uint8_t* __restrict__ flags, const void* __restrict__ xfer,
                                                           const uint4* __restrict__ hash, void* __restrict__ vba, uint8_t* __restrict__ states, int maxW,
                                                           void* __restrict__ mirror, AccelOrigin org) {
  const int i = blockIdx.x, id = ids[i], t = threadIdx.x;
  const HashEntry he = unpack_entry(hash[id]);
  if (flags[i] && he.ptr >= 0) {        // (the reference dereferences ptr unchecked; a state-1 entry without a block cannot be combined)
    const size_t vi = (size_t)he.ptr * kBlockVoxels + t;
    const typename VX::Reg r = combine_voxel<VX>(VX::load(xfer, (size_t)i * kBlockVoxels + t), VX::load(vba, vi), maxW);
    VX::store(vba, vi, r);
    using MC = MirrorCodec<VX::kShort>;
    size_t mbase;
    if (mirror && mirror_block_base<false>(org, he.px, he.py, he.pz, mbase)) ((typename MC::T*)mirror)[mbase + mirror_block_lin((uint32_t)t)] = MC::of(VX::raw_sdf(r));
  }
  if (t == 0) states[id] = 2;
}

// SaveToGlobalMemory, device half: candidate i (table order) copies its block to the transfer buffer and becomes state 0; while the
// free list has room (vbaIdx < SDF_BUCKET_NUM - 1, sic: ITMSwappingEngine_CPU.cpp:146) the voxel block goes back to the list, the
// entry's ptr becomes -1, the block is reset and its cells are emptied.  Successes are a prefix of the candidates (the bound is on a
// counter that only the successes advance), so candidate i succeeds iff lastFree + i < bucketNum - 1.
template <class VX>
__global__ void __launch_bounds__(512) swap_out_kernel(const int32_t* __restrict__ ids, void* __restrict__ xfer, uint4* hash, void* __restrict__ vba,
                                                       uint8_t* __restrict__ states, int32_t* __restrict__ allocList, const SceneCounters* __restrict__ counters,
                                                       int bucketNum, int localBlockNum, int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, void* __restrict__ mirror, AccelOrigin org) {
  const int i = blockIdx.x, id = ids[i], t = threadIdx.x;
  // the entry is read by ONE thread and handed to the others through LDS: thread 0 rewrites hash[id] below, and nothing else would
  // order that store after a late wave's load of the same entry
  __shared__ uint4 rawShared;
  if (t == 0) rawShared = hash[id];
  __syncthreads();
  const uint4 raw = rawShared;
  const HashEntry he = unpack_entry(raw);
  const size_t vi = (size_t)he.ptr * kBlockVoxels + t;
  VX::store(xfer, (size_t)i * kBlockVoxels + t, VX::load(vba, vi));
  // An exhausted pool leaves lastFreeBlockId BELOW -1 (the allocation sweep keeps decrementing it for every request it cannot serve,
  // _CPU.cpp:189,206), where the reference's `voxelAllocationList[vbaIdx + 1]` is an access in front of the list.  Here (and in the
  // oracle) such a counter counts as -1 = "list empty": the first freed block becomes allocationList[0] and the counter goes to 0,
  // so the blocks a swap-out frees are handed out again -- which is what swapping exists for.
  const int lastFree = counters->lastFreeBlockId < -1 ? -1 : counters->lastFreeBlockId;
  const int vbaIdx = lastFree + i;
  // (the second bound never binds in a consistent scene -- freed blocks were allocated before -- it keeps an uploaded, inconsistent
  // counter from writing past the list, where the reference would)
  const bool release = vbaIdx < bucketNum - 1 && vbaIdx + 1 < localBlockNum;
  if (release) {
    VX::store(vba, vi, VX::init());
    using MC = MirrorCodec<VX::kShort>;
    size_t mbase;
    if (mirror && mirror_block_base<false>(org, he.px, he.py, he.pz, mbase))
      ((typename MC::T*)mirror)[mbase + mirror_block_lin((uint32_t)t)] = VX::kShort ? (typename MC::T)-32768 : (typename MC::T)0xffffffffu;
  }
  if (t == 0) {
    states[id] = 0;
    if (release) {
      allocList[vbaIdx + 1] = he.ptr;
      hash[id] = make_uint4(raw.x, raw.y, raw.z, (uint32_t)-1);
      directory_insert(dirPtr, dirSlot, org, he.px, he.py, he.pz, -1, -1);
    }
  }
}
__global__ void swap_out_commit_kernel(SceneCounters* __restrict__ counters, const int32_t* __restrict__ ids, int cap, int bucketNum, int localBlockNum) {
  const int L = counters->lastFreeBlockId < -1 ? -1 : counters->lastFreeBlockId, n = ids[cap];      // (as in swap_out_kernel)
  int room = (bucketNum - 1 < localBlockNum - 1 ? bucketNum - 1 : localBlockNum - 1) - L;
  room = room < 0 ? 0 : room;
  counters->lastFreeBlockId = L + (n < room ? n : room);
}

int create_swap_state(itm_scene* s) {
  SwapHost* h = new (std::nothrow) SwapHost();
  if (!h) return set_error(ITM_ERR_DEVICE, "out of host memory");
  s->swapHost = h;
  h->cap = s->cfg.transferBlockNum > 0 ? s->cfg.transferBlockNum : 0x1000;
  const size_t N = (size_t)s->noTotalEntries, blockBytes = (size_t)kBlockVoxels * s->voxBytes;
  h->hasStored = (uint8_t*)calloc(N, 1);
  h->stored = (uint8_t*)calloc(N, blockBytes);
  if (!h->hasStored || !h->stored) return set_error(ITM_ERR_DEVICE, "out of host memory (global cache)");
  hipError_t e = hipMalloc((void**)&s->swapStates, N);
  if (e == hipSuccess) e = hipMemset(s->swapStates, 0, N);
  if (e == hipSuccess) e = hipMalloc(&h->xferBlocksDev, (size_t)h->cap * blockBytes);
  if (e == hipSuccess) e = hipMalloc((void**)&h->xferFlagsDev, (size_t)h->cap);
  if (e == hipSuccess) e = hipMalloc((void**)&h->xferIdsDev, ((size_t)h->cap + 1) * 4);
  if (e == hipSuccess) e = hipHostMalloc(&h->xferBlocksHost, (size_t)h->cap * blockBytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->xferFlagsHost, (size_t)h->cap, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->xferIdsHost, ((size_t)h->cap + 1) * 4, hipHostMallocDefault);
  if (e != hipSuccess) return hip_fail(e, "swapping buffers", __FILE__, __LINE__);
  return ITM_OK;
}
void free_swap_state(itm_scene* s) {
  (void)hipFree(s->swapStates); s->swapStates = nullptr;
  SwapHost* h = s->swapHost;
  if (!h) return;
  free(h->hasStored); free(h->stored);
  (void)hipFree(h->xferBlocksDev); (void)hipFree(h->xferFlagsDev); (void)hipFree(h->xferIdsDev);
  if (h->xferBlocksHost) (void)hipHostFree(h->xferBlocksHost);
  if (h->xferFlagsHost) (void)hipHostFree(h->xferFlagsHost);
  if (h->xferIdsHost) (void)hipHostFree(h->xferIdsHost);
  delete h;
  s->swapHost = nullptr;
}

int launch_swap_after_allocation(itm_scene* s, itm_render_state* rs, hipStream_t st) {
  const int mirrorFloat = (s->cfg.voxelType == ITM_VOXEL_F || s->cfg.voxelType == ITM_VOXEL_F_RGB) ? 1 : 0;
  swap_after_allocation_kernel<<<1, kSelThreads, 0, st>>>(s->swapStates, s->hash, rs->visibleType, s->noTotalEntries, s->allocList, s->counters,
                                                          s->dirPtr, s->dirSlot, s->sdfMirror, mirrorFloat, s->org);
  ITM_LAUNCH_CHECK();
  return ITM_OK;
}

}  // namespace itm

using namespace itm;

extern "C" {

int itm_swap_integrate_global_into_local(itm_scene* s, itm_render_state* rs, itm_stream stream) {
  if (!s || !rs || !s->swapHost) return set_error(ITM_ERR_INVALID, "scene without swapping");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  SwapHost* h = s->swapHost;
  hipStream_t st = as_stream(stream);
  const size_t blockBytes = (size_t)kBlockVoxels * s->voxBytes;
  // LoadFromGlobalMemory: which entries are needed (device), what the cache holds for them (host)
  swap_select_kernel<0><<<1, kSelThreads, 0, st>>>(s->swapStates, s->hash, rs->visibleType, s->noTotalEntries, h->cap, h->xferIdsDev);
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipMemcpyAsync(h->xferIdsHost, h->xferIdsDev, ((size_t)h->cap + 1) * 4, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  const int n = h->xferIdsHost[h->cap];
  if (n <= 0) return ITM_OK;
  bool any = false;
  for (int i = 0; i < n; ++i) {
    const int id = h->xferIdsHost[i];
    h->xferFlagsHost[i] = h->hasStored[id];
    if (h->hasStored[id]) { memcpy((uint8_t*)h->xferBlocksHost + (size_t)i * blockBytes, h->stored + (size_t)id * blockBytes, blockBytes); any = true; }
  }
  ITM_HIP(hipMemcpyAsync(h->xferFlagsDev, h->xferFlagsHost, (size_t)n, hipMemcpyHostToDevice, st));
  if (any) ITM_HIP(hipMemcpyAsync(h->xferBlocksDev, h->xferBlocksHost, (size_t)n * blockBytes, hipMemcpyHostToDevice, st));
  int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    swap_combine_kernel<VX><<<n, 512, 0, st>>>(h->xferIdsDev, h->xferFlagsDev, h->xferBlocksDev, s->hash, s->vba, s->swapStates, s->prm.maxW, s->sdfMirror, s->org);
    return ITM_OK;
  });
  if (rc) return rc;
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipStreamSynchronize(st));          // the pinned buffers are reused by the next call
  return ITM_OK;
}

int itm_swap_save_to_global_memory(itm_scene* s, itm_render_state* rs, itm_stream stream) {
  if (!s || !rs || !s->swapHost) return set_error(ITM_ERR_INVALID, "scene without swapping");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  { const int rc = enter_scene(s, rs); if (rc) return rc; }
  SwapHost* h = s->swapHost;
  hipStream_t st = as_stream(stream);
  const size_t blockBytes = (size_t)kBlockVoxels * s->voxBytes;
  swap_select_kernel<1><<<1, kSelThreads, 0, st>>>(s->swapStates, s->hash, rs->visibleType, s->noTotalEntries, h->cap, h->xferIdsDev);
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipMemcpyAsync(h->xferIdsHost, h->xferIdsDev, ((size_t)h->cap + 1) * 4, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  const int n = h->xferIdsHost[h->cap];
  if (n <= 0) return ITM_OK;
  int rc = dispatch_voxel(s->cfg.voxelType, [&](auto vx) {
    using VX = decltype(vx);
    swap_out_kernel<VX><<<n, 512, 0, st>>>(h->xferIdsDev, h->xferBlocksDev, s->hash, s->vba, s->swapStates, s->allocList, s->counters, s->cfg.bucketNum, s->cfg.localBlockNum,
                                           s->dirPtr, s->dirSlot, s->sdfMirror, s->org);
    return ITM_OK;
  });
  if (rc) return rc;
  swap_out_commit_kernel<<<1, 1, 0, st>>>(s->counters, h->xferIdsDev, h->cap, s->cfg.bucketNum, s->cfg.localBlockNum);
  ITM_LAUNCH_CHECK();
  ITM_HIP(hipMemcpyAsync(h->xferBlocksHost, h->xferBlocksDev, (size_t)n * blockBytes, hipMemcpyDeviceToHost, st));
  ITM_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {               // ITMGlobalCache::SetStoredData
    const int id = h->xferIdsHost[i];
    h->hasStored[id] = 1;
    memcpy(h->stored + (size_t)id * blockBytes, (const uint8_t*)h->xferBlocksHost + (size_t)i * blockBytes, blockBytes);
  }
  return ITM_OK;          // (the occupancy bit of a head whose block left stays set: its chain may hold resident blocks)
}

int itm_global_cache_get(const itm_scene* s, int entry, void* dst_host, int* has) {
  if (!s || !s->swapHost || !has || entry < 0 || entry >= s->noTotalEntries) return set_error(ITM_ERR_INVALID, "bad argument");
  *has = s->swapHost->hasStored[entry];
  const size_t blockBytes = (size_t)kBlockVoxels * s->voxBytes;
  if (*has && dst_host) memcpy(dst_host, s->swapHost->stored + (size_t)entry * blockBytes, blockBytes);
  return ITM_OK;
}
int itm_global_cache_flags(const itm_scene* s, uint8_t* dst_host, size_t bytes) {
  if (!s || !s->swapHost || !dst_host || bytes > (size_t)s->noTotalEntries) return set_error(ITM_ERR_INVALID, "bad argument");
  memcpy(dst_host, s->swapHost->hasStored, bytes);
  return ITM_OK;
}

}  // extern "C"
