// ITMEngines_HIP.h -- the binding a maintainer of the reference adds: two classes derived from the reference's OWN
// abstract engines (ITMLib/Engine/ITMSceneReconstructionEngine.h:28-52, ITMLib/Engine/ITMVisualisationEngine.h:18-107)
// that forward every virtual to the C-ABI of include/itm_hip.h.  Compiled only where the reference tree is on the
// include path (oracle/Makefile target `hipdemo`, tests/test_reference_integration.py); nothing of the reference is
// copied here -- the classes below are new code against its public interfaces.
//
// Memory model.  The scene lives in HBM inside libitmhip.so ("device twin" of an ITMScene object, created on first
// use).  This adapter is written for a reference build WITHOUT CUDA, whose images only have host storage:
//   * a view's depth / colour image is staged into HBM ONCE per frame: HipMarkViewUpdated(view) -- called by
//     ITMViewBuilder_HIP::UpdateView (ITMTrackers_HIP.h), or by whoever else fills the view -- starts a new generation of the
//     view, and the engine calls of a frame (AllocateSceneFromDepth, IntegrateIntoScene, CreateICPMaps: ITMDenseMapper.cpp:50-57,
//     ITMTrackingController.cpp:30-46) then hand the library byte-identical views, which is what lets it record the first
//     three and launch the fused frame at the fourth (include/itm_hip.h, "the four calls of a frame").  A view nobody
//     ever marked is staged on every call (always correct, never fused);
//   * what the rest of InfiniTAM reads on the host (visible list, range image, ray-cast result, ICP maps, shaded image,
//     counters) comes back under a POLICY: HIP_MIRROR_EAGER downloads after every call (a host that reads its images
//     right after each engine call: today's ITMLib with a host tracker), HIP_MIRROR_ON_DEMAND leaves everything in HBM
//     until HipSyncRenderStateToHost / HipSyncTrackingStateToHost / HipSyncViewToHost ask for it (a host whose next
//     consumer is on the device too -- ITMDepthTracker_HIP takes the ICP maps straight from HBM -- or that only looks at
//     results now and then: a UI, a mesh export).  Results are identical; ON_DEMAND is the one in which a frame is 5 launches
//     and no synchronisation (measured by ref_hip_demo --bench).
// A build whose MemoryBlocks have a device slot passes those pointers instead and drops the staging.
// SyncSceneToHost() copies table + voxels into the reference's host scene on demand (saving, meshing, tests).
#pragma once

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "ITMLib/Engine/ITMSceneReconstructionEngine.h"
#include "ITMLib/Engine/ITMVisualisationEngine.h"
#include "ITMLib/Engine/ITMMeshingEngine.h"
#include "itm_hip.h"

namespace ITMLib {
namespace Engine {

inline void HipCheck(int rc, const char* what) {
  if (rc != ITM_OK) throw std::runtime_error(std::string(what) + ": " + itm_last_error());
}
// itm_memcpy_d2h only ENQUEUES the copy (include/itm_hip.h, conventions); the reference's host code reads its images as soon
// as an engine method returns, so every download of an adapter ends with a synchronisation of the stream it was issued on.
inline int HipDownload(void* dst_host, const void* src_dev, size_t bytes, itm_stream stream) {
  const int rc = itm_memcpy_d2h(dst_host, src_dev, bytes, stream);
  return rc != ITM_OK ? rc : itm_stream_synchronize(stream);
}

enum HipMirrorPolicy { HIP_MIRROR_EAGER = 0, HIP_MIRROR_ON_DEMAND = 1 };

template <class TVoxel> struct HipVoxelTag;
template <> struct HipVoxelTag<ITMVoxel_s> { enum { value = ITM_VOXEL_S }; };
template <> struct HipVoxelTag<ITMVoxel_f> { enum { value = ITM_VOXEL_F }; };
template <> struct HipVoxelTag<ITMVoxel_s_rgb> { enum { value = ITM_VOXEL_S_RGB }; };
template <> struct HipVoxelTag<ITMVoxel_f_rgb> { enum { value = ITM_VOXEL_F_RGB }; };

// render-state shells that own their device twin (the reference deletes render states through the virtual destructor)
struct HipRenderStateTwin {
  itm_render_state* dev = nullptr;
  const itm_scene* devScene = nullptr;
  // HIP_MIRROR_ON_DEMAND: parts of the host shell that are older than the device twin (HipSyncRenderStateToHost brings them back)
  bool staleList = false, staleRange = false, staleRays = false, staleImage = false, staleForward = false;
  virtual ~HipRenderStateTwin() { itm_render_state_destroy(dev); }
};
struct ITMRenderState_VH_HIP : public ITMRenderState_VH, public HipRenderStateTwin {
  ITMRenderState_VH_HIP(int noTotalEntries, const Vector2i& s, float vfMin, float vfMax) : ITMRenderState_VH(noTotalEntries, s, vfMin, vfMax, MEMORYDEVICE_CPU) {}
};
struct ITMRenderState_HIP : public ITMRenderState, public HipRenderStateTwin {
  ITMRenderState_HIP(const Vector2i& s, float vfMin, float vfMax) : ITMRenderState(s, vfMin, vfMax, MEMORYDEVICE_CPU) {}
};

template <class TIndex> struct HipIndexTraits;
template <> struct HipIndexTraits<ITMVoxelBlockHash> {
  enum { value = ITM_INDEX_HASH };
  static void Configure(const ITMVoxelBlockHash&, itm_scene_config& c) {
    c.bucketNum = SDF_BUCKET_NUM; c.excessNum = SDF_EXCESS_LIST_SIZE; c.localBlockNum = SDF_LOCAL_BLOCK_NUM;
  }
  static ITMRenderState_VH_HIP* NewRenderState(const Vector2i& s, float vfMin, float vfMax) {
    return new ITMRenderState_VH_HIP(ITMVoxelBlockHash::noTotalEntries, s, vfMin, vfMax);
  }
};
template <> struct HipIndexTraits<ITMPlainVoxelArray> {
  enum { value = ITM_INDEX_DENSE };
  static void Configure(const ITMPlainVoxelArray& index, itm_scene_config& c) {
    const ITMPlainVoxelArray::IndexData* d = index.getIndexData();
    c.denseSize[0] = d->size.x; c.denseSize[1] = d->size.y; c.denseSize[2] = d->size.z;
    c.denseOffset[0] = d->offset.x; c.denseOffset[1] = d->offset.y; c.denseOffset[2] = d->offset.z;
    c.denseOffsetSet = 1;
  }
  static ITMRenderState_HIP* NewRenderState(const Vector2i& s, float vfMin, float vfMax) { return new ITMRenderState_HIP(s, vfMin, vfMax); }
};

// ---- device twins, keyed by the address of the reference object they shadow ------------------------------------
struct HipRegistry {
  HipMirrorPolicy policy = HIP_MIRROR_EAGER;
  std::map<const void*, itm_scene*> scenes;
  // a view's images in HBM.  `generation` counts the times the host said the view has new content (HipMarkViewUpdated);
  // `depthStaged` / `rgbStaged` say which generation the device copy holds.  hostDepthStale: the device copy is NEWER than the
  // host image (ITMViewBuilder_HIP converted the raw frame in HBM and the policy did not ask for the host copy).
  struct Stage {
    void* depth = nullptr; void* rgb = nullptr; size_t pixels = 0, rgbPixels = 0;
    bool tracked = false; unsigned long long generation = 0, depthStaged = ~0ull, rgbStaged = ~0ull;
    bool hostDepthStale = false;
    void* pinnedDepth = nullptr;                // host image page-locked for the ring's uploads
    // A tracked view's float depth travels through a ring of device slots on a copy stream of its own (the library's itm_depth_stager,
    // sized in bytes): the upload leaves the frame's stream, the HOST image may be rewritten as soon as the staging call returns (the
    // reference reuses ONE ITMView for every frame), and the frames still queued on the device keep reading their own slot.
    itm_depth_stager* ring = nullptr; bool holding = false;
    const void* cur = nullptr;                  // where the current generation's depth lies in HBM: a slot of the ring, or `depth`
  };
  std::map<const void*, Stage> views;
  // the ICP maps / point cloud of a tracking state in HBM (trackingState->pointCloud->locations / ->colours)
  // count: elements the last call wrote; icpMaps: what lies in HBM are the maps of the last CreateICPMaps and nothing has replaced them
  // since (CreatePointCloud writes a point list into the same buffers; HipMarkTrackingStateHostWritten: the host rewrote its image)
  struct Maps { void* points = nullptr; void* normals = nullptr; size_t pixels = 0, count = 0; bool hostStale = false, icpMaps = false; const void* locationsImage = nullptr; };
  std::map<const void*, Maps> maps;
  static HipRegistry& Get() { static HipRegistry r; return r; }
  static void Free(Stage& st) {
    if (st.ring) { if (st.holding) itm_depth_stager_release(st.ring, 0); itm_stream_synchronize(0); itm_depth_stager_destroy(st.ring); }
    itm_dev_free(st.depth); itm_dev_free(st.rgb);
    if (st.pinnedDepth) itm_host_unregister(st.pinnedDepth);
    st = Stage();
  }
  static void Free(Maps& m) { itm_dev_free(m.points); itm_dev_free(m.normals); m = Maps(); }
  ~HipRegistry() {
    for (auto& kv : scenes) itm_scene_destroy(kv.second);
    for (auto& kv : views) Free(kv.second);
    for (auto& kv : maps) Free(kv.second);
  }
};
inline void HipSetMirrorPolicy(HipMirrorPolicy p) { HipRegistry::Get().policy = p; }
inline bool HipEager() { return HipRegistry::Get().policy == HIP_MIRROR_EAGER; }

// "This view has new content": to be called once per frame by whoever fills the view's host images (ITMViewBuilder_HIP does;
// a host that keeps the reference's CPU view builder adds this one line behind its UpdateView call).  From then on the view
// is staged once per generation instead of on every engine call.
inline void HipMarkViewUpdated(const ITMView* view) {
  HipRegistry::Stage& st = HipRegistry::Get().views[view];
  st.tracked = true; ++st.generation; st.hostDepthStale = false;
}

template <class TVoxel, class TIndex>
itm_scene* HipSceneOf(const ITMScene<TVoxel, TIndex>* scene) {
  HipRegistry& r = HipRegistry::Get();
  auto it = r.scenes.find(scene);
  if (it != r.scenes.end()) return it->second;
  itm_scene_config c; std::memset(&c, 0, sizeof c);
  c.voxelType = HipVoxelTag<TVoxel>::value; c.indexType = HipIndexTraits<TIndex>::value;
  HipIndexTraits<TIndex>::Configure(scene->index, c);
  c.maxRenderingBlocks = MAX_RENDERING_BLOCKS;
  const ITMSceneParams* sp = scene->sceneParams;
  itm_scene_params p; std::memset(&p, 0, sizeof p);
  p.mu = sp->mu; p.maxW = sp->maxW; p.voxelSize = sp->voxelSize; p.viewFrustum_min = sp->viewFrustum_min; p.viewFrustum_max = sp->viewFrustum_max;
  p.stopIntegratingAtMaxW = sp->stopIntegratingAtMaxW ? 1 : 0;
  itm_scene* dev = nullptr;
  HipCheck(itm_scene_create(&c, &p, &dev), "itm_scene_create");
  // the reference's callers (ITMDenseMapper::ProcessFrame, ITMTrackingController::Prepare) issue the four per-frame calls back to
  // back and read results through the engines: the library may record the first three and fuse the frame (include/itm_hip.h)
  HipCheck(itm_scene_set_deferred_fusion(dev, 1), "itm_scene_set_deferred_fusion");
  r.scenes[scene] = dev;
  return dev;
}

// the scene twin is released by the visualisation engine bound to that scene (render states must be deleted first,
// as the reference does: ITMMainEngine::~ITMMainEngine deletes renderState_live before the engines)
inline void HipReleaseScene(const void* scene) {
  auto it = HipRegistry::Get().scenes.find(scene);
  if (it != HipRegistry::Get().scenes.end()) { itm_scene_destroy(it->second); HipRegistry::Get().scenes.erase(it); }
}
inline void HipReleaseView(const void* view) {
  auto it = HipRegistry::Get().views.find(view);
  if (it == HipRegistry::Get().views.end()) return;
  HipRegistry::Free(it->second);
  HipRegistry::Get().views.erase(it);
}
inline void HipReleaseTrackingState(const void* ts) {
  auto it = HipRegistry::Get().maps.find(ts);
  if (it == HipRegistry::Get().maps.end()) return;
  HipRegistry::Free(it->second);
  HipRegistry::Get().maps.erase(it);
}

inline HipRenderStateTwin* HipTwinOf(const ITMRenderState* rs) {
  HipRenderStateTwin* t = const_cast<HipRenderStateTwin*>(dynamic_cast<const HipRenderStateTwin*>(rs));
  if (!t || !t->dev) throw std::runtime_error("render state was not created by the HIP visualisation engine");
  return t;
}
inline itm_render_state* HipRenderStateOf(const ITMRenderState* rs) { return HipTwinOf(rs)->dev; }

// the device images of a view, allocated on first use
inline HipRegistry::Stage& HipStageOf(const ITMView* view) {
  HipRegistry::Stage& st = HipRegistry::Get().views[view];
  const Vector2i ds = view->depth->noDims, cs = view->rgb->noDims;
  const size_t px = (size_t)ds.x * ds.y, cpx = (size_t)cs.x * cs.y;
  if (st.pixels != px) {
    itm_dev_free(st.depth); HipCheck(itm_dev_malloc(&st.depth, px * 4), "dev_malloc"); st.pixels = px; st.depthStaged = ~0ull; st.cur = st.depth;
    if (st.ring) { if (st.holding) itm_depth_stager_release(st.ring, 0); itm_stream_synchronize(0); itm_depth_stager_destroy(st.ring); st.ring = nullptr; st.holding = false; }
  }
  if (st.rgbPixels != cpx) { itm_dev_free(st.rgb); HipCheck(itm_dev_malloc(&st.rgb, cpx * 4), "dev_malloc"); st.rgbPixels = cpx; st.rgbStaged = ~0ull; }
  return st;
}
// the device maps of a tracking state, allocated on first use
inline HipRegistry::Maps& HipMapsOf(const ITMTrackingState* ts) {
  HipRegistry::Maps& m = HipRegistry::Get().maps[ts];
  const Vector2i s = ts->pointCloud->locations->noDims;
  const size_t px = (size_t)s.x * s.y;
  if (m.pixels != px) {
    itm_dev_free(m.points); itm_dev_free(m.normals);
    HipCheck(itm_dev_malloc(&m.points, px * 16), "dev_malloc"); HipCheck(itm_dev_malloc(&m.normals, px * 16), "dev_malloc");
    m.pixels = px; m.hostStale = false; m.icpMaps = false; m.count = 0;
  }
  m.locationsImage = ts->pointCloud->locations;
  return m;
}

// page-locks a host image the first time it is uploaded from (the reference allocates its images with `new`; a pageable source
// makes every upload a blocking, bounce-buffered copy).  Failure to lock is not an error: the copy is merely slower.
inline void HipPin(void*& pinned, const void* host, size_t bytes) {
  if (pinned == host) return;
  if (pinned) itm_host_unregister(pinned);
  // only a registration made HERE is recorded (and later undone): ITM_ALREADY_REGISTERED means the range is someone else's
  pinned = (itm_host_register(const_cast<void*>(host), bytes) == ITM_OK) ? const_cast<void*>(host) : nullptr;
}

// stages the host images of a view in HBM -- once per generation for a tracked view -- and fills the POD view of the C-ABI
inline itm_view HipStageView(const ITMView* view, const ITMPose* pose_d, bool withRgb) {
  HipRegistry::Stage& st = HipStageOf(view);
  const Vector2i ds = view->depth->noDims, cs = view->rgb->noDims;
  if (!st.tracked) {
    // nobody ever said when this view changes: staged on every call, from pageable memory (the runtime copies the host image before the
    // call returns: whatever the host does to it afterwards is harmless)
    HipCheck(itm_memcpy_h2d(st.depth, view->depth->GetData(MEMORYDEVICE_CPU), st.pixels * 4, 0), "memcpy_h2d");
    st.cur = st.depth;
  } else if (st.depthStaged != st.generation) {
    // a new generation: through the ring.  The slot of the previous generation is released behind everything the frame's stream has
    // been given so far (the calls of the previous frame, launched by its CreateICPMaps); the upload runs on the ring's copy stream; the
    // host waits until the copy has READ the host image -- ~25-50 us, the one wait of a frame, and it overlaps the device's work on the
    // frames before -- so that the caller may rewrite the image; the frame's stream needs no dependency on the copy (it has finished).
    void* host = view->depth->GetData(MEMORYDEVICE_CPU);
    if (!st.ring) HipCheck(itm_depth_stager_create(2 * ds.x, ds.y, 4, &st.ring), "depth ring");      // (slots of 2w x h shorts = w x h floats)
    if (st.holding) { HipCheck(itm_depth_stager_release(st.ring, 0), "depth ring release"); st.holding = false; }
    HipPin(st.pinnedDepth, host, st.pixels * 4);
    HipCheck(itm_depth_stager_upload(st.ring, (const int16_t*)host), "depth ring upload");
    for (int busy = 1; busy;) HipCheck(itm_depth_stager_pending(st.ring, nullptr, &busy), "depth ring pending");
    const int16_t* slot = nullptr;
    HipCheck(itm_depth_stager_acquire(st.ring, 0, &slot), "depth ring acquire");
    st.holding = true; st.cur = slot;
    st.depthStaged = st.generation;
  }
  if (withRgb && (!st.tracked || st.rgbStaged != st.generation)) {
    // (colour images stay on the frame's stream and in pageable memory: staged by the runtime before the call returns)
    HipCheck(itm_memcpy_h2d(st.rgb, view->rgb->GetData(MEMORYDEVICE_CPU), st.rgbPixels * 4, 0), "memcpy_h2d");
    st.rgbStaged = st.generation;
  }
  itm_view v; std::memset(&v, 0, sizeof v);
  // (the rgb pointer is handed over whether or not this call uploaded it: the calls of one frame must name the same view)
  v.depth = (const float*)st.cur; v.rgb = (const uint8_t*)st.rgb;
  v.w = ds.x; v.h = ds.y; v.w_rgb = cs.x; v.h_rgb = cs.y;
  std::memcpy(v.M_d, pose_d->GetM().m, 64);
  std::memcpy(v.intr_d, &view->calib->intrinsics_d.projectionParamsSimple.all, 16);
  std::memcpy(v.intr_rgb, &view->calib->intrinsics_rgb.projectionParamsSimple.all, 16);
  std::memcpy(v.rgb_to_depth, view->calib->trafo_rgb_to_depth.calib.m, 64);
  std::memcpy(v.rgb_to_depth_inv, view->calib->trafo_rgb_to_depth.calib_inv.m, 64);
  return v;
}

// ---- bringing results back to the host shells (every call under HIP_MIRROR_EAGER, on request under HIP_MIRROR_ON_DEMAND) ----
template <class TIndex> struct HipMirror;   // the index-specific part of a render state
template <> struct HipMirror<ITMVoxelBlockHash> {
  static void VisibleList(const itm_scene* dev, ITMRenderState* rs) {
    ITMRenderState_VH* vh = (ITMRenderState_VH*)rs;
    itm_render_state* d = HipRenderStateOf(rs);
    itm_counters c; HipCheck(itm_get_counters(dev, d, &c, 0), "get_counters");
    vh->noVisibleEntries = c.noVisibleEntries;
    HipCheck(itm_download(dev, d, ITM_BUF_VISIBLE_IDS, vh->GetVisibleEntryIDs(), itm_buffer_bytes(dev, d, ITM_BUF_VISIBLE_IDS), 0), "download ids");
    HipCheck(itm_download(dev, d, ITM_BUF_VISIBLE_TYPE, vh->GetEntriesVisibleType(), itm_buffer_bytes(dev, d, ITM_BUF_VISIBLE_TYPE), 0), "download types");
  }
};
template <> struct HipMirror<ITMPlainVoxelArray> { static void VisibleList(const itm_scene*, ITMRenderState*) {} };

inline void HipMirrorImages(const itm_scene* dev, ITMRenderState* rs, bool range, bool rays, bool image) {
  itm_render_state* d = HipRenderStateOf(rs);
  const size_t px = (size_t)rs->raycastResult->noDims.x * rs->raycastResult->noDims.y;
  if (range) HipCheck(itm_download(dev, d, ITM_BUF_RANGE_IMAGE, rs->renderingRangeImage->GetData(MEMORYDEVICE_CPU), px * 8, 0), "download range image");
  if (rays) HipCheck(itm_download(dev, d, ITM_BUF_RAYCAST_RESULT, rs->raycastResult->GetData(MEMORYDEVICE_CPU), px * 16, 0), "download raycast result");
  if (image) HipCheck(itm_download(dev, d, ITM_BUF_RAYCAST_IMAGE, rs->raycastImage->GetData(MEMORYDEVICE_CPU), px * 4, 0), "download raycast image");
}
inline void HipMirrorForward(const itm_scene* dev, ITMRenderState* rs) {
  itm_render_state* d = HipRenderStateOf(rs);
  const size_t px = (size_t)rs->forwardProjection->noDims.x * rs->forwardProjection->noDims.y;
  HipCheck(itm_download(dev, d, ITM_BUF_FORWARD_PROJECTION, rs->forwardProjection->GetData(MEMORYDEVICE_CPU), px * 16, 0), "download");
  HipCheck(itm_download(dev, d, ITM_BUF_MISSING_POINTS, rs->fwdProjMissingPoints->GetData(MEMORYDEVICE_CPU), px * 4, 0), "download");
  itm_counters c; HipCheck(itm_get_counters(dev, d, &c, 0), "get_counters");
  rs->noFwdProjMissingPoints = c.noFwdProjMissingPoints;
}

// HIP_MIRROR_ON_DEMAND: everything of the render state's host shell that is older than its device twin.  TIndex picks the
// visible-list part (ITMRenderState_VH for the hash index).
template <class TIndex>
inline void HipSyncRenderStateToHost(ITMRenderState* rs) {
  HipRenderStateTwin* t = HipTwinOf(rs);
  if (t->staleList) HipMirror<TIndex>::VisibleList(t->devScene, rs);
  HipMirrorImages(t->devScene, rs, t->staleRange, t->staleRays, t->staleImage);
  if (t->staleForward) HipMirrorForward(t->devScene, rs);
  t->staleList = t->staleRange = t->staleRays = t->staleImage = t->staleForward = false;
}
// ... of a tracking state's point cloud (the ICP maps CreateICPMaps wrote, or the points of CreatePointCloud)
inline void HipSyncTrackingStateToHost(ITMTrackingState* ts) {
  HipRegistry::Maps& m = HipMapsOf(ts);
  if (!m.hostStale) return;
  const size_t n = m.count * 16;
  HipCheck(itm_memcpy_d2h(ts->pointCloud->locations->GetData(MEMORYDEVICE_CPU), m.points, n, 0), "memcpy_d2h");
  HipCheck(HipDownload(ts->pointCloud->colours->GetData(MEMORYDEVICE_CPU), m.normals, n, 0), "memcpy_d2h");
  m.hostStale = false;
}
// "The host wrote this tracking state's point cloud images itself" (a CPU visualisation engine, a file): the copies in HBM no longer
// say what the images say -- the depth tracker uploads the host images again instead of taking the maps in HBM.
inline void HipMarkTrackingStateHostWritten(const ITMTrackingState* ts) {
  auto it = HipRegistry::Get().maps.find(ts);
  if (it != HipRegistry::Get().maps.end()) { it->second.hostStale = false; it->second.icpMaps = false; }
}
// ... of a view whose float depth was produced in HBM (ITMViewBuilder_HIP)
inline void HipSyncViewToHost(ITMView* view) {
  HipRegistry::Stage& st = HipStageOf(view);
  if (!st.hostDepthStale) return;
  HipCheck(HipDownload(view->depth->GetData(MEMORYDEVICE_CPU), st.cur ? st.cur : st.depth, st.pixels * 4, 0), "memcpy_d2h");
  st.hostDepthStale = false;
}

// ---- ITMSceneReconstructionEngine ---------------------------------------------------------------------------
template <class TVoxel, class TIndex>
class ITMSceneReconstructionEngine_HIP : public ITMSceneReconstructionEngine<TVoxel, TIndex> {
  static bool Colour() { return HipVoxelTag<TVoxel>::value == ITM_VOXEL_S_RGB || HipVoxelTag<TVoxel>::value == ITM_VOXEL_F_RGB; }

 public:
  void ResetScene(ITMScene<TVoxel, TIndex>* scene) { HipCheck(itm_reset_scene(HipSceneOf(scene), 0), "ResetScene"); }

  // (colour scenes stage the rgb image with the first call of the frame, so that all calls of the frame name the same view)
  void AllocateSceneFromDepth(ITMScene<TVoxel, TIndex>* scene, const ITMView* view, const ITMTrackingState* trackingState,
                              const ITMRenderState* renderState, bool onlyUpdateVisibleList = false) {
    itm_scene* dev = HipSceneOf(scene);
    itm_view v = HipStageView(view, trackingState->pose_d, Colour());
    HipCheck(itm_allocate_scene_from_depth(dev, &v, HipRenderStateOf(renderState), onlyUpdateVisibleList ? 1 : 0, 0), "AllocateSceneFromDepth");
    if (HipEager()) HipMirror<TIndex>::VisibleList(dev, const_cast<ITMRenderState*>(renderState));
    else HipTwinOf(renderState)->staleList = true;
  }

  void IntegrateIntoScene(ITMScene<TVoxel, TIndex>* scene, const ITMView* view, const ITMTrackingState* trackingState,
                          const ITMRenderState* renderState) {
    itm_view v = HipStageView(view, trackingState->pose_d, Colour());
    HipCheck(itm_integrate_into_scene(HipSceneOf(scene), &v, HipRenderStateOf(renderState), 0), "IntegrateIntoScene");
  }

  // table, free lists, voxels and pool counters -> the host arrays of the reference's scene object
  static void SyncSceneToHost(ITMScene<TVoxel, TIndex>* scene);
};

template <class TVoxel, class TIndex> struct HipSceneSync;
template <class TVoxel> struct HipSceneSync<TVoxel, ITMVoxelBlockHash> {
  static void Run(ITMScene<TVoxel, ITMVoxelBlockHash>* scene, itm_scene* dev) {
    HipCheck(itm_download(dev, 0, ITM_BUF_HASH_ENTRIES, scene->index.GetEntries(), itm_buffer_bytes(dev, 0, ITM_BUF_HASH_ENTRIES), 0), "download table");
    HipCheck(itm_download(dev, 0, ITM_BUF_EXCESS_LIST, scene->index.GetExcessAllocationList(), itm_buffer_bytes(dev, 0, ITM_BUF_EXCESS_LIST), 0), "download excess list");
    itm_counters c; HipCheck(itm_get_counters(dev, 0, &c, 0), "get_counters");
    scene->index.SetLastFreeExcessListId(c.lastFreeExcessListId);
  }
};
template <class TVoxel> struct HipSceneSync<TVoxel, ITMPlainVoxelArray> { static void Run(ITMScene<TVoxel, ITMPlainVoxelArray>*, itm_scene*) {} };

template <class TVoxel, class TIndex>
void ITMSceneReconstructionEngine_HIP<TVoxel, TIndex>::SyncSceneToHost(ITMScene<TVoxel, TIndex>* scene) {
  itm_scene* dev = HipSceneOf(scene);
  HipSceneSync<TVoxel, TIndex>::Run(scene, dev);
  itm_counters c; HipCheck(itm_get_counters(dev, 0, &c, 0), "get_counters");
  scene->localVBA.lastFreeBlockId = c.lastFreeBlockId;
  HipCheck(itm_download(dev, 0, ITM_BUF_VOXEL_BLOCKS, scene->localVBA.GetVoxelBlocks(), itm_buffer_bytes(dev, 0, ITM_BUF_VOXEL_BLOCKS), 0), "download voxels");
  HipCheck(itm_download(dev, 0, ITM_BUF_ALLOCATION_LIST, scene->localVBA.GetAllocationList(), itm_buffer_bytes(dev, 0, ITM_BUF_ALLOCATION_LIST), 0), "download allocation list");
}

// ---- ITMVisualisationEngine ---------------------------------------------------------------------------------
template <class TVoxel, class TIndex>
class ITMVisualisationEngine_HIP : public ITMVisualisationEngine<TVoxel, TIndex> {
  static bool Colour() { return HipVoxelTag<TVoxel>::value == ITM_VOXEL_S_RGB || HipVoxelTag<TVoxel>::value == ITM_VOXEL_F_RGB; }
  itm_scene* Dev() const { return HipSceneOf(this->scene); }
  // after a call that wrote parts of the render state: back to the host now, or marked for HipSyncRenderStateToHost
  void Mirror(ITMRenderState* rs, bool range, bool rays, bool image) const {
    if (HipEager()) { HipMirrorImages(Dev(), rs, range, rays, image); return; }
    HipRenderStateTwin* t = HipTwinOf(rs);
    t->staleRange |= range; t->staleRays |= rays; t->staleImage |= image;
  }

 public:
  explicit ITMVisualisationEngine_HIP(const ITMScene<TVoxel, TIndex>* scene) : ITMVisualisationEngine<TVoxel, TIndex>(scene) {}
  ~ITMVisualisationEngine_HIP() { HipReleaseScene(this->scene); }

  typename IndexToRenderState<TIndex>::type* CreateRenderState(const Vector2i& imgSize) const {
    auto* rs =
        HipIndexTraits<TIndex>::NewRenderState(imgSize, this->scene->sceneParams->viewFrustum_min, this->scene->sceneParams->viewFrustum_max);
    HipCheck(itm_render_state_create(Dev(), imgSize.x, imgSize.y, &rs->dev), "CreateRenderState");
    rs->devScene = Dev();
    return rs;
  }

  void FindVisibleBlocks(const ITMPose* pose, const ITMIntrinsics* intrinsics, ITMRenderState* renderState) const {
    HipCheck(itm_find_visible_blocks(Dev(), pose->GetM().m, &intrinsics->projectionParamsSimple.all.x, HipRenderStateOf(renderState), 0), "FindVisibleBlocks");
    if (HipEager()) HipMirror<TIndex>::VisibleList(Dev(), renderState);
    else HipTwinOf(renderState)->staleList = true;
  }
  void CreateExpectedDepths(const ITMPose* pose, const ITMIntrinsics* intrinsics, ITMRenderState* renderState) const {
    HipCheck(itm_create_expected_depths(Dev(), pose->GetM().m, &intrinsics->projectionParamsSimple.all.x, HipRenderStateOf(renderState), 0), "CreateExpectedDepths");
    Mirror(renderState, true, false, false);
  }
  // (the caller reads outputImage as soon as this returns: a synchronous download under either policy)
  void RenderImage(const ITMPose* pose, const ITMIntrinsics* intrinsics, const ITMRenderState* renderState, ITMUChar4Image* outputImage,
                   IITMVisualisationEngine::RenderImageType type = IITMVisualisationEngine::RENDER_SHADED_GREYSCALE) const {
    const size_t px = (size_t)outputImage->noDims.x * outputImage->noDims.y;
    void* out = nullptr; HipCheck(itm_dev_malloc(&out, px * 4), "dev_malloc");
    // pixels the render does not touch keep their previous content, as on the host
    HipCheck(itm_memcpy_h2d(out, outputImage->GetData(MEMORYDEVICE_CPU), px * 4, 0), "memcpy_h2d");
    const int t = type == IITMVisualisationEngine::RENDER_COLOUR_FROM_VOLUME ? ITM_RENDER_COLOUR_FROM_VOLUME
                  : type == IITMVisualisationEngine::RENDER_COLOUR_FROM_NORMAL ? ITM_RENDER_COLOUR_FROM_NORMAL : ITM_RENDER_SHADED_GREYSCALE;
    int rc = itm_render_image(Dev(), pose->GetM().m, &intrinsics->projectionParamsSimple.all.x, HipRenderStateOf(renderState), (uint8_t*)out, t, 0);
    if (rc == ITM_OK) rc = HipDownload(outputImage->GetData(MEMORYDEVICE_CPU), out, px * 4, 0);
    itm_dev_free(out);
    HipCheck(rc, "RenderImage");
    Mirror(const_cast<ITMRenderState*>(renderState), false, true, false);
  }
  void FindSurface(const ITMPose* pose, const ITMIntrinsics* intrinsics, const ITMRenderState* renderState) const {
    HipCheck(itm_find_surface(Dev(), pose->GetM().m, &intrinsics->projectionParamsSimple.all.x, HipRenderStateOf(renderState), 0), "FindSurface");
    Mirror(const_cast<ITMRenderState*>(renderState), false, true, false);
  }
  // (noTotalPoints is a host member the colour tracker sizes its loops with: read back under either policy)
  void CreatePointCloud(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState, bool skipPoints) const {
    itm_view v = HipStageView(view, trackingState->pose_d, Colour());
    HipRegistry::Maps& m = HipMapsOf(trackingState);
    HipCheck(itm_create_point_cloud(Dev(), &v, HipRenderStateOf(renderState), skipPoints ? 1 : 0, (float*)m.points, (float*)m.normals, 0), "CreatePointCloud");
    itm_counters c; HipCheck(itm_get_counters(Dev(), HipRenderStateOf(renderState), &c, 0), "get_counters");
    trackingState->pointCloud->noTotalPoints = c.noTotalPoints;
    m.count = (size_t)c.noTotalPoints; m.hostStale = true; m.icpMaps = false;
    if (HipEager()) HipSyncTrackingStateToHost(trackingState);
    trackingState->pose_pointCloud->SetFrom(trackingState->pose_d);
    Mirror(renderState, false, true, false);
  }
  void CreateICPMaps(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState) const {
    itm_view v = HipStageView(view, trackingState->pose_d, Colour());
    HipRegistry::Maps& m = HipMapsOf(trackingState);
    HipCheck(itm_create_icp_maps(Dev(), &v, HipRenderStateOf(renderState), (float*)m.points, (float*)m.normals, 0), "CreateICPMaps");
    m.count = m.pixels; m.hostStale = true; m.icpMaps = true;
    if (HipEager()) HipSyncTrackingStateToHost(trackingState);
    trackingState->pose_pointCloud->SetFrom(trackingState->pose_d);   // ITMVisualisationEngine_CPU.cpp CreateICPMaps
    Mirror(renderState, false, true, true);
  }
  void ForwardRender(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState) const {
    itm_view v = HipStageView(view, trackingState->pose_d, Colour());
    HipCheck(itm_forward_render(Dev(), &v, HipRenderStateOf(renderState), 0), "ForwardRender");
    if (HipEager()) HipMirrorForward(Dev(), renderState);
    else HipTwinOf(renderState)->staleForward = true;
  }
};


// ITMMeshingEngine<TVoxel, TIndex> (Engine/ITMMeshingEngine.h:19-26) on the device scene twin: the mesh is built in HBM in the
// reference's triangle order and mirrored into the reference's ITMMesh (triangles + noTotalTriangles), so WriteOBJ / WriteSTL of
// the reference object work unchanged.
template <class TVoxel, class TIndex>
class ITMMeshingEngine_HIP : public ITMMeshingEngine<TVoxel, TIndex> {
  itm_mesh* dev = nullptr;
  const void* owner = nullptr;
 public:
  ~ITMMeshingEngine_HIP() { itm_mesh_destroy(dev); }
  void MeshScene(ITMMesh* mesh, const ITMScene<TVoxel, TIndex>* scene) {
    itm_scene* sc = HipSceneOf(scene);
    if (!dev || owner != scene) {
      itm_mesh_destroy(dev); dev = nullptr;
      HipCheck(itm_mesh_create(sc, ITMMesh::noMaxTriangles, &dev), "itm_mesh_create");
      owner = scene;
    }
    HipCheck(itm_mesh_scene(sc, dev, 0), "itm_mesh_scene");
    uint32_t n = 0;
    mesh->triangles->Clear();
    HipCheck(itm_mesh_download(dev, (float*)mesh->triangles->GetData(MEMORYDEVICE_CPU), ITMMesh::noMaxTriangles, &n, 0), "itm_mesh_download");
    mesh->noTotalTriangles = n;
  }
};

}  // namespace Engine
}  // namespace ITMLib
