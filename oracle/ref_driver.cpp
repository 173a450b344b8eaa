// ref_driver.cpp -- TEST INFRASTRUCTURE.  Thin C-ABI shim (prefix `itmr_`, same signatures as
// include/itm_hip.h) over the REAL reference CPU engines, compiled by oracle/Makefile target
// `ref` straight from the sources where they lie under /root/reference (never copied into this
// repo).  Output goes to oracle/_ref/ only (git-ignored).  It exists to pin oracle/itm_oracle.cpp
// and to generate tests/golden/; it is absent on machines without /root/reference and nothing in
// the product depends on it.
//
// The reference fixes the voxel/index types and the pool sizes at compile time
// (Utils/ITMLibDefines.h:37-62,205,210).  This shim instantiates the reference templates for all
// four voxel types and both index types, but can only offer the pool sizes the reference was
// compiled with (bucket 0x100000, excess 0x20000, local blocks 0x10000); other configurations
// return ITM_ERR_UNSUPPORTED.

#define ITM_FN(name) itmr_##name
#include "../include/itm_hip.h"
#include "../include/itm_debug.h"

#include <cstring>
#include <string>

// The two engine translation units are included so that extra <voxel, index> combinations can be
// instantiated; each ends with an explicit instantiation for
// <ITMVoxel_s, ITMVoxelBlockHash>.
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMMeshingEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMSwappingEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceAgnostic/ITMViewBuilder.h"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMDepthTracker_CPU.h"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMLowLevelEngine_CPU.h"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMViewBuilder_CPU.h"
#include "ITMLib/Utils/ITMCalibIO.h"
#include "Utils/FileUtils.h"
#include "ORUtils/MemoryBlockPersister.h"
#include <stdexcept>

using namespace ITMLib::Engine;
using namespace ITMLib::Objects;

template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_f, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_s_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_f, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_s_rgb, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_f_rgb, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMSwappingEngine_CPU<ITMVoxel_f, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMSwappingEngine_CPU<ITMVoxel_s_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMSwappingEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_f, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_s_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_f, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_s_rgb, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_f_rgb, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_f, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_s_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_f, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_s_rgb, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_f_rgb, ITMPlainVoxelArray>;

namespace {
thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }

struct RefSceneBase {
  itm_scene_config cfg;
  itm_scene_params prm;
  virtual ~RefSceneBase() {}
  virtual void reset() = 0;
  virtual ITMRenderState* createRenderState(int w, int h) = 0;
  virtual void allocate(const ITMView*, const ITMTrackingState*, ITMRenderState*, bool) = 0;
  virtual void integrate(const ITMView*, const ITMTrackingState*, ITMRenderState*) = 0;
  virtual void findVisible(const ITMPose*, const ITMIntrinsics*, ITMRenderState*) = 0;
  virtual void expectedDepths(const ITMPose*, const ITMIntrinsics*, ITMRenderState*) = 0;
  virtual void renderImage(const ITMPose*, const ITMIntrinsics*, ITMRenderState*, ITMUChar4Image*, int) = 0;
  virtual void findSurface(const ITMPose*, const ITMIntrinsics*, ITMRenderState*) = 0;
  virtual void pointCloud(const ITMView*, ITMTrackingState*, ITMRenderState*, bool) = 0;
  virtual void icpMaps(const ITMView*, ITMTrackingState*, ITMRenderState*) = 0;
  virtual void forwardRender(const ITMView*, ITMTrackingState*, ITMRenderState*) = 0;
  virtual void mesh(ITMMesh*) = 0;
  virtual void swapIn(ITMRenderState*) = 0;
  virtual void swapOut(ITMRenderState*) = 0;
  virtual bool cacheGet(int entry, void* dst) = 0;
  virtual void cacheFlags(uint8_t* dst, size_t n) = 0;
  virtual void* buffer(int which, size_t* bytes) = 0;
  virtual int* lastFreeBlockId() = 0;
  virtual int lastFreeExcess() = 0;
  virtual void setLastFreeExcess(int) = 0;
};

template <class TIndex> struct IndexOps;
template <> struct IndexOps<ITMVoxelBlockHash> {
  static void configure(ITMVoxelBlockHash&, const itm_scene_config&) {}
  static void* hashEntries(ITMVoxelBlockHash& i, size_t* b) { *b = (size_t)ITMVoxelBlockHash::noTotalEntries * sizeof(ITMHashEntry); return i.GetEntries(); }
  static void* excess(ITMVoxelBlockHash& i, size_t* b) { *b = (size_t)SDF_EXCESS_LIST_SIZE * 4; return i.GetExcessAllocationList(); }
  static int lastExcess(ITMVoxelBlockHash& i) { return i.GetLastFreeExcessListId(); }
  static void setLastExcess(ITMVoxelBlockHash& i, int v) { i.SetLastFreeExcessListId(v); }
  static size_t voxels(ITMVoxelBlockHash&) { return (size_t)SDF_LOCAL_BLOCK_NUM * SDF_BLOCK_SIZE3; }
  static size_t allocEntries() { return SDF_LOCAL_BLOCK_NUM; }
};
template <> struct IndexOps<ITMPlainVoxelArray> {
  // shrink / move the array inside the 512^3 allocation the reference always makes
  static void configure(ITMPlainVoxelArray& i, const itm_scene_config& c) {
    ITMPlainVoxelArray::IndexData* d = const_cast<ITMPlainVoxelArray::IndexData*>(i.getIndexData());
    d->size = Vector3i(c.denseSize[0], c.denseSize[1], c.denseSize[2]);
    d->offset = Vector3i(c.denseOffset[0], c.denseOffset[1], c.denseOffset[2]);
  }
  static void* hashEntries(ITMPlainVoxelArray&, size_t* b) { *b = 0; return 0; }
  static void* excess(ITMPlainVoxelArray&, size_t* b) { *b = 0; return 0; }
  static int lastExcess(ITMPlainVoxelArray&) { return 0; }
  static void setLastExcess(ITMPlainVoxelArray&, int) {}
  static size_t voxels(ITMPlainVoxelArray& i) { Vector3i s = i.getVolumeSize(); return (size_t)s.x * s.y * s.z; }
  static size_t allocEntries() { return 1; }
};

template <class TVoxel, class TIndex>
struct RefScene : RefSceneBase {
  ITMSceneParams sp;
  ITMScene<TVoxel, TIndex> scene;
  ITMSceneReconstructionEngine_CPU<TVoxel, TIndex> reco;
  ITMVisualisationEngine_CPU<TVoxel, TIndex> vis;
  ITMMeshingEngine_CPU<TVoxel, TIndex> mesher;
  ITMSwappingEngine_CPU<TVoxel, TIndex> swapper;
  RefScene(const itm_scene_config& c, const itm_scene_params& p)
      : sp(p.mu, p.maxW, p.voxelSize, p.viewFrustum_min, p.viewFrustum_max, p.stopIntegratingAtMaxW != 0),
        scene(&sp, c.useSwapping != 0 && c.indexType == ITM_INDEX_HASH, MEMORYDEVICE_CPU), vis(&scene) {
    cfg = c; prm = p;
    IndexOps<TIndex>::configure(scene.index, c);
  }
  void reset() override { reco.ResetScene(&scene); }
  ITMRenderState* createRenderState(int w, int h) override { return vis.CreateRenderState(Vector2i(w, h)); }
  void allocate(const ITMView* v, const ITMTrackingState* t, ITMRenderState* r, bool o) override { reco.AllocateSceneFromDepth(&scene, v, t, r, o); }
  void integrate(const ITMView* v, const ITMTrackingState* t, ITMRenderState* r) override { reco.IntegrateIntoScene(&scene, v, t, r); }
  void findVisible(const ITMPose* p, const ITMIntrinsics* i, ITMRenderState* r) override { vis.FindVisibleBlocks(p, i, r); }
  void expectedDepths(const ITMPose* p, const ITMIntrinsics* i, ITMRenderState* r) override { vis.CreateExpectedDepths(p, i, r); }
  void renderImage(const ITMPose* p, const ITMIntrinsics* i, ITMRenderState* r, ITMUChar4Image* o, int type) override {
    vis.RenderImage(p, i, r, o, (IITMVisualisationEngine::RenderImageType)type);
  }
  void findSurface(const ITMPose* p, const ITMIntrinsics* i, ITMRenderState* r) override { vis.FindSurface(p, i, r); }
  void pointCloud(const ITMView* v, ITMTrackingState* t, ITMRenderState* r, bool skip) override { vis.CreatePointCloud(v, t, r, skip); }
  void icpMaps(const ITMView* v, ITMTrackingState* t, ITMRenderState* r) override { vis.CreateICPMaps(v, t, r); }
  void forwardRender(const ITMView* v, ITMTrackingState* t, ITMRenderState* r) override { vis.ForwardRender(v, t, r); }
  void mesh(ITMMesh* m) override { mesher.MeshScene(m, &scene); }
  void swapIn(ITMRenderState* r) override { swapper.IntegrateGlobalIntoLocal(&scene, r); }
  void swapOut(ITMRenderState* r) override { swapper.SaveToGlobalMemory(&scene, r); }
  bool cacheGet(int entry, void* dst) override {
    if (!scene.useSwapping || !scene.globalCache->HasStoredData(entry)) return false;
    if (dst) std::memcpy(dst, scene.globalCache->GetStoredVoxelBlock(entry), sizeof(TVoxel) * SDF_BLOCK_SIZE3);
    return true;
  }
  void cacheFlags(uint8_t* dst, size_t n) override { for (size_t i = 0; i < n; ++i) dst[i] = scene.useSwapping && scene.globalCache->HasStoredData((int)i) ? 1 : 0; }
  void* buffer(int which, size_t* bytes) override {
    if (which == ITM_BUF_SWAP_STATES) {
      if (!scene.useSwapping) { *bytes = 0; return 0; }
      *bytes = (size_t)scene.globalCache->noTotalEntries * sizeof(ITMHashSwapState);
      return scene.globalCache->GetSwapStates(false);
    }
    switch (which) {
      case ITM_BUF_HASH_ENTRIES: return IndexOps<TIndex>::hashEntries(scene.index, bytes);
      case ITM_BUF_EXCESS_LIST: return IndexOps<TIndex>::excess(scene.index, bytes);
      case ITM_BUF_VOXEL_BLOCKS: *bytes = IndexOps<TIndex>::voxels(scene.index) * sizeof(TVoxel); return scene.localVBA.GetVoxelBlocks();
      case ITM_BUF_ALLOCATION_LIST: *bytes = IndexOps<TIndex>::allocEntries() * 4; return scene.localVBA.GetAllocationList();
    }
    *bytes = 0; return 0;
  }
  int* lastFreeBlockId() override { return &scene.localVBA.lastFreeBlockId; }
  int lastFreeExcess() override { return IndexOps<TIndex>::lastExcess(scene.index); }
  void setLastFreeExcess(int v) override { IndexOps<TIndex>::setLastExcess(scene.index, v); }
};

template <class TVoxel>
RefSceneBase* make_scene(const itm_scene_config& c, const itm_scene_params& p) {
  if (c.indexType == ITM_INDEX_HASH) return new RefScene<TVoxel, ITMVoxelBlockHash>(c, p);
  return new RefScene<TVoxel, ITMPlainVoxelArray>(c, p);
}
}  // namespace

struct itm_scene { RefSceneBase* impl; };

struct itm_render_state {
  ITMRenderState* rs;
  bool hash;
  int w, h;
  // the objects the reference engines read their inputs from
  ITMRGBDCalib calib;
  ITMView* view;
  ITMTrackingState* ts;
  Vector2i rgbSize;
  int noTotalPoints;
  itm_render_state() : rs(0), view(0), ts(0), noTotalPoints(0) {}
};

namespace {
void set_matrix(Matrix4f& m, const float* src) { std::memcpy(m.m, src, 64); }

// Fills the reference's ITMView / ITMTrackingState from the flat itm_view.
void load_view(itm_render_state* r, const itm_view* v) {
  Vector2i dsz(v->w, v->h), csz(v->w_rgb > 0 ? v->w_rgb : v->w, v->h_rgb > 0 ? v->h_rgb : v->h);
  if (!r->view || r->rgbSize.x != csz.x || r->rgbSize.y != csz.y) {
    delete r->view;
    r->view = new ITMView(&r->calib, csz, dsz, false);
    r->rgbSize = csz;
  }
  r->view->calib->intrinsics_d.SetFrom(v->intr_d[0], v->intr_d[1], v->intr_d[2], v->intr_d[3], (float)v->w, (float)v->h);
  r->view->calib->intrinsics_rgb.SetFrom(v->intr_rgb[0], v->intr_rgb[1], v->intr_rgb[2], v->intr_rgb[3], (float)csz.x, (float)csz.y);
  set_matrix(r->view->calib->trafo_rgb_to_depth.calib, v->rgb_to_depth);
  set_matrix(r->view->calib->trafo_rgb_to_depth.calib_inv, v->rgb_to_depth_inv);
  std::memcpy(r->view->depth->GetData(MEMORYDEVICE_CPU), v->depth, (size_t)v->w * v->h * sizeof(float));
  if (v->rgb) std::memcpy(r->view->rgb->GetData(MEMORYDEVICE_CPU), v->rgb, (size_t)csz.x * csz.y * 4);
  Matrix4f M; set_matrix(M, v->M_d);
  r->ts->pose_d->SetM(M);
}
}  // namespace

extern "C" {

const char* itmr_version(void) { return "itm-ref 1 (reference CPU engines compiled from /root/reference)"; }
const char* itmr_last_error(void) { return g_err.c_str(); }
int itmr_uses_device_memory(void) { return 0; }
size_t itmr_voxel_size_bytes(int t) {
  switch (t) { case ITM_VOXEL_S: return sizeof(ITMVoxel_s); case ITM_VOXEL_F: return sizeof(ITMVoxel_f);
    case ITM_VOXEL_S_RGB: return sizeof(ITMVoxel_s_rgb); case ITM_VOXEL_F_RGB: return sizeof(ITMVoxel_f_rgb); }
  return 0;
}
int itmr_dev_malloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? ITM_OK : ITM_ERR_DEVICE; }
int itmr_dev_free(void* p) { std::free(p); return ITM_OK; }
int itmr_memcpy_h2d(void* d, const void* s, size_t n, itm_stream) { std::memcpy(d, s, n); return ITM_OK; }
int itmr_memcpy_d2h(void* d, const void* s, size_t n, itm_stream) { std::memcpy(d, s, n); return ITM_OK; }
int itmr_stream_synchronize(itm_stream) { return ITM_OK; }
int itmr_set_device(int) { return ITM_OK; }
int itmr_debug_set(int, int) { return ITM_OK; }
int itmr_debug_divide(int mode, const float* a, const float* b, const float*, float* out, int n, itm_stream) { for (int i = 0; i < n; ++i) out[i] = (mode == 3) ? 1.0f / b[i] : a[i] / b[i]; return ITM_OK; }
int itmr_debug_div32767(const float* in, float* out, int n, itm_stream) { for (int i = 0; i < n; ++i) out[i] = in[i] / 32767.0f; return ITM_OK; }
int itmr_profile_enable(itm_scene*, uint32_t) { return ITM_OK; }
int itmr_profile_sample(itm_scene*, int) { return ITM_OK; }
int itmr_profile_calibrate(itm_scene*, int, itm_stream) { return ITM_OK; }
int itmr_profile_read(itm_scene*, itm_profile* out, int) { if (out) std::memset(out, 0, sizeof *out); return ITM_OK; }

int itmr_scene_create(const itm_scene_config* cin, const itm_scene_params* prm, itm_scene** out) {
  if (!cin || !prm || !out) return fail(ITM_ERR_INVALID, "null argument");
  itm_scene_config c = *cin;
  if (c.bucketNum == 0) c.bucketNum = SDF_BUCKET_NUM;
  if (c.excessNum == 0) c.excessNum = SDF_EXCESS_LIST_SIZE;
  if (c.localBlockNum == 0) c.localBlockNum = SDF_LOCAL_BLOCK_NUM;
  if (c.denseSize[0] == 0 && c.denseSize[1] == 0 && c.denseSize[2] == 0) {
    c.denseSize[0] = c.denseSize[1] = c.denseSize[2] = 512;
    if (!c.denseOffsetSet) { c.denseOffset[0] = -256; c.denseOffset[1] = -256; c.denseOffset[2] = 0; }
  }
  c.denseOffsetSet = 1;
  if (c.maxRenderingBlocks == 0) c.maxRenderingBlocks = MAX_RENDERING_BLOCKS;
  if (c.maxRenderingBlocks != MAX_RENDERING_BLOCKS) return fail(ITM_ERR_UNSUPPORTED, "MAX_RENDERING_BLOCKS is a compile-time constant of the reference");
  if (c.indexType == ITM_INDEX_HASH &&
      (c.bucketNum != SDF_BUCKET_NUM || c.excessNum != SDF_EXCESS_LIST_SIZE || c.localBlockNum != SDF_LOCAL_BLOCK_NUM))
    return fail(ITM_ERR_UNSUPPORTED, "reference pool sizes are compile-time constants");
  if (c.transferBlockNum != 0 && c.transferBlockNum != SDF_TRANSFER_BLOCK_NUM) return fail(ITM_ERR_UNSUPPORTED, "SDF_TRANSFER_BLOCK_NUM is a compile-time constant of the reference");
  if (c.indexType == ITM_INDEX_DENSE && (size_t)c.denseSize[0] * c.denseSize[1] * c.denseSize[2] > (size_t)512 * 512 * 512)
    return fail(ITM_ERR_UNSUPPORTED, "dense array larger than the reference allocation");
  RefSceneBase* impl = 0;
  switch (c.voxelType) {
    case ITM_VOXEL_S: impl = make_scene<ITMVoxel_s>(c, *prm); break;
    case ITM_VOXEL_F: impl = make_scene<ITMVoxel_f>(c, *prm); break;
    case ITM_VOXEL_S_RGB: impl = make_scene<ITMVoxel_s_rgb>(c, *prm); break;
    case ITM_VOXEL_F_RGB: impl = make_scene<ITMVoxel_f_rgb>(c, *prm); break;
    default: return fail(ITM_ERR_INVALID, "unknown voxel type");
  }
  *out = new itm_scene{impl};
  return ITM_OK;
}
int itmr_scene_destroy(itm_scene* s) { if (s) { delete s->impl; delete s; } return ITM_OK; }
int itmr_scene_get_config(const itm_scene* s, itm_scene_config* c, itm_scene_params* p) {
  if (c) *c = s->impl->cfg;
  if (p) *p = s->impl->prm;
  return ITM_OK;
}
int itmr_reset_scene(itm_scene* s, itm_stream) { s->impl->reset(); return ITM_OK; }

int itmr_render_state_create(const itm_scene* s, int w, int h, itm_render_state** out) {
  itm_render_state* r = new itm_render_state();
  r->rs = s->impl->createRenderState(w, h);
  r->hash = s->impl->cfg.indexType == ITM_INDEX_HASH;
  r->w = w; r->h = h;
  r->ts = new ITMTrackingState(Vector2i(w, h), MEMORYDEVICE_CPU);
  *out = r;
  return ITM_OK;
}
int itmr_render_state_destroy(itm_render_state* r) {
  if (r) { delete r->rs; delete r->view; delete r->ts; delete r; }
  return ITM_OK;
}

int itmr_allocate_scene_from_depth(itm_scene* s, const itm_view* v, itm_render_state* r, int only, itm_stream) {
  load_view(r, v);
  s->impl->allocate(r->view, r->ts, r->rs, only != 0);
  return ITM_OK;
}
int itmr_integrate_into_scene(itm_scene* s, const itm_view* v, itm_render_state* r, itm_stream) {
  load_view(r, v);
  s->impl->integrate(r->view, r->ts, r->rs);
  return ITM_OK;
}
static void pose_intr(const float M[16], const float intr[4], ITMPose& pose, ITMIntrinsics& in) {
  Matrix4f m; set_matrix(m, M); pose.SetM(m);
  in.SetFrom(intr[0], intr[1], intr[2], intr[3], 0, 0);
}
int itmr_find_visible_blocks(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* r, itm_stream) {
  ITMPose pose; ITMIntrinsics in; pose_intr(M, intr, pose, in);
  s->impl->findVisible(&pose, &in, r->rs);
  return ITM_OK;
}
int itmr_create_expected_depths(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* r, itm_stream) {
  ITMPose pose; ITMIntrinsics in; pose_intr(M, intr, pose, in);
  s->impl->expectedDepths(&pose, &in, r->rs);
  return ITM_OK;
}
int itmr_render_image(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* r, uint8_t* out, int type, itm_stream) {
  ITMPose pose; ITMIntrinsics in; pose_intr(M, intr, pose, in);
  if (!out) { s->impl->renderImage(&pose, &in, r->rs, r->rs->raycastImage, type); return ITM_OK; }
  ITMUChar4Image img(Vector2i(r->w, r->h), true, false);
  std::memcpy(img.GetData(MEMORYDEVICE_CPU), out, (size_t)r->w * r->h * 4);
  s->impl->renderImage(&pose, &in, r->rs, &img, type);
  std::memcpy(out, img.GetData(MEMORYDEVICE_CPU), (size_t)r->w * r->h * 4);
  return ITM_OK;
}
int itmr_find_surface(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* r, itm_stream) {
  ITMPose pose; ITMIntrinsics in; pose_intr(M, intr, pose, in);
  s->impl->findSurface(&pose, &in, r->rs);
  return ITM_OK;
}
int itmr_create_point_cloud(const itm_scene* s, const itm_view* v, itm_render_state* r, int skip, float* loc, float* col, itm_stream) {
  load_view(r, v);
  s->impl->pointCloud(r->view, r->ts, r->rs, skip != 0);
  size_t n = r->ts->pointCloud->noTotalPoints;
  r->noTotalPoints = (int)n;
  std::memcpy(loc, r->ts->pointCloud->locations->GetData(MEMORYDEVICE_CPU), n * 16);
  std::memcpy(col, r->ts->pointCloud->colours->GetData(MEMORYDEVICE_CPU), n * 16);
  return ITM_OK;
}
int itmr_create_icp_maps(const itm_scene* s, const itm_view* v, itm_render_state* r, float* pts, float* nrm, itm_stream) {
  load_view(r, v);
  s->impl->icpMaps(r->view, r->ts, r->rs);
  size_t n = (size_t)r->w * r->h;
  std::memcpy(pts, r->ts->pointCloud->locations->GetData(MEMORYDEVICE_CPU), n * 16);
  std::memcpy(nrm, r->ts->pointCloud->colours->GetData(MEMORYDEVICE_CPU), n * 16);
  return ITM_OK;
}
int itmr_forward_render(const itm_scene* s, const itm_view* v, itm_render_state* r, itm_stream) {
  load_view(r, v);
  s->impl->forwardRender(r->view, r->ts, r->rs);
  return ITM_OK;
}
// the reference's ITMSwappingEngine_CPU and ITMGlobalCache (scenes created with useSwapping)
int itmr_swap_integrate_global_into_local(itm_scene* s, itm_render_state* r, itm_stream) {
  if (!s->impl->cfg.useSwapping) return fail(ITM_ERR_INVALID, "scene without swapping");
  s->impl->swapIn(r->rs); return ITM_OK;
}
int itmr_swap_save_to_global_memory(itm_scene* s, itm_render_state* r, itm_stream) {
  if (!s->impl->cfg.useSwapping) return fail(ITM_ERR_INVALID, "scene without swapping");
  s->impl->swapOut(r->rs); return ITM_OK;
}
int itmr_global_cache_get(const itm_scene* s, int entry, void* dst, int* has) { *has = s->impl->cacheGet(entry, dst) ? 1 : 0; return ITM_OK; }
int itmr_global_cache_flags(const itm_scene* s, uint8_t* dst, size_t bytes) { s->impl->cacheFlags(dst, bytes); return ITM_OK; }

int itmr_process_frame(itm_scene* s, const itm_view* v, itm_render_state* r, float* pts, float* nrm, itm_stream st) {
  // ITMDenseMapper::ProcessFrame + ITMTrackingController::Prepare (requiresFullRendering)
  load_view(r, v);
  s->impl->allocate(r->view, r->ts, r->rs, false);
  s->impl->integrate(r->view, r->ts, r->rs);
  s->impl->expectedDepths(r->ts->pose_d, &r->view->calib->intrinsics_d, r->rs);
  return itmr_create_icp_maps(s, v, r, pts, nrm, st);
}

int itmr_process_frame_ahead(itm_scene* s, const itm_view* v, const itm_view*, itm_render_state* r, float* pts, float* nrm, itm_stream st) {
  return itmr_process_frame(s, v, r, pts, nrm, st);
}
int itmr_flush(itm_scene*, itm_render_state*, itm_stream) { return ITM_OK; }          // (nothing is ever recorded or issued ahead here)
int itmr_cancel_ahead(itm_scene*, itm_render_state*, itm_stream) { return ITM_OK; }

// ---- meshing: the reference's ITMMesh + ITMMeshingEngine_CPU ----------------------------------------------------------------
struct itm_mesh { const itm_scene* scene; ITMMesh* mesh; };
int itmr_mesh_create(const itm_scene* s, uint32_t maxTriangles, itm_mesh** out) {
  if (maxTriangles != 0 && maxTriangles != ITMMesh::noMaxTriangles) return fail(ITM_ERR_UNSUPPORTED, "ITMMesh::noMaxTriangles is a compile-time constant of the reference");
  *out = new itm_mesh{s, new ITMMesh(MEMORYDEVICE_CPU)};
  return ITM_OK;
}
int itmr_mesh_destroy(itm_mesh* m) { if (m) { delete m->mesh; delete m; } return ITM_OK; }
int itmr_mesh_scene(const itm_scene* s, itm_mesh* m, itm_stream) { s->impl->mesh(m->mesh); return ITM_OK; }
int itmr_mesh_info(const itm_mesh* m, uint32_t* n, uint32_t* cap, const float** tri, itm_stream) {
  if (n) *n = m->mesh->noTotalTriangles;
  if (cap) *cap = ITMMesh::noMaxTriangles;
  if (tri) *tri = (const float*)m->mesh->triangles->GetData(MEMORYDEVICE_CPU);
  return ITM_OK;
}
int itmr_mesh_download(const itm_mesh* m, float* dst, uint32_t capacity, uint32_t* n, itm_stream) {
  *n = m->mesh->noTotalTriangles;
  const uint32_t k = *n < capacity ? *n : capacity;
  if (k) std::memcpy(dst, m->mesh->triangles->GetData(MEMORYDEVICE_CPU), (size_t)k * sizeof(ITMMesh::Triangle));
  return ITM_OK;
}
int itmr_mesh_write_obj(const itm_mesh* m, const char* path, itm_stream) { m->mesh->WriteOBJ(path); return ITM_OK; }
int itmr_mesh_write_stl(const itm_mesh* m, const char* path, itm_stream) { m->mesh->WriteSTL(path); return ITM_OK; }

int itmr_convert_depth_affine(const int16_t* raw, float* out, int w, int h, float a, float b, itm_stream) {
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) convertDepthAffineToFloat(out, x, y, raw, Vector2i(w, h), Vector2f(a, b));
  return ITM_OK;
}
int itmr_convert_disparity(const int16_t* raw, float* out, int w, int h, float c0, float c1, float fx, itm_stream) {
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) convertDisparityToDepth(out, x, y, raw, Vector2f(c0, c1), fx, Vector2i(w, h));
  return ITM_OK;
}

// the reference's own file readers / writers (Utils/FileUtils.cpp, ITMLib/Utils/ITMCalibIO.cpp)
int itmr_read_depth_image(const char* path, int16_t* dst, int cap, int* w, int* h) {
  ITMShortImage img(Vector2i(2, 2), true, false);
  if (!ReadImageFromFile(&img, path)) return ITM_ERR_INVALID;
  if (img.noDims.x * img.noDims.y > cap) return ITM_ERR_INVALID;
  *w = img.noDims.x; *h = img.noDims.y;
  std::memcpy(dst, img.GetData(MEMORYDEVICE_CPU), (size_t)*w * *h * 2);
  return ITM_OK;
}
int itmr_read_rgb_image(const char* path, uint8_t* dst, int cap, int* w, int* h) {
  ITMUChar4Image img(Vector2i(2, 2), true, false);
  if (!ReadImageFromFile(&img, path)) return ITM_ERR_INVALID;
  if (img.noDims.x * img.noDims.y > cap) return ITM_ERR_INVALID;
  *w = img.noDims.x; *h = img.noDims.y;
  std::memcpy(dst, img.GetData(MEMORYDEVICE_CPU), (size_t)*w * *h * 4);
  return ITM_OK;
}
int itmr_write_depth_image(const char* path, const int16_t* src, int w, int h) {
  ITMShortImage img(Vector2i(w, h), true, false);
  std::memcpy(img.GetData(MEMORYDEVICE_CPU), src, (size_t)w * h * 2);
  SaveImageToFile(&img, path);
  return ITM_OK;
}
int itmr_write_rgb_image(const char* path, const uint8_t* src, int w, int h) {
  ITMUChar4Image img(Vector2i(w, h), true, false);
  std::memcpy(img.GetData(MEMORYDEVICE_CPU), src, (size_t)w * h * 4);
  SaveImageToFile(&img, path, false);
  return ITM_OK;
}
int itmr_write_float_depth_image(const char* path, const float* src, int w, int h) {
  ITMFloatImage img(Vector2i(w, h), true, false);
  std::memcpy(img.GetData(MEMORYDEVICE_CPU), src, (size_t)w * h * 4);
  SaveImageToFile(&img, path);
  return ITM_OK;
}
int itmr_read_rgbd_calib(const char* path, itm_rgbd_calib* out) {
  ITMRGBDCalib c;
  std::memset(out, 0, sizeof *out);
  if (!readRGBDCalib(path, c)) return ITM_ERR_INVALID;
  std::memcpy(out->intr_rgb, &c.intrinsics_rgb.projectionParamsSimple.all, 16);
  std::memcpy(out->intr_d, &c.intrinsics_d.projectionParamsSimple.all, 16);
  std::memcpy(out->rgb_to_depth, c.trafo_rgb_to_depth.calib.m, 64);
  std::memcpy(out->rgb_to_depth_inv, c.trafo_rgb_to_depth.calib_inv.m, 64);
  out->disparityType = c.disparityCalib.type == ITMDisparityCalib::TRAFO_KINECT ? 0 : 1;
  out->disparityParams[0] = c.disparityCalib.params.x; out->disparityParams[1] = c.disparityCalib.params.y;
  return ITM_OK;
}

// ORUtils::MemoryBlockPersister reads a block file written by the product's checkpoint
int itmr_debug_load_hash_block(const char* path, void* dst, int maxEntries) {
  try {
    ORUtils::MemoryBlock<ITMHashEntry>* b = ORUtils::MemoryBlockPersister::LoadMemoryBlock<ITMHashEntry>(std::string(path));
    const int n = (int)b->dataSize;
    if (n > maxEntries) { delete b; return -1; }
    std::memcpy(dst, b->GetData(MEMORYDEVICE_CPU), (size_t)n * sizeof(ITMHashEntry));
    delete b;
    return n;
  } catch (...) { return -1; }
}

// the reference's own view builder (ITMViewBuilder_CPU) on host images
int itmr_filter_depth(const float* in, float* out, int w, int h, itm_stream) {
  ITMRGBDCalib calib; ITMViewBuilder_CPU vb(&calib);
  ITMFloatImage a(Vector2i(w, h), true, false), b(Vector2i(w, h), true, false);
  std::memcpy(a.GetData(MEMORYDEVICE_CPU), in, (size_t)w * h * 4);
  vb.DepthFiltering(&b, &a);
  std::memcpy(out, b.GetData(MEMORYDEVICE_CPU), (size_t)w * h * 4);
  return ITM_OK;
}
int itmr_compute_normal_and_weights(const float* depth, float* normals, float* sigmaZ, int w, int h, const float intr[4], itm_stream) {
  ITMRGBDCalib calib; ITMViewBuilder_CPU vb(&calib);
  ITMFloatImage d(Vector2i(w, h), true, false), sz(Vector2i(w, h), true, false);
  ITMFloat4Image n(Vector2i(w, h), true, false);
  std::memcpy(d.GetData(MEMORYDEVICE_CPU), depth, (size_t)w * h * 4);
  std::memcpy(n.GetData(MEMORYDEVICE_CPU), normals, (size_t)w * h * 16);   // rejected pixels keep old components
  std::memcpy(sz.GetData(MEMORYDEVICE_CPU), sigmaZ, (size_t)w * h * 4);
  vb.ComputeNormalAndWeights(&n, &sz, &d, Vector4f(intr[0], intr[1], intr[2], intr[3]));
  std::memcpy(normals, n.GetData(MEMORYDEVICE_CPU), (size_t)w * h * 16);
  std::memcpy(sigmaZ, sz.GetData(MEMORYDEVICE_CPU), (size_t)w * h * 4);
  return ITM_OK;
}
int itmr_update_view(const int16_t* raw, int w, int h, int calibType, float c0, float c1, const float intr_d[4], int useBilateralFilter,
                     int modelSensorNoise, float* depth_out, float* scratch, float* normals, float* sigmaZ, itm_stream) {
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(intr_d[0], intr_d[1], intr_d[2], intr_d[3], w, h);
  calib.disparityCalib.params = Vector2f(c0, c1);
  calib.disparityCalib.type = calibType == 0 ? ITMDisparityCalib::TRAFO_KINECT : ITMDisparityCalib::TRAFO_AFFINE;
  ITMViewBuilder_CPU vb(&calib);
  ITMUChar4Image rgb(Vector2i(w, h), true, false);
  ITMShortImage rawImg(Vector2i(w, h), true, false);
  std::memcpy(rawImg.GetData(MEMORYDEVICE_CPU), raw, (size_t)w * h * 2);
  ITMView* view = NULL;
  vb.UpdateView(&view, &rgb, &rawImg, useBilateralFilter != 0, modelSensorNoise != 0);
  std::memcpy(depth_out, view->depth->GetData(MEMORYDEVICE_CPU), (size_t)w * h * 4);
  if (modelSensorNoise) {
    std::memcpy(normals, view->depthNormal->GetData(MEMORYDEVICE_CPU), (size_t)w * h * 16);
    std::memcpy(sigmaZ, view->depthUncertainty->GetData(MEMORYDEVICE_CPU), (size_t)w * h * 4);
  }
  (void)scratch;
  delete view;
  return ITM_OK;
}

int itmr_get_counters(const itm_scene* s, const itm_render_state* r, itm_counters* c, itm_stream) {
  std::memset(c, 0, sizeof *c);
  if (s) { c->lastFreeBlockId = *s->impl->lastFreeBlockId(); c->lastFreeExcessListId = s->impl->lastFreeExcess(); }
  if (r) {
    if (r->hash) c->noVisibleEntries = ((ITMRenderState_VH*)r->rs)->noVisibleEntries;
    c->noFwdProjMissingPoints = r->rs->noFwdProjMissingPoints;
    c->noTotalPoints = r->noTotalPoints;
  }
  return ITM_OK;
}
int itmr_set_counters(itm_scene* s, itm_render_state* r, const itm_counters* c, itm_stream) {
  if (s) { *s->impl->lastFreeBlockId() = c->lastFreeBlockId; s->impl->setLastFreeExcess(c->lastFreeExcessListId); }
  if (r && r->hash) ((ITMRenderState_VH*)r->rs)->noVisibleEntries = c->noVisibleEntries;
  return ITM_OK;
}

static void* buf_of(const itm_scene* s, const itm_render_state* r, int which, size_t* bytes) {
  *bytes = 0;
  if (which <= ITM_BUF_ALLOCATION_LIST || which == ITM_BUF_SWAP_STATES) return s ? s->impl->buffer(which, bytes) : 0;
  if (!r) return 0;
  size_t P = (size_t)r->w * r->h;
  switch (which) {
    case ITM_BUF_VISIBLE_IDS: if (r->hash) { *bytes = (size_t)SDF_LOCAL_BLOCK_NUM * 4; return ((ITMRenderState_VH*)r->rs)->GetVisibleEntryIDs(); } break;
    case ITM_BUF_VISIBLE_TYPE: if (r->hash) { *bytes = ITMVoxelBlockHash::noTotalEntries; return ((ITMRenderState_VH*)r->rs)->GetEntriesVisibleType(); } break;
    case ITM_BUF_RANGE_IMAGE: *bytes = P * 8; return r->rs->renderingRangeImage->GetData(MEMORYDEVICE_CPU);
    case ITM_BUF_RAYCAST_RESULT: *bytes = P * 16; return r->rs->raycastResult->GetData(MEMORYDEVICE_CPU);
    case ITM_BUF_RAYCAST_IMAGE: *bytes = P * 4; return r->rs->raycastImage->GetData(MEMORYDEVICE_CPU);
    case ITM_BUF_FORWARD_PROJECTION: *bytes = P * 16; return r->rs->forwardProjection->GetData(MEMORYDEVICE_CPU);
    case ITM_BUF_MISSING_POINTS: *bytes = P * 4; return r->rs->fwdProjMissingPoints->GetData(MEMORYDEVICE_CPU);
  }
  return 0;
}
size_t itmr_buffer_bytes(const itm_scene* s, const itm_render_state* r, int which) { size_t b; buf_of(s, r, which, &b); return b; }
void* itmr_buffer_ptr(const itm_scene* s, const itm_render_state* r, int which) { size_t b; return buf_of(s, r, which, &b); }
int itmr_download(const itm_scene* s, const itm_render_state* r, int which, void* dst, size_t bytes, itm_stream) {
  size_t b; void* p = buf_of(s, r, which, &b);
  if (!p && bytes == 0) return ITM_OK;
  if (!p || bytes > b) return fail(ITM_ERR_INVALID, "bad buffer / size");
  std::memcpy(dst, p, bytes);
  return ITM_OK;
}
int itmr_upload(itm_scene* s, itm_render_state* r, int which, const void* src, size_t bytes, itm_stream) {
  size_t b; void* p = buf_of(s, r, which, &b);
  if (!p || bytes > b) return fail(ITM_ERR_INVALID, "bad buffer / size");
  std::memcpy(p, src, bytes);
  return ITM_OK;
}
int itmr_export_visible_record(const itm_render_state* r, const float M_d[16], int max_ids, void* dst, itm_stream) {
  if (!r || !r->hash) return fail(ITM_ERR_INVALID, "no visible list");
  std::memcpy(dst, M_d, 64);
  int32_t* d = (int32_t*)dst + 16;
  ITMRenderState_VH* vh = (ITMRenderState_VH*)r->rs;
  d[0] = vh->noVisibleEntries;
  for (int i = 0; i < max_ids; ++i) d[1 + i] = (i < vh->noVisibleEntries) ? vh->GetVisibleEntryIDs()[i] : -1;
  return ITM_OK;
}

}  // extern "C"

// ---- ICP depth tracker: the reference's own ITMDepthTracker_CPU / ITMLowLevelEngine_CPU -----------------
namespace {
// exposes the protected evaluation state of ITMDepthTracker so that ComputeGandH can be called for one level
struct TrackerProbe : ITMDepthTracker_CPU {
  TrackerProbe(Vector2i sz, TrackerIterationType* regime, const ITMLowLevelEngine* ll)
      : ITMDepthTracker_CPU(sz, regime, 1, 0, 0.01f, 1e-3f, ll) {}
  int eval(ITMTemplatedHierarchyLevel<ITMFloatImage>* vl, ITMSceneHierarchyLevel* sl, const Matrix4f& sp, TrackerIterationType it,
           float dth, Matrix4f inv, float& f, float* nabla, float* hessian) {
    this->viewHierarchyLevel = vl; this->sceneHierarchyLevel = sl; this->scenePose = sp; this->iterationType = it;
    this->levelId = 0; this->distThresh[0] = dth;
    return this->ComputeGandH(f, nabla, hessian, inv);
  }
};
}  // namespace

extern "C" {

int itmr_filter_subsample_with_holes(const float* in, int w_in, int h_in, float* out, itm_stream) {
  ITMLowLevelEngine_CPU ll;
  ITMFloatImage src(Vector2i(w_in, h_in), true, false), dst(Vector2i(w_in, h_in), true, false);
  std::memcpy(src.GetData(MEMORYDEVICE_CPU), in, (size_t)w_in * h_in * 4);
  ll.FilterSubsampleWithHoles(&dst, &src);
  std::memcpy(out, dst.GetData(MEMORYDEVICE_CPU), (size_t)(w_in / 2) * (h_in / 2) * 4);
  return ITM_OK;
}

int itmr_tracker_compute_g_and_h(const float* depth, int w, int h, const float vi[4], const float* pts, const float* nrm, int sW, int sH,
                                 const float si[4], const float invPose[16], const float scenePose[16], float distThresh, int type,
                                 itm_tracker_gh* out, itm_stream) {
  std::memset(out, 0, sizeof *out);
  ITMLowLevelEngine_CPU ll;
  TrackerIterationType regime[1] = {(TrackerIterationType)type};
  TrackerProbe probe(Vector2i(w, h), regime, &ll);
  ITMTemplatedHierarchyLevel<ITMFloatImage> vl(Vector2i(w, h), 0, regime[0], MEMORYDEVICE_CPU, false);
  std::memcpy(vl.depth->GetData(MEMORYDEVICE_CPU), depth, (size_t)w * h * 4);
  vl.intrinsics = Vector4f(vi[0], vi[1], vi[2], vi[3]);
  ITMSceneHierarchyLevel sl(Vector2i(sW, sH), 0, regime[0], MEMORYDEVICE_CPU, false);
  std::memcpy(sl.pointsMap->GetData(MEMORYDEVICE_CPU), pts, (size_t)sW * sH * 16);
  std::memcpy(sl.normalsMap->GetData(MEMORYDEVICE_CPU), nrm, (size_t)sW * sH * 16);
  sl.intrinsics = Vector4f(si[0], si[1], si[2], si[3]);
  Matrix4f sp, inv; set_matrix(sp, scenePose); set_matrix(inv, invPose);
  float hessian[36]; for (int i = 0; i < 36; ++i) hessian[i] = 0;
  float nabla[6] = {0, 0, 0, 0, 0, 0}, f = 0;
  int n = probe.eval(&vl, &sl, sp, regime[0], distThresh, inv, f, nabla, hessian);
  out->noValidPoints = n; out->f = f;
  for (int i = 0; i < 6; ++i) out->nabla[i] = nabla[i];
  for (int i = 0; i < 36; ++i) out->hessian[i] = hessian[i];
  return ITM_OK;
}

int itmr_track_camera(const itm_tracker_config* cfg, const itm_view* view, const float* pts, const float* nrm, const float scenePose[16],
                      float M_out[16], itm_stream) {
  const int L = cfg->noHierarchyLevels;
  ITMLowLevelEngine_CPU ll;
  TrackerIterationType regime[8];
  for (int i = 0; i < L; ++i) regime[i] = (TrackerIterationType)cfg->trackingRegime[i];
  Vector2i sz(view->w, view->h);
  ITMDepthTracker_CPU tracker(sz, regime, L, cfg->noICPRunTillLevel, cfg->distThresh, cfg->terminationThreshold, &ll);
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(view->intr_d[0], view->intr_d[1], view->intr_d[2], view->intr_d[3], (float)view->w, (float)view->h);
  ITMView v(&calib, sz, sz, false);
  std::memcpy(v.depth->GetData(MEMORYDEVICE_CPU), view->depth, (size_t)view->w * view->h * 4);
  ITMTrackingState ts(sz, MEMORYDEVICE_CPU);
  std::memcpy(ts.pointCloud->locations->GetData(MEMORYDEVICE_CPU), pts, (size_t)view->w * view->h * 16);
  std::memcpy(ts.pointCloud->colours->GetData(MEMORYDEVICE_CPU), nrm, (size_t)view->w * view->h * 16);
  Matrix4f M, sp; set_matrix(M, view->M_d); set_matrix(sp, scenePose);
  ts.pose_d->SetM(M);
  ts.pose_pointCloud->SetM(sp);
  tracker.TrackCamera(&ts, &v);
  std::memcpy(M_out, ts.pose_d->GetM().m, 64);
  return ITM_OK;
}

}  // extern "C"

// ---- the callers of the path, on the reference's objects --------------------------------------------------------------------------
// ITMMainEngine::ProcessFrame (Engine/ITMMainEngine.cpp:111-127) over a sequence of raw frames, with the reference's view builder,
// ITMTrackingState (TrackerFarFromPointCloud is ITS code, Objects/ITMTrackingState.h:41-59), engines and ITMDepthTracker_CPU.  The three
// functions of the control flow itself -- ITMMainEngine::ProcessFrame, ITMTrackingController::Track / ::Prepare
// (Engine/ITMTrackingController.cpp:11-46) and ITMDenseMapper::ProcessFrame (Engine/ITMDenseMapper.cpp:50-58) -- cannot be compiled
// from the reference here: their translation units include ITMLib.h -> ITMTrackerFactory.h -> <glog/logging.h>, which this image
// lacks (and no stand-in is written).  Their statements are restated below, in their order, around the reference's objects.
// trackerType: 0 colour (external poses, point cloud for the colour tracker), 1 ICP (ITMDepthTracker_CPU), 2 external poses.
// Per frame k: age[k], full[k], pose[16 k ..], digest[4 k ..] = FNV-1a of {points, colours/normals, raycastImage, visible ids}.
namespace {
uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

extern "C" int itmr_debug_main_engine_sequence(itm_scene* s, itm_render_state* r, int w, int h, const float intr[4], int nFrames, const int16_t* raw,
                                               const float* externalPoses, int trackerType, int useApproximateRaycast, int skipPoints,
                                               const uint8_t* fusionActive, const uint8_t* mainProcessingActive, const itm_tracker_config* cfg,
                                               int32_t* age, int32_t* full, float* poses, uint64_t* digest) {
  const Vector2i sz(w, h);
  ITMRGBDCalib& calib = r->calib;
  calib.intrinsics_d.SetFrom(intr[0], intr[1], intr[2], intr[3], (float)w, (float)h);
  calib.intrinsics_rgb.SetFrom(intr[0], intr[1], intr[2], intr[3], (float)w, (float)h);
  calib.disparityCalib.params = Vector2f(0.001f, 0.0f);
  calib.disparityCalib.type = ITMDisparityCalib::TRAFO_AFFINE;
  ITMViewBuilder_CPU viewBuilder(&calib);
  ITMLowLevelEngine_CPU ll;
  ITMDepthTracker_CPU* tracker = NULL;
  if (trackerType == 1) {
    TrackerIterationType regime[8];
    for (int i = 0; i < cfg->noHierarchyLevels; ++i) regime[i] = (TrackerIterationType)cfg->trackingRegime[i];
    tracker = new ITMDepthTracker_CPU(sz, regime, cfg->noHierarchyLevels, cfg->noICPRunTillLevel, cfg->distThresh, cfg->terminationThreshold, &ll);
  }
  ITMTrackingState* ts = r->ts;
  ITMView* view = NULL;
  ITMUChar4Image rgb(sz, true, false);
  for (int i = 0; i < w * h; ++i) rgb.GetData(MEMORYDEVICE_CPU)[i] = Vector4u((unsigned char)(i % w), (unsigned char)(i / w), (unsigned char)((i % w) ^ (i / w)), 255);
  ITMShortImage rawImg(sz, true, false);
  for (int k = 0; k < nFrames; ++k) {
    std::memcpy(rawImg.GetData(MEMORYDEVICE_CPU), raw + (size_t)k * w * h, (size_t)w * h * 2);
    if (externalPoses) { Matrix4f M; set_matrix(M, externalPoses + 16 * k); ts->pose_d->SetM(M); }     // the pose source of this fork writes it before the frame
    // ---- ITMMainEngine::ProcessFrame
    viewBuilder.UpdateView(&view, &rgb, &rawImg, false, false);
    if (mainProcessingActive[k]) {
      // ---- ITMTrackingController::Track
      if (ts->age_pointCloud != -1 && tracker) tracker->TrackCamera(ts, view);
      ts->requiresFullRendering = ts->TrackerFarFromPointCloud() || !useApproximateRaycast;
      // ---- ITMDenseMapper::ProcessFrame (no swapping)
      if (fusionActive[k]) {
        s->impl->allocate(view, ts, r->rs, false);
        s->impl->integrate(view, ts, r->rs);
      }
      // ---- ITMTrackingController::Prepare
      if (trackerType == 0) {
        ITMPose pose_rgb(view->calib->trafo_rgb_to_depth.calib_inv * ts->pose_d->GetM());
        s->impl->expectedDepths(&pose_rgb, &(view->calib->intrinsics_rgb), r->rs);
        s->impl->pointCloud(view, ts, r->rs, skipPoints != 0);
        ts->age_pointCloud = 0;
      } else {
        s->impl->expectedDepths(ts->pose_d, &(view->calib->intrinsics_d), r->rs);
        if (ts->requiresFullRendering) {
          s->impl->icpMaps(view, ts, r->rs);
          ts->pose_pointCloud->SetFrom(ts->pose_d);
          if (ts->age_pointCloud == -1) ts->age_pointCloud = -2;
          else ts->age_pointCloud = 0;
        } else {
          s->impl->forwardRender(view, ts, r->rs);
          ts->age_pointCloud++;
        }
      }
    }
    age[k] = ts->age_pointCloud; full[k] = ts->requiresFullRendering ? 1 : 0;
    std::memcpy(poses + 16 * k, ts->pose_d->GetM().m, 64);
    const size_t nPts = trackerType == 0 ? (size_t)ts->pointCloud->noTotalPoints : (size_t)w * h;
    digest[4 * k + 0] = fnv(ts->pointCloud->locations->GetData(MEMORYDEVICE_CPU), nPts * 16);
    digest[4 * k + 1] = fnv(ts->pointCloud->colours->GetData(MEMORYDEVICE_CPU), nPts * 16);
    digest[4 * k + 2] = fnv(r->rs->raycastImage->GetData(MEMORYDEVICE_CPU), (size_t)w * h * 4);
    uint64_t dv = nPts;
    if (r->hash) {
      ITMRenderState_VH* vh = (ITMRenderState_VH*)r->rs;
      dv = fnv(vh->GetVisibleEntryIDs(), (size_t)vh->noVisibleEntries * 4, fnv(&vh->noVisibleEntries, 4));
    }
    digest[4 * k + 3] = dv;
  }
  delete view;
  delete tracker;
  return ITM_OK;
}
