// ref_hip_demo.cpp -- drives the REFERENCE's own abstract engine interfaces twice, side by side: once with the
// reference's CPU engines, once with the HIP adapters of ITMEngines_HIP.h (libitmhip.so behind them), on the same
// synthetic frames, and compares everything the rest of InfiniTAM would read -- bit for bit.
//
// Every configuration runs under BOTH mirror policies of the adapter (ITMEngines_HIP.h): HIP_MIRROR_EAGER (results back on the host
// after every call) and HIP_MIRROR_ON_DEMAND (results stay in HBM until asked for) -- and checks, with the library's per-kernel
// timers, that under ON_DEMAND the reference's four per-frame calls really ran as the fused frame (no separate expected-depth
// launches), under EAGER they did not.
//
//   ref_hip_demo             parity at 160 x 120 (hash, colour hash, dense), view builder, tracker
//   ref_hip_demo --bench N   BASELINE configs[1] through the reference's classes: 640 x 480, ITMVoxel_s, hash, 4 mm, bench trajectory;
//                            parity over 5 frames at that size, then N timed frames per policy in the reference's call order
//                            (ITMDenseMapper.cpp:50-57 + ITMTrackingController.cpp:30-46), the depth image uploaded from the
//                            reference's host ITMView every frame
//
// Built only where the reference tree exists (oracle/Makefile target `hipdemo` -> oracle/_ref/ref_hip_demo, g++ only;
// the binary travels to the GPU box).  Prints one JSON line per configuration; exit code 0 iff all are equal.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ITMLib/Engine/DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMMeshingEngine_CPU.cpp"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMDepthTracker_CPU.h"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMLowLevelEngine_CPU.h"
#include "ITMLib/Engine/DeviceSpecific/CPU/ITMViewBuilder_CPU.h"
#include "ITMEngines_HIP.h"
#include "ITMTrackers_HIP.h"

using namespace ITMLib::Engine;
using namespace ITMLib::Objects;

template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_f_rgb, ITMVoxelBlockHash>;
template class ITMLib::Engine::ITMMeshingEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;
template class ITMLib::Engine::ITMVisualisationEngine_CPU<ITMVoxel_s, ITMPlainVoxelArray>;

static int W = 160, H = 120;      // (--bench switches to 640 x 480)

// sphere (r 0.5 m at z 1.5 m) in front of a wall at 2.5 m, camera at t, fp32 without contraction
static void make_depth(float* d, float tx, float ty) {
  const float fx = 145.0f * (float)W / 160.0f, fy = fx, cx = 0.5f * (float)W, cy = 0.5f * (float)H;
  for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
    const float dx = ((float)x - cx) / fx, dy = ((float)y - cy) / fy;
    const float ox = tx, oy = ty, oz = 0.0f - 1.5f;
    const float A = dx * dx + dy * dy + 1.0f, B = 2.0f * (ox * dx + oy * dy + oz), C = ox * ox + oy * oy + oz * oz - 0.25f;
    const float disc = B * B - 4.0f * A * C;
    float z = 2.5f;
    if (disc > 0) { const float t = (-B - std::sqrt(disc)) / (2.0f * A); if (t > 0) z = t; }
    d[x + y * W] = z;
  }
}

template <class T> static bool same(const T* a, const T* b, size_t n) { return std::memcmp(a, b, n * sizeof(T)) == 0; }

template <class TIndex> struct VisibleCmp;
template <> struct VisibleCmp<ITMVoxelBlockHash> {
  static bool Equal(const ITMRenderState* a, const ITMRenderState* b, std::string& why) {
    const ITMRenderState_VH* x = (const ITMRenderState_VH*)a; const ITMRenderState_VH* y = (const ITMRenderState_VH*)b;
    if (x->noVisibleEntries != y->noVisibleEntries) { why = "noVisibleEntries"; return false; }
    if (!same(x->GetVisibleEntryIDs(), y->GetVisibleEntryIDs(), (size_t)x->noVisibleEntries)) { why = "visibleEntryIDs"; return false; }
    if (!same(const_cast<ITMRenderState_VH*>(x)->GetEntriesVisibleType(), const_cast<ITMRenderState_VH*>(y)->GetEntriesVisibleType(), (size_t)ITMVoxelBlockHash::noTotalEntries)) { why = "entriesVisibleType"; return false; }
    return true;
  }
  static bool SceneEqual(ITMVoxelBlockHash& a, ITMVoxelBlockHash& b, std::string& why) {
    if (!same(a.GetEntries(), b.GetEntries(), (size_t)ITMVoxelBlockHash::noTotalEntries)) { why = "hash entries"; return false; }
    if (!same(a.GetExcessAllocationList(), b.GetExcessAllocationList(), (size_t)SDF_EXCESS_LIST_SIZE)) { why = "excess list"; return false; }
    if (a.GetLastFreeExcessListId() != b.GetLastFreeExcessListId()) { why = "lastFreeExcessListId"; return false; }
    return true;
  }
  static size_t Voxels(ITMVoxelBlockHash&) { return (size_t)SDF_LOCAL_BLOCK_NUM * SDF_BLOCK_SIZE3; }
  static void Configure(ITMVoxelBlockHash&) {}
};
template <> struct VisibleCmp<ITMPlainVoxelArray> {
  static bool Equal(const ITMRenderState*, const ITMRenderState*, std::string&) { return true; }
  static bool SceneEqual(ITMPlainVoxelArray&, ITMPlainVoxelArray&, std::string&) { return true; }
  static size_t Voxels(ITMPlainVoxelArray& i) { Vector3i s = i.getVolumeSize(); return (size_t)s.x * s.y * s.z; }
  static void Configure(ITMPlainVoxelArray& i) {   // a 128^3 window of the 512^3 allocation keeps the CPU side quick
    ITMPlainVoxelArray::IndexData* d = const_cast<ITMPlainVoxelArray::IndexData*>(i.getIndexData());
    d->size = Vector3i(128, 128, 128); d->offset = Vector3i(-64, -64, 100);
  }
};

static void set_calib(ITMRGBDCalib& calib) {
  const float f = 145.0f * (float)W / 160.0f;
  calib.intrinsics_d.SetFrom(f, f, 0.5f * (float)W, 0.5f * (float)H, W, H); calib.intrinsics_rgb.SetFrom(f, f, 0.5f * (float)W, 0.5f * (float)H, W, H);
}

template <class TVoxel, class TIndex>
static bool run(const char* name, float voxelSize, int frames, HipMirrorPolicy policy) {
  HipSetMirrorPolicy(policy);
  const bool onDemand = policy == HIP_MIRROR_ON_DEMAND;
  ITMSceneParams sp(0.02f, 100, voxelSize, 0.2f, 3.0f, false);
  ITMScene<TVoxel, TIndex> sceneA(&sp, false, MEMORYDEVICE_CPU), sceneB(&sp, false, MEMORYDEVICE_CPU);
  VisibleCmp<TIndex>::Configure(sceneA.index); VisibleCmp<TIndex>::Configure(sceneB.index);
  // everything below goes through the reference's abstract interfaces
  ITMSceneReconstructionEngine<TVoxel, TIndex>* reco[2] = {new ITMSceneReconstructionEngine_CPU<TVoxel, TIndex>(), new ITMSceneReconstructionEngine_HIP<TVoxel, TIndex>()};
  ITMVisualisationEngine<TVoxel, TIndex>* vis[2] = {new ITMVisualisationEngine_CPU<TVoxel, TIndex>(&sceneA), new ITMVisualisationEngine_HIP<TVoxel, TIndex>(&sceneB)};
  ITMScene<TVoxel, TIndex>* scene[2] = {&sceneA, &sceneB};
  ITMRGBDCalib calib;
  set_calib(calib);
  ITMRenderState* rs[2]; ITMView* view[2]; ITMTrackingState* ts[2];
  for (int e = 0; e < 2; ++e) {
    reco[e]->ResetScene(scene[e]);
    rs[e] = vis[e]->CreateRenderState(Vector2i(W, H));
    view[e] = new ITMView(&calib, Vector2i(W, H), Vector2i(W, H), false);
    ts[e] = new ITMTrackingState(Vector2i(W, H), MEMORYDEVICE_CPU);
  }
  // the library's per-kernel timers say which launches the calls became: range = the separate CreateExpectedDepths launches
  itm_scene* dev = HipSceneOf(&sceneB);
  HipCheck(itm_profile_enable(dev, (1u << ITM_TK_RANGE) | (1u << ITM_TK_RAYCAST)), "profile_enable");
  std::string why; bool ok = true;
  ITMUChar4Image img[2] = {ITMUChar4Image(Vector2i(W, H), true, false), ITMUChar4Image(Vector2i(W, H), true, false)};
  for (int k = 0; k < frames && ok; ++k) {
    const float tx = 0.01f * (float)k;
    Matrix4f M; M.setIdentity(); M.m[12] = -tx;
    for (int e = 0; e < 2; ++e) {
      make_depth(view[e]->depth->GetData(MEMORYDEVICE_CPU), tx, 0.0f);
      Vector4u* c = view[e]->rgb->GetData(MEMORYDEVICE_CPU);
      for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) c[x + y * W] = Vector4u((uchar)(x & 255), (uchar)(y & 255), (uchar)((x ^ y) & 255), 255);
      ts[e]->pose_d->SetM(M);
      if (e == 1) HipMarkViewUpdated(view[e]);      // what ITMViewBuilder_HIP::UpdateView does: the view has new content
      // ITMDenseMapper::ProcessFrame + ITMTrackingController::Prepare, then a free-view render
      reco[e]->AllocateSceneFromDepth(scene[e], view[e], ts[e], rs[e]);
      reco[e]->IntegrateIntoScene(scene[e], view[e], ts[e], rs[e]);
      vis[e]->CreateExpectedDepths(ts[e]->pose_d, &view[e]->calib->intrinsics_d, rs[e]);
      vis[e]->CreateICPMaps(view[e], ts[e], rs[e]);
      vis[e]->RenderImage(ts[e]->pose_d, &view[e]->calib->intrinsics_d, rs[e], &img[e], IITMVisualisationEngine::RENDER_COLOUR_FROM_NORMAL);
    }
    if (onDemand) { HipSyncRenderStateToHost<TIndex>(rs[1]); HipSyncTrackingStateToHost(ts[1]); }      // what a host does before it reads
    const size_t px = (size_t)W * H;
    if (!VisibleCmp<TIndex>::Equal(rs[0], rs[1], why)) ok = false;
    // the range image is only defined on the sub-sampled region the rays read
    for (int y = 0; y < (H + 7) / 8 && ok; ++y)
      if (!same(rs[0]->renderingRangeImage->GetData(MEMORYDEVICE_CPU) + y * W, rs[1]->renderingRangeImage->GetData(MEMORYDEVICE_CPU) + y * W, (size_t)((W + 7) / 8))) { ok = false; why = "range image"; }
    if (ok && !same(ts[0]->pointCloud->locations->GetData(MEMORYDEVICE_CPU), ts[1]->pointCloud->locations->GetData(MEMORYDEVICE_CPU), px)) { ok = false; why = "ICP points"; }
    if (ok && !same(ts[0]->pointCloud->colours->GetData(MEMORYDEVICE_CPU), ts[1]->pointCloud->colours->GetData(MEMORYDEVICE_CPU), px)) { ok = false; why = "ICP normals"; }
    if (ok && !same(rs[0]->raycastImage->GetData(MEMORYDEVICE_CPU), rs[1]->raycastImage->GetData(MEMORYDEVICE_CPU), px)) { ok = false; why = "raycast image"; }
    if (ok && !same(img[0].GetData(MEMORYDEVICE_CPU), img[1].GetData(MEMORYDEVICE_CPU), px)) { ok = false; why = "free-view render"; }
    if (ok && !same(ts[0]->pose_pointCloud->GetM().m, ts[1]->pose_pointCloud->GetM().m, 16)) { ok = false; why = "pose_pointCloud"; }
    if (!ok) why += " (frame " + std::to_string(k) + ")";
  }
  long long hits = 0;
  itm_profile prof; std::memset(&prof, 0, sizeof prof);
  HipCheck(itm_profile_read(dev, &prof, 1), "profile_read");
  HipCheck(itm_profile_enable(dev, 0), "profile_enable");
  if (ok && HipIndexTraits<TIndex>::value == ITM_INDEX_HASH) {
    // ON_DEMAND: the four calls of every frame formed the fused frame (the free-view render adds one ray cast per frame, no range launch
    // of the tracked view); EAGER: the downloads between the calls launched them one by one
    const int rangeLaunches = prof.calls[ITM_TK_RANGE];
    if (onDemand && rangeLaunches != 0) { ok = false; why = "ON_DEMAND: " + std::to_string(rangeLaunches) + " separate expected-depth launches -- the frame was not fused"; }
    if (!onDemand && rangeLaunches == 0) { ok = false; why = "EAGER: no separate expected-depth launch although every call was mirrored"; }
  }
  if (ok) {
    ITMSceneReconstructionEngine_HIP<TVoxel, TIndex>::SyncSceneToHost(&sceneB);
    if (!VisibleCmp<TIndex>::SceneEqual(sceneA.index, sceneB.index, why)) ok = false;
    else if (!same(sceneA.localVBA.GetVoxelBlocks(), sceneB.localVBA.GetVoxelBlocks(), VisibleCmp<TIndex>::Voxels(sceneA.index))) { ok = false; why = "voxel blocks"; }
    else if (sceneA.localVBA.lastFreeBlockId != sceneB.localVBA.lastFreeBlockId) { ok = false; why = "lastFreeBlockId"; }
    const Vector4f* p = ts[1]->pointCloud->locations->GetData(MEMORYDEVICE_CPU);
    for (int i = 0; i < W * H; ++i) hits += p[i].w > 0;
  }
  long long triangles = -1;
  if (ok) {
    // ITMMainEngine::UpdateMesh: the reference's meshing engine on its scene vs the HIP one on the device twin, through the base class
    ITMMeshingEngine<TVoxel, TIndex>* mesher[2] = {new ITMMeshingEngine_CPU<TVoxel, TIndex>(), new ITMMeshingEngine_HIP<TVoxel, TIndex>()};
    ITMMesh* mesh[2] = {new ITMMesh(MEMORYDEVICE_CPU), new ITMMesh(MEMORYDEVICE_CPU)};
    for (int e = 0; e < 2; ++e) mesher[e]->MeshScene(mesh[e], scene[e]);
    triangles = mesh[1]->noTotalTriangles;
    if (mesh[0]->noTotalTriangles != mesh[1]->noTotalTriangles) { ok = false; why = "noTotalTriangles"; }
    else if (!same(mesh[0]->triangles->GetData(MEMORYDEVICE_CPU), mesh[1]->triangles->GetData(MEMORYDEVICE_CPU), (size_t)mesh[0]->noTotalTriangles)) { ok = false; why = "mesh triangles"; }
    for (int e = 0; e < 2; ++e) { delete mesh[e]; delete mesher[e]; }
  }
  std::printf("{\"config\": \"%s\", \"policy\": \"%s\", \"frames\": %d, \"equal\": %s, \"icp_points\": %lld, \"lastFreeBlockId\": %d, \"triangles\": %lld, "
              "\"range_launches\": %d, \"raycast_launches\": %d, \"mismatch\": \"%s\"}\n", name, onDemand ? "on_demand" : "eager", frames,
              ok ? "true" : "false", hits, sceneB.localVBA.lastFreeBlockId, triangles, (int)prof.calls[ITM_TK_RANGE], (int)prof.calls[ITM_TK_RAYCAST], why.c_str());
  // teardown in the reference's order: render states, then engines (the HIP visualisation engine releases the scene twin)
  for (int e = 0; e < 2; ++e) { HipReleaseView(view[e]); HipReleaseTrackingState(ts[e]); delete rs[e]; delete view[e]; delete ts[e]; delete reco[e]; delete vis[e]; }
  HipSetMirrorPolicy(HIP_MIRROR_EAGER);
  return ok;
}

// ---- --bench: BASELINE configs[1] through the reference's own classes --------------------------------------------------------------
static int tri(int k) { return std::abs(((k + 25) % 100) - 50) - 25; }      // the bench trajectory: bounded triangle waves, period 100

// `frames` timed frames (after 20 of warm-up) of: view has new content -> AllocateSceneFromDepth -> IntegrateIntoScene (ITMDenseMapper::
// ProcessFrame) -> CreateExpectedDepths -> CreateICPMaps (ITMTrackingController::Prepare), every call through the reference's base-class
// pointers; the 100 distinct frames of the trajectory live in 100 host ITMViews and are uploaded again every time they come round
static double time_frames(ITMScene<ITMVoxel_s, ITMVoxelBlockHash>* scene, ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco,
                          ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis, std::vector<ITMView*>& views, ITMTrackingState* ts, ITMRenderState* rs,
                          int frames, bool hip) {
  reco->ResetScene(scene);
  const int warm = hip ? (int)views.size() : 1;      // (every view once: its device stage and staging ring are created on first use)
  std::chrono::steady_clock::time_point t0;
  for (int k = 0; k < warm + frames; ++k) {
    if (k == warm) { if (hip) HipCheck(itm_stream_synchronize(0), "sync"); t0 = std::chrono::steady_clock::now(); }
    ITMView* v = views[(size_t)k % views.size()];
    Matrix4f M; M.setIdentity(); M.m[12] = -(0.004f * (float)tri(k)); M.m[13] = -(0.002f * (float)tri(2 * k));
    ts->pose_d->SetM(M);
    if (hip) HipMarkViewUpdated(v);
    reco->AllocateSceneFromDepth(scene, v, ts, rs);
    reco->IntegrateIntoScene(scene, v, ts, rs);
    vis->CreateExpectedDepths(ts->pose_d, &v->calib->intrinsics_d, rs);
    vis->CreateICPMaps(v, ts, rs);
  }
  if (hip) HipCheck(itm_stream_synchronize(0), "sync");
  return (double)frames / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static bool run_bench(int frames) {
  W = 640; H = 480;
  bool ok = true;
  ok &= run<ITMVoxel_s, ITMVoxelBlockHash>("BASELINE configs[1] size: hash ITMVoxel_s 4 mm 640x480", 0.004f, 5, HIP_MIRROR_EAGER);
  ok &= run<ITMVoxel_s, ITMVoxelBlockHash>("BASELINE configs[1] size: hash ITMVoxel_s 4 mm 640x480", 0.004f, 5, HIP_MIRROR_ON_DEMAND);
  ITMSceneParams sp(0.02f, 100, 0.004f, 0.35f, 3.0f, false);
  ITMRGBDCalib calib; set_calib(calib);
  std::vector<ITMView*> views;
  for (int k = 0; k < 100; ++k) {
    views.push_back(new ITMView(&calib, Vector2i(W, H), Vector2i(W, H), false));
    make_depth(views.back()->depth->GetData(MEMORYDEVICE_CPU), 0.004f * (float)tri(k), 0.002f * (float)tri(2 * k));
  }
  double fps[4] = {0, 0, 0, 0};
  {
    // the same frames as the sensor delivers them: raw 16-bit depth through the reference's ITMViewBuilder interface (UpdateView: 614 KB
    // upload + conversion in HBM, straight into the view's device stage), then the four calls -- ITMMainEngine::ProcessFrame without the tracker
    ITMScene<ITMVoxel_s, ITMVoxelBlockHash> scene(&sp, false, MEMORYDEVICE_CPU);
    ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco = new ITMSceneReconstructionEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>();
    ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis = new ITMVisualisationEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>(&scene);
    ITMRenderState* rs = vis->CreateRenderState(Vector2i(W, H));
    ITMTrackingState ts(Vector2i(W, H), MEMORYDEVICE_CPU);
    calib.disparityCalib.type = ITMDisparityCalib::TRAFO_AFFINE; calib.disparityCalib.params = Vector2f(0.001f, 0.0f);
    ITMViewBuilder* vb = new ITMViewBuilder_HIP(&calib);
    std::vector<ITMShortImage*> raws;
    for (int k = 0; k < 100; ++k) {
      raws.push_back(new ITMShortImage(Vector2i(W, H), true, false));
      const float* d = views[(size_t)k]->depth->GetData(MEMORYDEVICE_CPU);
      short* r = raws.back()->GetData(MEMORYDEVICE_CPU);
      for (int i = 0; i < W * H; ++i) r[i] = (short)(d[i] * 1000.0f);
    }
    ITMUChar4Image rgb(Vector2i(W, H), true, false);
    ITMShortImage raw(Vector2i(W, H), true, false);
    ITMView* view = NULL;
    HipSetMirrorPolicy(HIP_MIRROR_ON_DEMAND);
    reco->ResetScene(&scene);
    std::chrono::steady_clock::time_point t0;
    for (int k = 0; k < 20 + frames; ++k) {
      if (k == 20) { HipCheck(itm_stream_synchronize(0), "sync"); t0 = std::chrono::steady_clock::now(); }
      Matrix4f M; M.setIdentity(); M.m[12] = -(0.004f * (float)tri(k)); M.m[13] = -(0.002f * (float)tri(2 * k));
      ts.pose_d->SetM(M);
      // (an image source reuses ONE raw image: the frame is copied into it, 614 KB of host memcpy that is part of the figure)
      std::memcpy(raw.GetData(MEMORYDEVICE_CPU), raws[(size_t)k % raws.size()]->GetData(MEMORYDEVICE_CPU), (size_t)W * H * sizeof(short));
      vb->UpdateView(&view, &rgb, &raw, false, false);
      reco->AllocateSceneFromDepth(&scene, view, &ts, rs);
      reco->IntegrateIntoScene(&scene, view, &ts, rs);
      vis->CreateExpectedDepths(ts.pose_d, &view->calib->intrinsics_d, rs);
      vis->CreateICPMaps(view, &ts, rs);
    }
    HipCheck(itm_stream_synchronize(0), "sync");
    fps[3] = (double)frames / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HipSyncRenderStateToHost<ITMVoxelBlockHash>(rs); HipSyncTrackingStateToHost(&ts);
    HipSetMirrorPolicy(HIP_MIRROR_EAGER);
    HipReleaseTrackingState(&ts); HipReleaseView(view);
    for (ITMShortImage* r : raws) delete r;
    delete view; delete vb; delete rs; delete reco; delete vis;
  }
  {
    ITMScene<ITMVoxel_s, ITMVoxelBlockHash> scene(&sp, false, MEMORYDEVICE_CPU);
    ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco = new ITMSceneReconstructionEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>();
    ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis = new ITMVisualisationEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>(&scene);
    ITMRenderState* rs = vis->CreateRenderState(Vector2i(W, H));
    ITMTrackingState ts(Vector2i(W, H), MEMORYDEVICE_CPU);
    for (int pol = 0; pol < 2; ++pol) {
      HipSetMirrorPolicy(pol ? HIP_MIRROR_ON_DEMAND : HIP_MIRROR_EAGER);
      fps[pol] = time_frames(&scene, reco, vis, views, &ts, rs, pol ? frames : (frames < 100 ? frames : 100), true);
    }
    HipSyncRenderStateToHost<ITMVoxelBlockHash>(rs); HipSyncTrackingStateToHost(&ts);
    HipSetMirrorPolicy(HIP_MIRROR_EAGER);
    HipReleaseTrackingState(&ts);
    delete rs; delete reco; delete vis;
  }
  {
    ITMScene<ITMVoxel_s, ITMVoxelBlockHash> scene(&sp, false, MEMORYDEVICE_CPU);
    ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco = new ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>();
    ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis = new ITMVisualisationEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>(&scene);
    ITMRenderState* rs = vis->CreateRenderState(Vector2i(W, H));
    ITMTrackingState ts(Vector2i(W, H), MEMORYDEVICE_CPU);
    fps[2] = time_frames(&scene, reco, vis, views, &ts, rs, 10, false);
    delete rs; delete reco; delete vis;
  }
  for (ITMView* v : views) { HipReleaseView(v); delete v; }
  std::printf("{\"bench\": \"BASELINE configs[1] through the reference's abstract engines (ITMDenseMapper::ProcessFrame + ITMTrackingController::Prepare call order), "
              "640x480 hash ITMVoxel_s 4 mm, pool SDF_LOCAL_BLOCK_NUM = %d blocks, one 1.2 MB float depth image uploaded from the reference's host ITMView per frame\", "
              "\"frames\": %d, \"fps_on_demand\": %.1f, \"fps_eager\": %.1f, \"fps_on_demand_raw_depth_through_ITMViewBuilder_HIP\": %.1f, \"fps_reference_cpu_engines_1_thread\": %.2f, "
              "\"equal\": %s, \"icp_points\": 100000, \"mismatch\": \"\"}\n",
              (int)SDF_LOCAL_BLOCK_NUM, frames, fps[1], fps[0], fps[3], fps[2], ok ? "true" : "false");
  return ok;
}

// view builder: the reference's UpdateView on a raw frame, CPU implementation vs HIP adapter (exp / acos may differ in
// the last place between the device library and the host libm: relative tolerance 2e-6 on the filtered depth)
static bool run_view_builder() {
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(145, 145, 80, 60, W, H); calib.intrinsics_rgb.SetFrom(145, 145, 80, 60, W, H);
  calib.disparityCalib.type = ITMDisparityCalib::TRAFO_AFFINE; calib.disparityCalib.params = Vector2f(0.001f, 0.0f);
  ITMUChar4Image rgb(Vector2i(W, H), true, false);
  ITMShortImage raw(Vector2i(W, H), true, false);
  std::vector<float> d((size_t)W * H); make_depth(d.data(), 0.01f, 0.0f);
  for (int i = 0; i < W * H; ++i) raw.GetData(MEMORYDEVICE_CPU)[i] = (i % 37 == 0) ? 0 : (short)(d[i] * 1000.0f + 0.5f);
  ITMViewBuilder* vb[2] = {new ITMViewBuilder_CPU(&calib), new ITMViewBuilder_HIP(&calib)};
  ITMView* view[2] = {NULL, NULL};
  for (int e = 0; e < 2; ++e) vb[e]->UpdateView(&view[e], &rgb, &raw, true, true);
  const float* a = view[0]->depth->GetData(MEMORYDEVICE_CPU); const float* b = view[1]->depth->GetData(MEMORYDEVICE_CPU);
  double maxRel = 0; bool holes = true; long valid = 0;
  for (int i = 0; i < W * H; ++i) {
    if ((a[i] <= 0) != (b[i] <= 0)) holes = false;
    if (a[i] > 0) { ++valid; const double r = std::fabs((double)a[i] - b[i]) / a[i]; if (r > maxRel) maxRel = r; }
  }
  const Vector4f* na = view[0]->depthNormal->GetData(MEMORYDEVICE_CPU); const Vector4f* nb = view[1]->depthNormal->GetData(MEMORYDEVICE_CPU);
  long sameW = 0; double maxN = 0;
  for (int i = 0; i < W * H; ++i) {
    if (na[i].w == nb[i].w) ++sameW;
    if (na[i].w == 1.0f && nb[i].w == 1.0f) maxN = std::fmax(maxN, std::fmax(std::fabs(na[i].x - nb[i].x), std::fmax(std::fabs(na[i].y - nb[i].y), std::fabs(na[i].z - nb[i].z))));
  }
  const bool ok = holes && maxRel <= 2e-6 && sameW >= (long)(0.999 * W * H) && maxN < 5e-3;
  std::printf("{\"config\": \"view builder (affine, bilateral, noise model)\", \"equal\": %s, \"valid\": %ld, \"max_rel_depth\": %.3g, \"normal_flags_equal\": %ld, \"max_normal_diff\": %.3g, \"icp_points\": %ld, \"mismatch\": \"\"}\n",
              ok ? "true" : "false", valid, maxRel, sameW, maxN, valid);
  for (int e = 0; e < 2; ++e) { delete view[e]; delete vb[e]; }
  return ok;
}

// tracker: three frames of mapping with the reference CPU engines, then the reference's own TrackCamera (hierarchy,
// LM loop, pose algebra) twice from the same start pose -- ComputeGandH on the CPU vs through the HIP adapter
static bool run_tracker() {
  const int TW = 640, TH = 480;   // the default 5-level hierarchy needs VGA
  ITMSceneParams sp(0.02f, 100, 0.01f, 0.2f, 3.0f, false);
  ITMScene<ITMVoxel_s, ITMVoxelBlockHash> scene(&sp, false, MEMORYDEVICE_CPU);
  ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash> reco; ITMVisualisationEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash> vis(&scene);
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(580, 580, 320, 240, TW, TH); calib.intrinsics_rgb.SetFrom(580, 580, 320, 240, TW, TH);
  reco.ResetScene(&scene);
  ITMRenderState* rs = vis.CreateRenderState(Vector2i(TW, TH));
  ITMView view(&calib, Vector2i(TW, TH), Vector2i(TW, TH), false);
  ITMTrackingState ts(Vector2i(TW, TH), MEMORYDEVICE_CPU);
  auto depth_at = [&](float tx) {
    float* d = view.depth->GetData(MEMORYDEVICE_CPU);
    for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x) {
      const float dx = ((float)x - 320.0f) / 580.0f, dy = ((float)y - 240.0f) / 580.0f, oz = -1.5f;
      const float A = dx * dx + dy * dy + 1.0f, B = 2.0f * (tx * dx + oz), C = tx * tx + oz * oz - 0.25f, disc = B * B - 4.0f * A * C;
      float z = 2.5f; if (disc > 0) { const float t = (-B - std::sqrt(disc)) / (2.0f * A); if (t > 0) z = t; }
      d[x + y * TW] = z;
    }
  };
  Matrix4f M; M.setIdentity();
  for (int k = 0; k < 3; ++k) {
    depth_at(0.01f * k); M.m[12] = -0.01f * k; ts.pose_d->SetM(M);
    reco.AllocateSceneFromDepth(&scene, &view, &ts, rs); reco.IntegrateIntoScene(&scene, &view, &ts, rs);
    vis.CreateExpectedDepths(ts.pose_d, &calib.intrinsics_d, rs); vis.CreateICPMaps(&view, &ts, rs);
  }
  depth_at(0.03f);                     // next frame, tracked from the previous pose
  TrackerIterationType regime[5] = {TRACKER_ITERATION_BOTH, TRACKER_ITERATION_BOTH, TRACKER_ITERATION_ROTATION, TRACKER_ITERATION_ROTATION, TRACKER_ITERATION_ROTATION};
  ITMLowLevelEngine_CPU low;
  ITMTracker* trk[2] = {new ITMDepthTracker_CPU(Vector2i(TW, TH), regime, 5, 0, 0.1f * 0.1f, 1e-3f, &low),
                        new ITMDepthTracker_HIP(Vector2i(TW, TH), regime, 5, 0, 0.1f * 0.1f, 1e-3f, &low)};
  Matrix4f out[2];
  for (int e = 0; e < 2; ++e) { ts.pose_d->SetM(M); trk[e]->TrackCamera(&ts, &view); out[e] = ts.pose_d->GetM(); delete trk[e]; }
  double maxDiff = 0; for (int i = 0; i < 16; ++i) maxDiff = std::fmax(maxDiff, std::fabs((double)out[0].m[i] - out[1].m[i]));
  const bool moved = std::fabs(out[1].m[12] - (-0.03f)) < 2e-3f;     // the tracker found the 1 cm step
  const bool ok = maxDiff < 2e-5 && moved;
  std::printf("{\"config\": \"reference TrackCamera with ComputeGandH on the GPU\", \"equal\": %s, \"max_pose_diff\": %.3g, \"tx\": %.6f, \"icp_points\": 100000, \"mismatch\": \"\"}\n",
              ok ? "true" : "false", maxDiff, out[1].m[12]);
  delete rs;
  return ok;
}

// The whole stack the way a HIP build of InfiniTAM would run it, under HIP_MIRROR_ON_DEMAND, against the reference's CPU stack: per frame
// UpdateView (raw 16-bit depth) -> TrackCamera against the previous frame's ICP maps -> AllocateSceneFromDepth -> IntegrateIntoScene ->
// CreateExpectedDepths -> CreateICPMaps, all through the reference's base classes (ITMMainEngine::ProcessFrame with TRACKER_ICP,
// Engine/ITMMainEngine.cpp:111-127).  On the HIP side nothing is mirrored: the view builder leaves the float depth in the view's device stage
// (the reference's TrackCamera builds its pyramid on the host: ITMDepthTracker_HIP brings the image back first), the tracker takes the ICP
// maps straight from HBM (the registry's copy written by CreateICPMaps), the four engine calls form the fused frame.  Poses must agree
// with the CPU stack within the tracker's tolerance frame after frame; at the end the maps are synchronised and compared where both hit.
static bool run_closed_loop(int frames) {
  const int TW = 640, TH = 480;
  ITMSceneParams sp(0.02f, 100, 0.01f, 0.2f, 3.0f, false);
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(580, 580, 320, 240, TW, TH); calib.intrinsics_rgb.SetFrom(580, 580, 320, 240, TW, TH);
  calib.disparityCalib.type = ITMDisparityCalib::TRAFO_AFFINE; calib.disparityCalib.params = Vector2f(0.001f, 0.0f);
  ITMScene<ITMVoxel_s, ITMVoxelBlockHash> sceneA(&sp, false, MEMORYDEVICE_CPU), sceneB(&sp, false, MEMORYDEVICE_CPU);
  ITMScene<ITMVoxel_s, ITMVoxelBlockHash>* scene[2] = {&sceneA, &sceneB};
  ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco[2] = {new ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>(), new ITMSceneReconstructionEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>()};
  ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis[2] = {new ITMVisualisationEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>(&sceneA), new ITMVisualisationEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>(&sceneB)};
  ITMViewBuilder* vb[2] = {new ITMViewBuilder_CPU(&calib), new ITMViewBuilder_HIP(&calib)};
  TrackerIterationType regime[5] = {TRACKER_ITERATION_BOTH, TRACKER_ITERATION_BOTH, TRACKER_ITERATION_ROTATION, TRACKER_ITERATION_ROTATION, TRACKER_ITERATION_ROTATION};
  ITMLowLevelEngine_CPU low;
  ITMTracker* trk[2] = {new ITMDepthTracker_CPU(Vector2i(TW, TH), regime, 5, 0, 0.1f * 0.1f, 1e-3f, &low),
                        new ITMDepthTracker_HIP(Vector2i(TW, TH), regime, 5, 0, 0.1f * 0.1f, 1e-3f, &low)};
  ITMRenderState* rs[2]; ITMView* view[2] = {NULL, NULL}; ITMTrackingState* ts[2];
  for (int e = 0; e < 2; ++e) { reco[e]->ResetScene(scene[e]); rs[e] = vis[e]->CreateRenderState(Vector2i(TW, TH)); ts[e] = new ITMTrackingState(Vector2i(TW, TH), MEMORYDEVICE_CPU); }
  ITMUChar4Image rgb(Vector2i(TW, TH), true, false);
  ITMShortImage raw(Vector2i(TW, TH), true, false);
  HipSetMirrorPolicy(HIP_MIRROR_ON_DEMAND);
  double maxDiff = 0; bool ok = true; std::string why;
  for (int k = 0; k < frames && ok; ++k) {
    const float tx = 0.008f * (float)k;                 // 8 mm per frame: the tracker has something to find
    // three spheres off the axis in front of a tilted wall: every one of the six degrees of freedom is observable (ONE sphere on the
    // optical axis before a wall parallel to the image leaves the rotation about that axis to the rounding of the sums, and the two
    // stacks -- whose sums run in different orders -- wander apart in it)
    short* r = raw.GetData(MEMORYDEVICE_CPU);
    static const float sph[3][4] = {{0.0f, 0.0f, 1.5f, 0.5f}, {0.45f, -0.25f, 1.9f, 0.3f}, {-0.5f, 0.3f, 1.7f, 0.25f}};
    for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x) {
      const float dx = ((float)x - 320.0f) / 580.0f, dy = ((float)y - 240.0f) / 580.0f;
      float z = (2.5f + 0.15f * tx) / (1.0f - 0.15f * dx + 0.1f * dy);      // the wall z = 2.5 + 0.15 x - 0.1 y
      for (int q = 0; q < 3; ++q) {
        const float ox = tx - sph[q][0], oy = -sph[q][1], oz = -sph[q][2];
        const float A = dx * dx + dy * dy + 1.0f, B = 2.0f * (ox * dx + oy * dy + oz), C = ox * ox + oy * oy + oz * oz - sph[q][3] * sph[q][3], disc = B * B - 4.0f * A * C;
        if (disc > 0) { const float t = (-B - std::sqrt(disc)) / (2.0f * A); if (t > 0 && t < z) z = t; }
      }
      r[x + y * TW] = (short)(z * 1000.0f);
    }
    for (int e = 0; e < 2; ++e) {
      vb[e]->UpdateView(&view[e], &rgb, &raw, false, false);
      if (k > 0) trk[e]->TrackCamera(ts[e], view[e]);                       // frame 0: identity, as ITMMainEngine starts
      reco[e]->AllocateSceneFromDepth(scene[e], view[e], ts[e], rs[e]);
      reco[e]->IntegrateIntoScene(scene[e], view[e], ts[e], rs[e]);
      vis[e]->CreateExpectedDepths(ts[e]->pose_d, &view[e]->calib->intrinsics_d, rs[e]);
      vis[e]->CreateICPMaps(view[e], ts[e], rs[e]);
    }
    const Matrix4f a = ts[0]->pose_d->GetM(), b = ts[1]->pose_d->GetM();
    for (int i = 0; i < 16; ++i) maxDiff = std::fmax(maxDiff, std::fabs((double)a.m[i] - b.m[i]));
    if (std::getenv("ITM_DEMO_VERBOSE")) for (int e = 0; e < 2; ++e) { const Matrix4f m = ts[e]->pose_d->GetM(); std::fprintf(stderr, "frame %d %s:", k, e ? "hip" : "cpu"); for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %.6f", m.m[i]); std::fprintf(stderr, "\n"); }
    if (!(maxDiff < 1e-4)) { ok = false; why = "pose of frame " + std::to_string(k) + " differs by " + std::to_string(maxDiff); }
    if (k > 0 && !(std::fabs(b.m[12] - (-tx)) < 3e-3f)) { ok = false; why = "the HIP stack lost track at frame " + std::to_string(k); }
  }
  // what a host does before it looks at the maps
  HipSyncRenderStateToHost<ITMVoxelBlockHash>(rs[1]); HipSyncTrackingStateToHost(ts[1]); HipSyncViewToHost(view[1]);
  long both = 0, sameHit = 0, apart = 0; double maxPoint = 0;
  const Vector4f* pa = ts[0]->pointCloud->locations->GetData(MEMORYDEVICE_CPU); const Vector4f* pb = ts[1]->pointCloud->locations->GetData(MEMORYDEVICE_CPU);
  for (int i = 0; i < TW * TH; ++i) {
    if ((pa[i].w > 0) == (pb[i].w > 0)) ++sameHit;
    if (pa[i].w > 0 && pb[i].w > 0) {
      ++both;
      const double d = std::fmax(std::fabs(pa[i].x - pb[i].x), std::fmax(std::fabs(pa[i].y - pb[i].y), std::fabs(pa[i].z - pb[i].z)));
      if (d > 2e-3) ++apart; else maxPoint = std::fmax(maxPoint, d);
    }
  }
  // (the two stacks fuse from poses that differ in the fifth digit, so their maps agree to a fraction of a voxel, not bit for bit: within
  // 2 mm everywhere but on the silhouettes, where a ray of one stack grazes a sphere and the other stack's passes it and meets the wall)
  if (ok && !(both > 250000 && sameHit > (long)(0.995 * TW * TH) && apart < (long)(0.005 * TW * TH))) { ok = false; why = "ICP maps of the two stacks differ"; }
  const float* da = view[0]->depth->GetData(MEMORYDEVICE_CPU); const float* db = view[1]->depth->GetData(MEMORYDEVICE_CPU);
  if (ok && !same(da, db, (size_t)TW * TH)) { ok = false; why = "float depth of the view (HipSyncViewToHost)"; }
  std::printf("{\"config\": \"closed loop on_demand: ITMViewBuilder_HIP + ITMDepthTracker_HIP (maps from HBM) + HIP engines vs the reference's CPU stack\", \"frames\": %d, \"equal\": %s, "
              "\"max_pose_diff\": %.3g, \"tx\": %.6f, \"icp_points\": %ld, \"max_point_diff_m\": %.3g, \"silhouette_pixels_apart\": %ld, \"mismatch\": \"%s\"}\n",
              frames, ok ? "true" : "false", maxDiff, ts[1]->pose_d->GetM().m[12], both, maxPoint, apart, why.c_str());
  HipSetMirrorPolicy(HIP_MIRROR_EAGER);
  for (int e = 0; e < 2; ++e) { HipReleaseView(view[e]); HipReleaseTrackingState(ts[e]); delete rs[e]; delete view[e]; delete ts[e]; delete trk[e]; delete vb[e]; delete reco[e]; delete vis[e]; }
  return ok;
}

// ONE ITMView for every frame, as ITMMainEngine holds it: under HIP_MIRROR_ON_DEMAND the host overwrites view->depth with the next frame
// as soon as CreateICPMaps has returned -- nothing waits for the device in between, which is still fusing earlier frames.  The staging ring
// of the adapter (HipStageView) must have READ the host image before the call that staged it returns, and the frames in flight must keep
// their own device slot.  Final scene and maps bit-equal to the reference's CPU engines fed the same sequence.
static bool run_reused_view(int frames) {
  W = 640; H = 480;
  ITMSceneParams sp(0.02f, 100, 0.01f, 0.2f, 3.0f, false);
  ITMRGBDCalib calib; set_calib(calib);
  std::vector<std::vector<float> > pool((size_t)frames, std::vector<float>((size_t)W * H));
  for (int k = 0; k < frames; ++k) make_depth(pool[(size_t)k].data(), 0.01f * (float)k, 0.004f * (float)(k % 3));
  ITMScene<ITMVoxel_s, ITMVoxelBlockHash> sceneA(&sp, false, MEMORYDEVICE_CPU), sceneB(&sp, false, MEMORYDEVICE_CPU);
  ITMScene<ITMVoxel_s, ITMVoxelBlockHash>* scene[2] = {&sceneA, &sceneB};
  ITMSceneReconstructionEngine<ITMVoxel_s, ITMVoxelBlockHash>* reco[2] = {new ITMSceneReconstructionEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>(), new ITMSceneReconstructionEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>()};
  ITMVisualisationEngine<ITMVoxel_s, ITMVoxelBlockHash>* vis[2] = {new ITMVisualisationEngine_CPU<ITMVoxel_s, ITMVoxelBlockHash>(&sceneA), new ITMVisualisationEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>(&sceneB)};
  HipSetMirrorPolicy(HIP_MIRROR_ON_DEMAND);
  ITMRenderState* rs[2]; ITMView* view[2]; ITMTrackingState* ts[2];
  for (int e = 1; e >= 0; --e) {
    reco[e]->ResetScene(scene[e]); rs[e] = vis[e]->CreateRenderState(Vector2i(W, H));
    view[e] = new ITMView(&calib, Vector2i(W, H), Vector2i(W, H), false); ts[e] = new ITMTrackingState(Vector2i(W, H), MEMORYDEVICE_CPU);
    for (int k = 0; k < frames; ++k) {
      std::memcpy(view[e]->depth->GetData(MEMORYDEVICE_CPU), pool[(size_t)k].data(), (size_t)W * H * sizeof(float));
      if (e) HipMarkViewUpdated(view[e]);
      Matrix4f M; M.setIdentity(); M.m[12] = -0.01f * (float)k; M.m[13] = -0.004f * (float)(k % 3);
      ts[e]->pose_d->SetM(M);
      reco[e]->AllocateSceneFromDepth(scene[e], view[e], ts[e], rs[e]);
      reco[e]->IntegrateIntoScene(scene[e], view[e], ts[e], rs[e]);
      vis[e]->CreateExpectedDepths(ts[e]->pose_d, &view[e]->calib->intrinsics_d, rs[e]);
      vis[e]->CreateICPMaps(view[e], ts[e], rs[e]);
    }
    // (the last frame's image is scribbled over too: by now its staging call has returned)
    if (e) std::memset(view[e]->depth->GetData(MEMORYDEVICE_CPU), 0, (size_t)W * H * sizeof(float));
  }
  HipSyncRenderStateToHost<ITMVoxelBlockHash>(rs[1]); HipSyncTrackingStateToHost(ts[1]);
  ITMSceneReconstructionEngine_HIP<ITMVoxel_s, ITMVoxelBlockHash>::SyncSceneToHost(&sceneB);
  bool ok = true; std::string why;
  if (!VisibleCmp<ITMVoxelBlockHash>::Equal(rs[0], rs[1], why)) ok = false;
  else if (!VisibleCmp<ITMVoxelBlockHash>::SceneEqual(sceneA.index, sceneB.index, why)) ok = false;
  else if (!same(sceneA.localVBA.GetVoxelBlocks(), sceneB.localVBA.GetVoxelBlocks(), VisibleCmp<ITMVoxelBlockHash>::Voxels(sceneA.index))) { ok = false; why = "voxel blocks"; }
  else if (!same(ts[0]->pointCloud->locations->GetData(MEMORYDEVICE_CPU), ts[1]->pointCloud->locations->GetData(MEMORYDEVICE_CPU), (size_t)W * H)) { ok = false; why = "pointsMap"; }
  else if (!same(ts[0]->pointCloud->colours->GetData(MEMORYDEVICE_CPU), ts[1]->pointCloud->colours->GetData(MEMORYDEVICE_CPU), (size_t)W * H)) { ok = false; why = "normalsMap"; }
  long long hits = 0; const Vector4f* p = ts[1]->pointCloud->locations->GetData(MEMORYDEVICE_CPU);
  for (int i = 0; i < W * H; ++i) hits += p[i].w > 0;
  std::printf("{\"config\": \"one ITMView rewritten by the host for every frame, on_demand, no wait in between\", \"frames\": %d, \"equal\": %s, \"icp_points\": %lld, "
              "\"lastFreeBlockId\": %d, \"mismatch\": \"%s\"}\n", frames, ok ? "true" : "false", hits, sceneB.localVBA.lastFreeBlockId, why.c_str());
  HipSetMirrorPolicy(HIP_MIRROR_EAGER);
  for (int e = 0; e < 2; ++e) { HipReleaseView(view[e]); HipReleaseTrackingState(ts[e]); delete rs[e]; delete view[e]; delete ts[e]; delete reco[e]; delete vis[e]; }
  W = 160; H = 120;
  return ok;
}

int main(int argc, char** argv) {
  std::printf("{\"library\": \"%s\"}\n", itm_version());
  if (argc >= 2 && std::string(argv[1]) == "--bench") return run_bench(argc >= 3 ? std::atoi(argv[2]) : 300) ? 0 : 1;
  bool ok = true;
  for (int pol = 0; pol < 2; ++pol) {
    const HipMirrorPolicy policy = pol ? HIP_MIRROR_ON_DEMAND : HIP_MIRROR_EAGER;
    ok &= run<ITMVoxel_s, ITMVoxelBlockHash>("hash ITMVoxel_s 10 mm", 0.01f, 4, policy);
    ok &= run<ITMVoxel_f_rgb, ITMVoxelBlockHash>("hash ITMVoxel_f_rgb 10 mm", 0.01f, 3, policy);
    ok &= run<ITMVoxel_s, ITMPlainVoxelArray>("dense 128^3 ITMVoxel_s 10 mm", 0.01f, 3, policy);
  }
  ok &= run_view_builder();
  ok &= run_tracker();
  ok &= run_closed_loop(6);
  ok &= run_reused_view(12);
  return ok ? 0 : 1;
}
