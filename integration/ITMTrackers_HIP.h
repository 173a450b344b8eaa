// ITMTrackers_HIP.h -- reference-side bindings for the steps either side of the path: a depth
// tracker and a view builder derived from the reference's OWN base classes (Engine/ITMDepthTracker.h:23-74,
// Engine/ITMViewBuilder.h:17-60) that forward their device-specific virtuals to the C-ABI of include/itm_hip.h.
// New code against the reference's public interfaces; compiled only where the reference tree is on the include path.
//
// As in ITMEngines_HIP.h, this variant serves a reference build without CUDA: host images are staged into HBM.  The
// tracker keeps the reference's own TrackCamera (hierarchy, Levenberg-Marquardt loop, pose algebra) and only replaces
// ComputeGandH -- exactly the split between ITMDepthTracker and ITMDepthTracker_CUDA.
#pragma once

#include "ITMLib/Engine/ITMDepthTracker.h"
#include "ITMLib/Engine/ITMViewBuilder.h"
#include "ITMEngines_HIP.h"

namespace ITMLib {
namespace Engine {

class ITMDepthTracker_HIP : public ITMDepthTracker {
  void* devPoints = nullptr; void* devNormals = nullptr; void* devDepth = nullptr;
  const void* mapPoints = nullptr; const void* mapNormals = nullptr;      // the maps of this call: the registry's (written by CreateICPMaps) or the staged copies
  size_t mapPixels = 0, depthPixels = 0;
  const void* stagedDepth = nullptr; int lastLevel = -1;

  void Stage() {
    const Vector2i ss = sceneHierarchyLevel->pointsMap->noDims, ds = viewHierarchyLevel->depth->noDims;
    const size_t mp = (size_t)ss.x * ss.y, dp = (size_t)ds.x * ds.y;
    if (mp != mapPixels) {
      itm_dev_free(devPoints); itm_dev_free(devNormals);
      HipCheck(itm_dev_malloc(&devPoints, mp * 16), "dev_malloc"); HipCheck(itm_dev_malloc(&devNormals, mp * 16), "dev_malloc");
      mapPixels = mp; lastLevel = -1;
    }
    if (dp > depthPixels) { itm_dev_free(devDepth); HipCheck(itm_dev_malloc(&devDepth, dp * 4), "dev_malloc"); depthPixels = dp; stagedDepth = nullptr; }
    // TrackCamera walks the levels from coarse (high id) to fine: a level id above the previous one means a new call,
    // i.e. new ICP maps.  When CreateICPMaps of the HIP visualisation engine wrote them, they are still in HBM (whole maps): taken
    // from there, nothing is uploaded
    if (levelId > lastLevel) {
      mapPoints = devPoints; mapNormals = devNormals;
      const HipRegistry::Maps* fresh = nullptr;
      for (auto& kv : HipRegistry::Get().maps) {
        if (kv.second.locationsImage != (const void*)sceneHierarchyLevel->pointsMap) continue;
        // the maps in HBM only while they ARE the last CreateICPMaps' (icpMaps: not a point list of CreatePointCloud, not superseded by
        // a host-side write, HipMarkTrackingStateHostWritten); otherwise the host image is uploaded -- after bringing it up to date when
        // the newest content still lies in HBM only (HIP_MIRROR_ON_DEMAND)
        if (kv.second.icpMaps && kv.second.pixels == mp && kv.second.count == mp) fresh = &kv.second;
        else if (kv.second.hostStale) HipSyncTrackingStateToHost((ITMTrackingState*)kv.first);
      }
      if (fresh) { mapPoints = fresh->points; mapNormals = fresh->normals; }
      else {
        HipCheck(itm_memcpy_h2d(devPoints, sceneHierarchyLevel->pointsMap->GetData(MEMORYDEVICE_CPU), mp * 16, 0), "memcpy_h2d");
        HipCheck(itm_memcpy_h2d(devNormals, sceneHierarchyLevel->normalsMap->GetData(MEMORYDEVICE_CPU), mp * 16, 0), "memcpy_h2d");
      }
      stagedDepth = nullptr;
    }
    const float* hostDepth = viewHierarchyLevel->depth->GetData(MEMORYDEVICE_CPU);
    if (stagedDepth != hostDepth || levelId != lastLevel) {
      HipCheck(itm_memcpy_h2d(devDepth, hostDepth, dp * 4, 0), "memcpy_h2d");
      stagedDepth = hostDepth;
    }
    lastLevel = levelId;
  }

 protected:
  int ComputeGandH(float& f, float* nabla, float* hessian, Matrix4f approxInvPose) {
    if (iterationType == TRACKER_ITERATION_NONE) return 0;
    Stage();
    const Vector2i ss = sceneHierarchyLevel->pointsMap->noDims, ds = viewHierarchyLevel->depth->noDims;
    const int it = iterationType == TRACKER_ITERATION_ROTATION ? ITM_TRACKER_ITERATION_ROTATION
                   : iterationType == TRACKER_ITERATION_TRANSLATION ? ITM_TRACKER_ITERATION_TRANSLATION : ITM_TRACKER_ITERATION_BOTH;
    itm_tracker_gh gh;
    HipCheck(itm_tracker_compute_g_and_h((const float*)devDepth, ds.x, ds.y, &viewHierarchyLevel->intrinsics.x, (const float*)mapPoints,
                                         (const float*)mapNormals, ss.x, ss.y, &sceneHierarchyLevel->intrinsics.x, approxInvPose.m, scenePose.m,
                                         distThresh[levelId], it, &gh, 0), "ComputeGandH");
    const int noPara = (it == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
    for (int r = 0; r < noPara; ++r) for (int c = 0; c < noPara; ++c) hessian[r + c * 6] = gh.hessian[r + c * 6];
    for (int r = 0; r < noPara; ++r) nabla[r] = gh.nabla[r];
    f = gh.f;
    return gh.noValidPoints;
  }

 public:
  ITMDepthTracker_HIP(Vector2i imgSize, TrackerIterationType* trackingRegime, int noHierarchyLevels, int noICPRunTillLevel, float distThresh,
                      float terminationThreshold, const ITMLowLevelEngine* lowLevelEngine)
      : ITMDepthTracker(imgSize, trackingRegime, noHierarchyLevels, noICPRunTillLevel, distThresh, terminationThreshold, lowLevelEngine, MEMORYDEVICE_CPU) {}
  ~ITMDepthTracker_HIP() { itm_dev_free(devPoints); itm_dev_free(devNormals); itm_dev_free(devDepth); }

  // the reference's own TrackCamera builds the depth pyramid on the HOST (ITMDepthTracker::PrepareForEvaluation, ITMDepthTracker.cpp:62-75,
  // from view->depth): under HIP_MIRROR_ON_DEMAND a float depth that ITMViewBuilder_HIP produced in HBM comes back first.  The ICP maps
  // are only ever read at level 0 (SetEvaluationParams, ITMDepthTracker.cpp:81) and only by ComputeGandH: they stay in HBM.
  void TrackCamera(ITMTrackingState* trackingState, const ITMView* view) {
    HipSyncViewToHost(const_cast<ITMView*>(view));
    ITMDepthTracker::TrackCamera(trackingState, view);
  }
};

class ITMViewBuilder_HIP : public ITMViewBuilder {
  void* devRaw = nullptr; void* devA = nullptr; void* devB = nullptr; void* devN = nullptr; void* devS = nullptr;
  void* pinnedRaw = nullptr;
  itm_depth_stager* rawRing = nullptr; int ringW = 0, ringH = 0;
  bool ringPlain = false, rawHeld = false; int ringCalib = -1; float ringC0 = 0, ringC1 = 0, ringFx = 0;
  size_t pixels = 0;
  void Ensure(size_t px) {
    if (px == pixels) return;
    itm_dev_free(devRaw); itm_dev_free(devA); itm_dev_free(devB); itm_dev_free(devN); itm_dev_free(devS);
    HipCheck(itm_dev_malloc(&devRaw, px * 2), "dev_malloc"); HipCheck(itm_dev_malloc(&devA, px * 4), "dev_malloc"); HipCheck(itm_dev_malloc(&devB, px * 4), "dev_malloc");
    HipCheck(itm_dev_malloc(&devN, px * 16), "dev_malloc"); HipCheck(itm_dev_malloc(&devS, px * 4), "dev_malloc");
    pixels = px;
  }

 public:
  explicit ITMViewBuilder_HIP(const ITMRGBDCalib* calib) : ITMViewBuilder(calib) {}
  ~ITMViewBuilder_HIP() { if (rawRing) { itm_stream_synchronize(0); itm_depth_stager_destroy(rawRing); } if (pinnedRaw) itm_host_unregister(pinnedRaw); itm_dev_free(devRaw); itm_dev_free(devA); itm_dev_free(devB); itm_dev_free(devN); itm_dev_free(devS); }

  void ConvertDisparityToDepth(ITMFloatImage* depth_out, const ITMShortImage* disp_in, const ITMIntrinsics* depthIntrinsics, Vector2f disparityCalibParams) {
    const size_t px = (size_t)disp_in->noDims.x * disp_in->noDims.y; Ensure(px);
    HipCheck(itm_memcpy_h2d(devRaw, disp_in->GetData(MEMORYDEVICE_CPU), px * 2, 0), "memcpy_h2d");
    HipCheck(itm_convert_disparity((const int16_t*)devRaw, (float*)devA, disp_in->noDims.x, disp_in->noDims.y, disparityCalibParams.x, disparityCalibParams.y,
                                   depthIntrinsics->projectionParamsSimple.fx, 0), "ConvertDisparityToDepth");
    HipCheck(HipDownload(depth_out->GetData(MEMORYDEVICE_CPU), devA, px * 4, 0), "memcpy_d2h");
  }
  void ConvertDepthAffineToFloat(ITMFloatImage* depth_out, const ITMShortImage* depth_in, Vector2f depthCalibParams) {
    const size_t px = (size_t)depth_in->noDims.x * depth_in->noDims.y; Ensure(px);
    HipCheck(itm_memcpy_h2d(devRaw, depth_in->GetData(MEMORYDEVICE_CPU), px * 2, 0), "memcpy_h2d");
    HipCheck(itm_convert_depth_affine((const int16_t*)devRaw, (float*)devA, depth_in->noDims.x, depth_in->noDims.y, depthCalibParams.x, depthCalibParams.y, 0), "ConvertDepthAffineToFloat");
    HipCheck(HipDownload(depth_out->GetData(MEMORYDEVICE_CPU), devA, px * 4, 0), "memcpy_d2h");
  }
  void DepthFiltering(ITMFloatImage* image_out, const ITMFloatImage* image_in) {
    const size_t px = (size_t)image_in->noDims.x * image_in->noDims.y; Ensure(px);
    HipCheck(itm_memcpy_h2d(devA, image_in->GetData(MEMORYDEVICE_CPU), px * 4, 0), "memcpy_h2d");
    HipCheck(itm_filter_depth((const float*)devA, (float*)devB, image_in->noDims.x, image_in->noDims.y, 0), "DepthFiltering");
    HipCheck(HipDownload(image_out->GetData(MEMORYDEVICE_CPU), devB, px * 4, 0), "memcpy_d2h");
  }
  void ComputeNormalAndWeights(ITMFloat4Image* normal_out, ITMFloatImage* sigmaZ_out, const ITMFloatImage* depth_in, Vector4f intrinsic) {
    const size_t px = (size_t)depth_in->noDims.x * depth_in->noDims.y; Ensure(px);
    HipCheck(itm_memcpy_h2d(devA, depth_in->GetData(MEMORYDEVICE_CPU), px * 4, 0), "memcpy_h2d");
    // pixels the kernel rejects keep their previous content, as on the host
    HipCheck(itm_memcpy_h2d(devN, normal_out->GetData(MEMORYDEVICE_CPU), px * 16, 0), "memcpy_h2d");
    HipCheck(itm_memcpy_h2d(devS, sigmaZ_out->GetData(MEMORYDEVICE_CPU), px * 4, 0), "memcpy_h2d");
    HipCheck(itm_compute_normal_and_weights((const float*)devA, (float*)devN, (float*)devS, depth_in->noDims.x, depth_in->noDims.y, &intrinsic.x, 0), "ComputeNormalAndWeights");
    HipCheck(HipDownload(normal_out->GetData(MEMORYDEVICE_CPU), devN, px * 16, 0), "memcpy_d2h");
    HipCheck(HipDownload(sigmaZ_out->GetData(MEMORYDEVICE_CPU), devS, px * 4, 0), "memcpy_d2h");
  }

  // same object management as ITMViewBuilder_CPU::UpdateView; the image work is one itm_update_view call
  void UpdateView(ITMView** view_ptr, ITMUChar4Image* rgbImage, ITMShortImage* rawDepthImage, bool useBilateralFilter, bool modelSensorNoise = false) {
    if (*view_ptr == NULL) {
      *view_ptr = new ITMView(calib, rgbImage->noDims, rawDepthImage->noDims, false);
      if (modelSensorNoise) {
        (*view_ptr)->depthNormal = new ITMFloat4Image(rawDepthImage->noDims, true, false);
        (*view_ptr)->depthUncertainty = new ITMFloatImage(rawDepthImage->noDims, true, false);
      }
    }
    ITMView* view = *view_ptr;
    view->rgb->SetFrom(rgbImage, ORUtils::MemoryBlock<Vector4u>::CPU_TO_CPU);
    const int w = rawDepthImage->noDims.x, h = rawDepthImage->noDims.y;
    const size_t px = (size_t)w * h; Ensure(px);
    // the raw frame goes through a ring of device slots on a copy stream of its own (itm_depth_stager): the upload leaves the frame's
    // stream, and the host waits only until the copy has READ the image source's buffer -- which the source reuses for the next frame
    // (the reference's CUDA view builder copies synchronously here, ITMViewBuilder_CUDA.cu:53)
    int16_t* rawHost = rawDepthImage->GetData(MEMORYDEVICE_CPU);
    const ITMDisparityCalib& dc = view->calib->disparityCalib;
    const int calibType = dc.type == ITMDisparityCalib::TRAFO_KINECT ? 0 : 1;
    const float fx = view->calib->intrinsics_d.projectionParamsSimple.all.x;
    // without filter and noise model the float depth is the conversion alone: the ring's copy does it too (itm_depth_stager_set_conversion)
    // and the slot's float image becomes the view's device stage -- no conversion launch on the frame's stream.  (The stage then lives
    // in this builder's ring: no engine call may use the view after its builder has been deleted.  ITMMainEngine's destructor -- builder,
    // then view, Engine/ITMMainEngine.cpp:86-89 -- is fine: nothing reads the stage in between, HipReleaseView only frees what it owns.)
    const bool plain = !useBilateralFilter && !modelSensorNoise;
    if (rawHeld) { HipCheck(itm_depth_stager_release(rawRing, 0), "raw ring release"); rawHeld = false; }      // the previous frame's slot: its readers have been submitted
    if (!rawRing || ringW != w || ringH != h || ringPlain != plain || (plain && (ringCalib != calibType || ringC0 != dc.params.x || ringC1 != dc.params.y || ringFx != fx))) {
      if (rawRing) { itm_stream_synchronize(0); itm_depth_stager_destroy(rawRing); }
      HipCheck(itm_depth_stager_create(w, h, 4, &rawRing), "raw ring"); ringW = w; ringH = h; ringPlain = plain;
      if (plain) {
        HipCheck(itm_depth_stager_set_conversion(rawRing, calibType, dc.params.x, dc.params.y, fx), "raw ring conversion");
        ringCalib = calibType; ringC0 = dc.params.x; ringC1 = dc.params.y; ringFx = fx;
      }
    }
    HipPin(pinnedRaw, rawHost, px * 2);
    HipCheck(itm_depth_stager_upload(rawRing, rawHost), "raw ring upload");
    for (int busy = 1; busy;) HipCheck(itm_depth_stager_pending(rawRing, nullptr, &busy), "raw ring pending");
    if (plain) {
      const float* converted = nullptr;
      HipCheck(itm_depth_stager_acquire_depth(rawRing, 0, nullptr, &converted), "raw ring acquire");
      rawHeld = true;
      HipRegistry::Stage& st = HipStageOf(view);
      if (st.holding) { HipCheck(itm_depth_stager_release(st.ring, 0), "depth ring release"); st.holding = false; }      // (a view that used to be staged from the host)
      HipMarkViewUpdated(view);
      st.depthStaged = st.generation; st.hostDepthStale = true; st.cur = converted;
      if (HipEager()) HipSyncViewToHost(view);
      return;
    }
    const int16_t* rawDev = nullptr;
    HipCheck(itm_depth_stager_acquire(rawRing, 0, &rawDev), "raw ring acquire");
    if (modelSensorNoise) {
      HipCheck(itm_memcpy_h2d(devN, view->depthNormal->GetData(MEMORYDEVICE_CPU), px * 16, 0), "memcpy_h2d");
      HipCheck(itm_memcpy_h2d(devS, view->depthUncertainty->GetData(MEMORYDEVICE_CPU), px * 4, 0), "memcpy_h2d");
    }
    // the float depth is written straight into the view's device stage: the engine calls of this frame find it there (no float
    // image crosses PCIe towards the device), the host image follows now or on request (HipSyncViewToHost)
    HipRegistry::Stage& st = HipStageOf(view);
    if (st.holding) { HipCheck(itm_depth_stager_release(st.ring, 0), "depth ring release"); st.holding = false; }      // (a view that used to be staged from the host)
    HipCheck(itm_update_view(rawDev, w, h, calibType, dc.params.x, dc.params.y,
                             &view->calib->intrinsics_d.projectionParamsSimple.all.x, useBilateralFilter ? 1 : 0, modelSensorNoise ? 1 : 0,
                             (float*)st.depth, (float*)devB, (float*)devN, (float*)devS, 0), "UpdateView");
    HipCheck(itm_depth_stager_release(rawRing, 0), "raw ring release");      // the conversion is what read the slot
    HipMarkViewUpdated(view);
    st.depthStaged = st.generation; st.hostDepthStale = true; st.cur = st.depth;
    if (HipEager()) HipSyncViewToHost(view);
    if (modelSensorNoise) {
      HipCheck(HipDownload(view->depthNormal->GetData(MEMORYDEVICE_CPU), devN, px * 16, 0), "memcpy_d2h");
      HipCheck(HipDownload(view->depthUncertainty->GetData(MEMORYDEVICE_CPU), devS, px * 4, 0), "memcpy_d2h");
    }
  }
  void UpdateView(ITMView** view_ptr, ITMUChar4Image* rgbImage, ITMFloatImage* depthImage) {
    if (*view_ptr == NULL) *view_ptr = new ITMView(calib, rgbImage->noDims, depthImage->noDims, false);
    // host build: the caller already wrote the float depth into the view (ITMViewBuilder_CPU.cpp:65-74)
    HipMarkViewUpdated(*view_ptr);
  }
  void UpdateView(ITMView** view_ptr, ITMUChar4Image* rgbImage, ITMShortImage* depthImage, bool useBilateralFilter, ITMIMUMeasurement* imuMeasurement) {
    if (*view_ptr == NULL) *view_ptr = new ITMViewIMU(calib, rgbImage->noDims, depthImage->noDims, false);
    ((ITMViewIMU*)(*view_ptr))->imu->SetFrom(imuMeasurement);
    this->UpdateView(view_ptr, rgbImage, depthImage, useBilateralFilter);
  }
};

}  // namespace Engine
}  // namespace ITMLib
