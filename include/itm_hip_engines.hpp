// itm_hip_engines.hpp -- C++ host side above the C-ABI (include/itm_hip.h).
//
// Header-only adapter classes with the method names, argument meaning and (void / throwing) error
// behaviour of the reference's engine interfaces, so that code written against
//   ITMLib::Engine::ITMSceneReconstructionEngine<TVoxel,TIndex>   (Engine/ITMSceneReconstructionEngine.h:28-52)
//   ITMLib::Engine::ITMVisualisationEngine<TVoxel,TIndex>         (Engine/ITMVisualisationEngine.h:18-107)
// reads the same against the HIP back-end.  The reference's own object headers are NOT included:
// the few host-side value types the methods take (pose, intrinsics, view, tracking state, scene
// shell, render-state shell) are re-declared here as thin shells over device handles.  A maintainer
// of the reference instead derives two classes from the reference's abstract engines and forwards
// to the same C functions -- see INTEGRATION.md for that stub.
//
// Everything lives in HBM and is owned by the C library; the shells only carry handles.  Device
// errors become std::runtime_error (the reference prints and exit(-1)s: ORUtils/CUDADefines.h:27-36).
#pragma once

#include <cstring>
#include <stdexcept>
#include <string>

#include "itm_hip.h"

namespace itmhip {

inline void check(int rc, const char* what) {
  if (rc != ITM_OK) throw std::runtime_error(std::string(what) + ": " + itm_last_error());
}

// voxel / index tags replacing the reference's TVoxel / TIndex template arguments
struct ITMVoxel_s { static constexpr int kType = ITM_VOXEL_S; };
struct ITMVoxel_f { static constexpr int kType = ITM_VOXEL_F; };
struct ITMVoxel_s_rgb { static constexpr int kType = ITM_VOXEL_S_RGB; };
struct ITMVoxel_f_rgb { static constexpr int kType = ITM_VOXEL_F_RGB; };
struct ITMVoxelBlockHash { static constexpr int kType = ITM_INDEX_HASH; };
struct ITMPlainVoxelArray { static constexpr int kType = ITM_INDEX_DENSE; };

struct Vector2i { int x, y; };

// ITMPose: only the model-view matrix is needed by the path (Objects/ITMPose.h GetM()).
struct ITMPose {
  float M[16];
  ITMPose() { std::memset(M, 0, sizeof M); M[0] = M[5] = M[10] = M[15] = 1.0f; }
  void SetM(const float* m) { std::memcpy(M, m, sizeof M); }
  const float* GetM() const { return M; }
};

// ITMIntrinsics::projectionParamsSimple.all (Objects/ITMIntrinsics.h)
struct ITMIntrinsics {
  float all[4];
  ITMIntrinsics() { SetFrom(580, 580, 320, 240); }
  void SetFrom(float fx, float fy, float cx, float cy) { all[0] = fx; all[1] = fy; all[2] = cx; all[3] = cy; }
};

// ITMRGBDCalib (Objects/ITMRGBDCalib.h): intrinsics + rgb->depth extrinsics
struct ITMRGBDCalib {
  ITMIntrinsics intrinsics_rgb, intrinsics_d;
  float trafo_rgb_to_depth_calib[16], trafo_rgb_to_depth_calib_inv[16];
  ITMRGBDCalib() {
    std::memset(trafo_rgb_to_depth_calib, 0, 64);
    trafo_rgb_to_depth_calib[0] = trafo_rgb_to_depth_calib[5] = trafo_rgb_to_depth_calib[10] = trafo_rgb_to_depth_calib[15] = 1.0f;
    std::memcpy(trafo_rgb_to_depth_calib_inv, trafo_rgb_to_depth_calib, 64);
  }
};

// ITMView (Objects/ITMView.h): device images + calibration
struct ITMView {
  ITMRGBDCalib calib;
  const float* depth = nullptr;  // device float[h*w]
  const uint8_t* rgb = nullptr;  // device uchar4[h_rgb*w_rgb]
  Vector2i depthSize{0, 0}, rgbSize{0, 0};
};

struct ITMSceneParamsDefaults { float mu = 0.02f; int maxW = 100; float voxelSize = 0.005f, viewFrustum_min = 0.35f, viewFrustum_max = 3.0f; bool stopIntegratingAtMaxW = false; };

// ITMTrackingState (Objects/ITMTrackingState.h): pose + the ICP maps written by CreateICPMaps
struct ITMTrackingState {
  ITMPose pose_d, pose_pointCloud;
  float* pointCloud_locations = nullptr;  // device Vector4f[h*w]
  float* pointCloud_colours = nullptr;    // device Vector4f[h*w] (normals for the ICP tracker)
  int age_pointCloud = -1;
  bool requiresFullRendering = true;

  // camera centre -1 * (R^T T) of a world->camera matrix, component i = column i of R dotted with T, summed left to right
  // (Matrix3::t() and Matrix3 * Vector3 of ORUtils/Matrix.h:270-296)
  static void CameraCentre(const float* M, float c[3]) {
    for (int i = 0; i < 3; ++i) c[i] = -1.0f * (M[0 + 4 * i] * M[12] + M[1 + 4 * i] * M[13] + M[2 + 4 * i] * M[14]);
  }
  // ITMTrackingState::TrackerFarFromPointCloud (Objects/ITMTrackingState.h:41-59)
  bool TrackerFarFromPointCloud() const {
    if (age_pointCloud < 0) return true;       // no point cloud exists yet
    if (age_pointCloud > 5) return true;       // older than n frames
    float pc[3], live[3];
    CameraCentre(pose_pointCloud.GetM(), pc);
    CameraCentre(pose_d.GetM(), live);
    const float dx = pc[0] - live[0], dy = pc[1] - live[1], dz = pc[2] - live[2];
    const float diff = dx * dx + dy * dy + dz * dz;
    return diff > 0.0005f;                     // the camera centre has moved by more than the threshold
  }
};

// ITMLibSettings (Utils/ITMLibSettings.h / .cpp:9-90): the members the path's callers read, with the reference's defaults
struct ITMLibSettings {
  enum TrackerType { TRACKER_COLOR, TRACKER_ICP, TRACKER_EXTERNAL };
  ITMSceneParamsDefaults sceneParamsDefaults;   // (0.02, 100, 0.005, 0.35, 3.0, false), ITMLibSettings.cpp:10
  float depthTrackerICPThreshold = 0.1f * 0.1f;
  float depthTrackerTerminationThreshold = 1e-3f;
  bool skipPoints = true;
  bool useApproximateRaycast = false;
  bool useBilateralFilter = false;
  bool modelSensorNoise = false;
  TrackerType trackerType = TRACKER_EXTERNAL;   // this fork's default: poses come from outside (ITMExternalTracker.cpp:27-30)
  int noHierarchyLevels = 5;
  int trackingRegime[8] = {ITM_TRACKER_ITERATION_BOTH, ITM_TRACKER_ITERATION_BOTH, ITM_TRACKER_ITERATION_ROTATION, ITM_TRACKER_ITERATION_ROTATION,
                           ITM_TRACKER_ITERATION_ROTATION, 0, 0, 0};
  int noICPRunTillLevel = 0;
};

struct ITMSceneParams : itm_scene_params {
  ITMSceneParams(float mu_, int maxW_, float voxelSize_, float vfMin, float vfMax, bool stopAtMax) {
    mu = mu_; maxW = maxW_; voxelSize = voxelSize_; viewFrustum_min = vfMin; viewFrustum_max = vfMax;
    stopIntegratingAtMaxW = stopAtMax ? 1 : 0;
  }
};

// ITMScene<TVoxel,TIndex> (Objects/ITMScene.h:37-43): owns the device scene
template <class TVoxel, class TIndex>
class ITMScene {
 public:
  itm_scene* handle = nullptr;
  const ITMSceneParams* sceneParams;
  bool useSwapping;
  explicit ITMScene(const ITMSceneParams* params, int localBlockNum = 0, int bucketNum = 0, int excessNum = 0, bool useSwapping_ = false)
      : sceneParams(params), useSwapping(useSwapping_) {
    itm_scene_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.voxelType = TVoxel::kType; cfg.indexType = TIndex::kType;
    cfg.localBlockNum = localBlockNum; cfg.bucketNum = bucketNum; cfg.excessNum = excessNum;
    cfg.useSwapping = useSwapping ? 1 : 0;        // the scene then owns an ITMGlobalCache in host memory (Objects/ITMScene.h:37-43)
    check(itm_scene_create(&cfg, params, &handle), "itm_scene_create");
    // the engines below are driven in the reference's call order by hosts that read results through the engines: the library may
    // record AllocateSceneFromDepth / IntegrateIntoScene / CreateExpectedDepths and launch the fused frame at CreateICPMaps
    // (include/itm_hip.h, "the four calls of a frame")
    check(itm_scene_set_deferred_fusion(handle, 1), "itm_scene_set_deferred_fusion");
  }
  ~ITMScene() { itm_scene_destroy(handle); }
  ITMScene(const ITMScene&) = delete;
  ITMScene& operator=(const ITMScene&) = delete;
};

// ITMRenderState / ITMRenderState_VH (Objects/ITMRenderState.h, ITMRenderState_VH.h)
class ITMRenderState {
 public:
  itm_render_state* handle = nullptr;
  Vector2i imgSize;
  ITMRenderState(itm_render_state* h, Vector2i s) : handle(h), imgSize(s) {}
  ~ITMRenderState() { itm_render_state_destroy(handle); }
  ITMRenderState(const ITMRenderState&) = delete;
  ITMRenderState& operator=(const ITMRenderState&) = delete;
};

inline itm_view make_view(const ITMView* view, const ITMTrackingState* ts) {
  itm_view v;
  std::memset(&v, 0, sizeof v);
  v.depth = view->depth; v.rgb = view->rgb;
  v.w = view->depthSize.x; v.h = view->depthSize.y; v.w_rgb = view->rgbSize.x; v.h_rgb = view->rgbSize.y;
  std::memcpy(v.M_d, ts->pose_d.GetM(), 64);
  std::memcpy(v.intr_d, view->calib.intrinsics_d.all, 16);
  std::memcpy(v.intr_rgb, view->calib.intrinsics_rgb.all, 16);
  std::memcpy(v.rgb_to_depth, view->calib.trafo_rgb_to_depth_calib, 64);
  std::memcpy(v.rgb_to_depth_inv, view->calib.trafo_rgb_to_depth_calib_inv, 64);
  return v;
}

// ITMSceneReconstructionEngine_HIP: same three methods as the reference interface
template <class TVoxel, class TIndex>
class ITMSceneReconstructionEngine_HIP {
 public:
  itm_stream stream = nullptr;
  void ResetScene(ITMScene<TVoxel, TIndex>* scene) { check(itm_reset_scene(scene->handle, stream), "ResetScene"); }
  void AllocateSceneFromDepth(ITMScene<TVoxel, TIndex>* scene, const ITMView* view, const ITMTrackingState* trackingState,
                              const ITMRenderState* renderState, bool onlyUpdateVisibleList = false) {
    itm_view v = make_view(view, trackingState);
    check(itm_allocate_scene_from_depth(scene->handle, &v, renderState->handle, onlyUpdateVisibleList ? 1 : 0, stream), "AllocateSceneFromDepth");
  }
  void IntegrateIntoScene(ITMScene<TVoxel, TIndex>* scene, const ITMView* view, const ITMTrackingState* trackingState,
                          const ITMRenderState* renderState) {
    itm_view v = make_view(view, trackingState);
    check(itm_integrate_into_scene(scene->handle, &v, renderState->handle, stream), "IntegrateIntoScene");
  }
};

// ITMVisualisationEngine_HIP: the IITMVisualisationEngine methods
template <class TVoxel, class TIndex>
class ITMVisualisationEngine_HIP {
  const ITMScene<TVoxel, TIndex>* scene;

 public:
  enum RenderImageType { RENDER_SHADED_GREYSCALE, RENDER_COLOUR_FROM_VOLUME, RENDER_COLOUR_FROM_NORMAL };
  itm_stream stream = nullptr;
  explicit ITMVisualisationEngine_HIP(const ITMScene<TVoxel, TIndex>* s) : scene(s) {}

  ITMRenderState* CreateRenderState(const Vector2i& imgSize) const {
    itm_render_state* h = nullptr;
    check(itm_render_state_create(scene->handle, imgSize.x, imgSize.y, &h), "CreateRenderState");
    return new ITMRenderState(h, imgSize);
  }
  void FindVisibleBlocks(const ITMPose* pose, const ITMIntrinsics* intrinsics, ITMRenderState* renderState) const {
    check(itm_find_visible_blocks(scene->handle, pose->GetM(), intrinsics->all, renderState->handle, stream), "FindVisibleBlocks");
  }
  void CreateExpectedDepths(const ITMPose* pose, const ITMIntrinsics* intrinsics, ITMRenderState* renderState) const {
    check(itm_create_expected_depths(scene->handle, pose->GetM(), intrinsics->all, renderState->handle, stream), "CreateExpectedDepths");
  }
  // outputImage: device uchar4[h*w]; nullptr renders into renderState->raycastImage
  void RenderImage(const ITMPose* pose, const ITMIntrinsics* intrinsics, const ITMRenderState* renderState, uint8_t* outputImage,
                   RenderImageType type = RENDER_SHADED_GREYSCALE) const {
    check(itm_render_image(scene->handle, pose->GetM(), intrinsics->all, renderState->handle, outputImage, (int)type, stream), "RenderImage");
  }
  void FindSurface(const ITMPose* pose, const ITMIntrinsics* intrinsics, const ITMRenderState* renderState) const {
    check(itm_find_surface(scene->handle, pose->GetM(), intrinsics->all, renderState->handle, stream), "FindSurface");
  }
  void CreatePointCloud(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState, bool skipPoints) const {
    itm_view v = make_view(view, trackingState);
    check(itm_create_point_cloud(scene->handle, &v, renderState->handle, skipPoints ? 1 : 0, trackingState->pointCloud_locations,
                                 trackingState->pointCloud_colours, stream), "CreatePointCloud");
    trackingState->pose_pointCloud = trackingState->pose_d;
  }
  void CreateICPMaps(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState) const {
    itm_view v = make_view(view, trackingState);
    check(itm_create_icp_maps(scene->handle, &v, renderState->handle, trackingState->pointCloud_locations,
                              trackingState->pointCloud_colours, stream), "CreateICPMaps");
    trackingState->pose_pointCloud = trackingState->pose_d;   // ITMVisualisationEngine_CPU.cpp:272
  }
  void ForwardRender(const ITMView* view, ITMTrackingState* trackingState, ITMRenderState* renderState) const {
    itm_view v = make_view(view, trackingState);
    check(itm_forward_render(scene->handle, &v, renderState->handle, stream), "ForwardRender");
  }
};

// The callers of the path, same call order and state machine as
// ITMDenseMapper::ProcessFrame / UpdateVisibleList (Engine/ITMDenseMapper.cpp:50-71) and
// ITMTrackingController::Prepare (Engine/ITMTrackingController.cpp:18-46, non-colour trackers).
// ITMSwappingEngine<TVoxel, TIndex> (Engine/ITMSwappingEngine.h:19-36)
template <class TVoxel, class TIndex>
class ITMSwappingEngine_HIP {
 public:
  itm_stream stream = nullptr;
  void IntegrateGlobalIntoLocal(ITMScene<TVoxel, TIndex>* scene, ITMRenderState* renderState) {
    check(itm_swap_integrate_global_into_local(scene->handle, renderState->handle, stream), "IntegrateGlobalIntoLocal");
  }
  void SaveToGlobalMemory(ITMScene<TVoxel, TIndex>* scene, ITMRenderState* renderState) {
    check(itm_swap_save_to_global_memory(scene->handle, renderState->handle, stream), "SaveToGlobalMemory");
  }
};

template <class TVoxel, class TIndex>
class ITMDenseMapper_HIP {
  ITMSceneReconstructionEngine_HIP<TVoxel, TIndex> reco;
  ITMSwappingEngine_HIP<TVoxel, TIndex> swappingEngine;

 public:
  void SetStream(itm_stream s) { reco.stream = s; swappingEngine.stream = s; }
  void ResetScene(ITMScene<TVoxel, TIndex>* scene) { reco.ResetScene(scene); }
  void ProcessFrame(const ITMView* view, const ITMTrackingState* ts, ITMScene<TVoxel, TIndex>* scene, ITMRenderState* rs) {
    reco.AllocateSceneFromDepth(scene, view, ts, rs);
    reco.IntegrateIntoScene(scene, view, ts, rs);
    if (scene->useSwapping) {                     // ITMDenseMapper.cpp:59-64 (settings->useSwapping)
      swappingEngine.IntegrateGlobalIntoLocal(scene, rs);     // swapping: host -> device
      swappingEngine.SaveToGlobalMemory(scene, rs);           // swapping: device -> host
    }
  }
  void UpdateVisibleList(const ITMView* view, const ITMTrackingState* ts, ITMScene<TVoxel, TIndex>* scene, ITMRenderState* rs) {
    reco.AllocateSceneFromDepth(scene, view, ts, rs, true);
  }
};

// ITMViewBuilder (Engine/ITMViewBuilder.h:17-60): raw depth frame (device short image) -> ITMView::depth in metres,
// optional 5-pass bilateral filter and normal / uncertainty images.  The caller owns the device images.
class ITMViewBuilder_HIP {
  const ITMRGBDCalib* calib;
  int calibType; float c0, c1;

 public:
  itm_stream stream = nullptr;
  // calibType: 0 = ITMDisparityCalib::TRAFO_KINECT (params c0, c1), 1 = TRAFO_AFFINE (depth = raw * c0 + c1)
  ITMViewBuilder_HIP(const ITMRGBDCalib* calib_, int calibType_, float c0_, float c1_) : calib(calib_), calibType(calibType_), c0(c0_), c1(c1_) {}
  void ConvertDisparityToDepth(float* depth_out, const int16_t* disp_in, Vector2i size) {
    check(itm_convert_disparity(disp_in, depth_out, size.x, size.y, c0, c1, calib->intrinsics_d.all[0], stream), "ConvertDisparityToDepth");
  }
  void ConvertDepthAffineToFloat(float* depth_out, const int16_t* depth_in, Vector2i size) {
    check(itm_convert_depth_affine(depth_in, depth_out, size.x, size.y, c0, c1, stream), "ConvertDepthAffineToFloat");
  }
  void DepthFiltering(float* image_out, const float* image_in, Vector2i size) {
    check(itm_filter_depth(image_in, image_out, size.x, size.y, stream), "DepthFiltering");
  }
  void ComputeNormalAndWeights(float* normal_out, float* sigmaZ_out, const float* depth_in, Vector2i size) {
    check(itm_compute_normal_and_weights(depth_in, normal_out, sigmaZ_out, size.x, size.y, calib->intrinsics_d.all, stream), "ComputeNormalAndWeights");
  }
  // UpdateView(view, rgb, rawDepth, useBilateralFilter, modelSensorNoise): fills view->depth (and the optional images)
  void UpdateView(ITMView* view, const int16_t* rawDepth, float* depth, float* scratch, bool useBilateralFilter,
                  bool modelSensorNoise = false, float* depthNormal = nullptr, float* depthUncertainty = nullptr) {
    check(itm_update_view(rawDepth, view->depthSize.x, view->depthSize.y, calibType, c0, c1, calib->intrinsics_d.all, useBilateralFilter ? 1 : 0,
                          modelSensorNoise ? 1 : 0, depth, scratch, depthNormal, depthUncertainty, stream), "UpdateView");
    view->depth = depth;
  }
  // The reference's UpdateView takes HOST images and opens with a synchronous copy (shortImage->SetFrom(rawDepthImage, CPU_TO_CUDA),
  // DeviceSpecific/CUDA/ITMViewBuilder_CUDA.cu:53).  Here a raw frame in pinned host memory travels on the stager's copy stream;
  // Prefetch(next frame) while the current one is fused hides the transfer altogether.
  void Prefetch(const int16_t* rawDepthHost, Vector2i size) {
    if (!stager) check(itm_depth_stager_create(size.x, size.y, 4, &stager), "depth stager");
    check(itm_depth_stager_upload(stager, rawDepthHost), "Prefetch");
    prefetched = rawDepthHost;
  }
  void UpdateViewFromHost(ITMView* view, const int16_t* rawDepthHost, float* depth, float* scratch, bool useBilateralFilter,
                          bool modelSensorNoise = false, float* depthNormal = nullptr, float* depthUncertainty = nullptr) {
    // the slot of the PREVIOUS frame: everything that read its depth has been submitted to the stream by now
    if (holding) { check(itm_depth_stager_release(stager, stream), "UpdateView (release)"); holding = false; }
    // without filter and noise model the float depth is the conversion alone: the stager's copy does it (itm_depth_stager_set_conversion)
    // and the view's depth is the slot's image -- no conversion launch on the frame's stream
    const bool plain = !useBilateralFilter && !modelSensorNoise;
    auto discard_waiting = [&]() {
      int waiting = 0;
      check(itm_depth_stager_pending(stager, &waiting, nullptr), "UpdateView (pending)");
      for (; waiting > 0; --waiting) {
        const int16_t* stale = nullptr;
        check(itm_depth_stager_acquire(stager, stream, &stale), "UpdateView (discard)");
        check(itm_depth_stager_release(stager, stream), "UpdateView (discard)");
      }
      prefetched = nullptr;
    };
    if (!stager) check(itm_depth_stager_create(view->depthSize.x, view->depthSize.y, 4, &stager), "depth stager");
    // frames uploaded ahead that are not the one asked for (the image source changed its mind), or uploaded in the other form: taken off
    // the ring unread, else the stager -- strictly first in, first out -- would hand the stale frame to this call and stay one frame behind
    if (prefetched != rawDepthHost || plain != converting) discard_waiting();
    if (plain != converting) {
      if (plain) check(itm_depth_stager_set_conversion(stager, calibType, c0, c1, calib->intrinsics_d.all[0]), "depth stager (conversion)");
      converting = plain;
    }
    if (prefetched != rawDepthHost) Prefetch(rawDepthHost, view->depthSize);
    prefetched = nullptr;
    if (plain) {
      const float* converted = nullptr;
      check(itm_depth_stager_acquire_depth(stager, stream, nullptr, &converted), "UpdateView (acquire)");
      view->depth = converted;
      holding = true;           // released by the next call: the frame's launches read it
      return;
    }
    const int16_t* raw = nullptr;
    check(itm_depth_stager_acquire(stager, stream, &raw), "UpdateView (acquire)");
    UpdateView(view, raw, depth, scratch, useBilateralFilter, modelSensorNoise, depthNormal, depthUncertainty);
    check(itm_depth_stager_release(stager, stream), "UpdateView (release)");
  }
  // true once the copy stream has READ every host frame handed to Prefetch / UpdateViewFromHost so far: from then on the image source
  // may rewrite those buffers (the uploads are asynchronous; a source that reuses ONE raw buffer asks -- or waits -- before it refills it)
  bool HostFramesRead() const {
    int busy = 0;
    if (stager) check(itm_depth_stager_pending(stager, nullptr, &busy), "HostFramesRead");
    return busy == 0;
  }
  void WaitHostFramesRead() const { while (!HostFramesRead()) {} }
  ~ITMViewBuilder_HIP() { if (stager) itm_depth_stager_destroy(stager); }
  ITMViewBuilder_HIP(const ITMViewBuilder_HIP&) = delete;
  ITMViewBuilder_HIP& operator=(const ITMViewBuilder_HIP&) = delete;

 private:
  itm_depth_stager* stager = nullptr;
  const int16_t* prefetched = nullptr;
  bool converting = false, holding = false;
};

// ITMDepthTracker (Engine/ITMDepthTracker.h): TrackCamera refines trackingState->pose_d against the ICP maps
// of the previous frame; the gradient / Hessian reduction and the depth pyramid run on the GPU.
class ITMDepthTracker_HIP {
  itm_tracker_config cfg;
  itm_tracker* tracker = nullptr;     // this object's hierarchy + reduction buffers (ITMDepthTracker.cpp:18-44); one per tracker object

 public:
  itm_stream stream = nullptr;
  ITMDepthTracker_HIP(const int* trackingRegime, int noHierarchyLevels, int noICPRunTillLevel, float distThresh, float terminationThreshold) {
    std::memset(&cfg, 0, sizeof cfg);
    cfg.noHierarchyLevels = noHierarchyLevels;
    for (int i = 0; i < noHierarchyLevels && i < 8; ++i) cfg.trackingRegime[i] = trackingRegime[i];
    cfg.noICPRunTillLevel = noICPRunTillLevel; cfg.distThresh = distThresh; cfg.terminationThreshold = terminationThreshold;
    check(itm_tracker_create(&tracker), "itm_tracker_create");
  }
  ~ITMDepthTracker_HIP() { itm_tracker_destroy(tracker); }
  ITMDepthTracker_HIP(const ITMDepthTracker_HIP&) = delete;
  ITMDepthTracker_HIP& operator=(const ITMDepthTracker_HIP&) = delete;
  void TrackCamera(ITMTrackingState* trackingState, const ITMView* view) {
    itm_view v = make_view(view, trackingState);
    float M[16];
    check(itm_tracker_track_camera(tracker, &cfg, &v, trackingState->pointCloud_locations, trackingState->pointCloud_colours,
                                   trackingState->pose_pointCloud.GetM(), M, stream), "TrackCamera");
    trackingState->pose_d.SetM(M);
  }
};

// ITMTracker (Engine/ITMTracker.h): what ITMTrackingController::Track calls
class ITMTracker {
 public:
  virtual void TrackCamera(ITMTrackingState* trackingState, const ITMView* view) = 0;
  virtual ~ITMTracker() {}
};
// ITMExternalTracker of this fork (Engine/ITMExternalTracker.cpp:27-30): the pose was put into trackingState->pose_d from outside
class ITMExternalTracker : public ITMTracker {
 public:
  void TrackCamera(ITMTrackingState*, const ITMView*) override {}
};
// adapter: ITMDepthTracker_HIP behind the ITMTracker interface
class ITMDepthTrackerAdapter : public ITMTracker {
  ITMDepthTracker_HIP* t;
 public:
  explicit ITMDepthTrackerAdapter(ITMDepthTracker_HIP* t_) : t(t_) {}
  void TrackCamera(ITMTrackingState* ts, const ITMView* view) override { t->TrackCamera(ts, view); }
};

// 4x4 product as ORUtils/Matrix.h:96-104 forms it (column-major, r(x, y) accumulated from zero over k)
inline void matmul4(const float* lhs, const float* rhs, float* out) {
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) {
    float r = 0.0f;
    for (int k = 0; k < 4; ++k) r += lhs[k * 4 + y] * rhs[x * 4 + k];
    out[x * 4 + y] = r;
  }
}

// ITMTrackingController (Engine/ITMTrackingController.cpp:11-46): Track and Prepare, statement for statement
template <class TVoxel, class TIndex>
class ITMTrackingController_HIP {
  ITMTracker* tracker;
  const ITMVisualisationEngine_HIP<TVoxel, TIndex>* visualisationEngine;
  const ITMLibSettings* settings;
  ITMLibSettings ownSettings;

 public:
  ITMTrackingController_HIP(ITMTracker* tracker_, const ITMVisualisationEngine_HIP<TVoxel, TIndex>* vis, const ITMLibSettings* settings_)
      : tracker(tracker_), visualisationEngine(vis), settings(settings_) {}
  // external poses (no tracker object), `approximate` = ITMLibSettings::useApproximateRaycast
  explicit ITMTrackingController_HIP(const ITMVisualisationEngine_HIP<TVoxel, TIndex>* vis, bool approximate = false)
      : tracker(nullptr), visualisationEngine(vis), settings(&ownSettings) { ownSettings.useApproximateRaycast = approximate; }

  void Track(ITMTrackingState* trackingState, const ITMView* view) {
    if (trackingState->age_pointCloud != -1 && tracker) tracker->TrackCamera(trackingState, view);
    trackingState->requiresFullRendering = trackingState->TrackerFarFromPointCloud() || !settings->useApproximateRaycast;
  }
  void Prepare(ITMTrackingState* trackingState, const ITMView* view, ITMRenderState* renderState) {
    if (settings->trackerType == ITMLibSettings::TRACKER_COLOR) {
      ITMPose pose_rgb;
      float M[16];
      matmul4(view->calib.trafo_rgb_to_depth_calib_inv, trackingState->pose_d.GetM(), M);
      pose_rgb.SetM(M);
      visualisationEngine->CreateExpectedDepths(&pose_rgb, &view->calib.intrinsics_rgb, renderState);
      visualisationEngine->CreatePointCloud(view, trackingState, renderState, settings->skipPoints);
      trackingState->age_pointCloud = 0;
    } else {
      visualisationEngine->CreateExpectedDepths(&trackingState->pose_d, &view->calib.intrinsics_d, renderState);
      if (trackingState->requiresFullRendering) {
        visualisationEngine->CreateICPMaps(view, trackingState, renderState);
        trackingState->pose_pointCloud = trackingState->pose_d;
        if (trackingState->age_pointCloud == -1) trackingState->age_pointCloud = -2;
        else trackingState->age_pointCloud = 0;
      } else {
        visualisationEngine->ForwardRender(view, trackingState, renderState);
        trackingState->age_pointCloud++;
      }
    }
  }
};

// ITMMainEngine (Engine/ITMMainEngine.cpp:7-127,194-197) for raw frames already in device memory: builds the view, tracks, fuses
// unless integration is switched off, and prepares the maps for the next frame -- with the reference's switches.  Owns the scene,
// the engines, the live render state, the tracking state and the view's depth images.
template <class TVoxel, class TIndex>
class ITMMainEngine_HIP {
  ITMLibSettings settings;
  ITMSceneParams sceneParams;
  ITMScene<TVoxel, TIndex> scene;
  ITMDenseMapper_HIP<TVoxel, TIndex> denseMapper;
  ITMVisualisationEngine_HIP<TVoxel, TIndex> visualisationEngine;
  ITMDepthTracker_HIP* depthTracker = nullptr;
  ITMTracker* tracker = nullptr;
  ITMTrackingController_HIP<TVoxel, TIndex>* trackingController = nullptr;
  ITMViewBuilder_HIP* viewBuilder = nullptr;
  ITMRenderState* renderState_live = nullptr;
  ITMTrackingState trackingState;
  ITMView view;
  void *depthBuf = nullptr, *scratchBuf = nullptr, *normalBuf = nullptr, *sigmaBuf = nullptr, *pointsBuf = nullptr, *coloursBuf = nullptr;
  bool fusionActive = true, mainProcessingActive = true;

 public:
  // calibType / c0 / c1: ITMDisparityCalib (0 = TRAFO_KINECT, 1 = TRAFO_AFFINE); sizes as ITMMainEngine's imgSize_rgb / imgSize_d
  ITMMainEngine_HIP(const ITMLibSettings& settings_, const ITMSceneParams& params, const ITMRGBDCalib& calib, Vector2i imgSize_rgb, Vector2i imgSize_d,
                    int calibType = 1, float c0 = 0.001f, float c1 = 0.0f, int localBlockNum = 0)
      : settings(settings_), sceneParams(params), scene(&sceneParams, localBlockNum), visualisationEngine(&scene) {
    view.calib = calib;
    view.depthSize = imgSize_d; view.rgbSize = imgSize_rgb;
    const size_t P = (size_t)imgSize_d.x * imgSize_d.y;
    // the tracked image: rgb for the colour tracker, depth otherwise (ITMTrackingController::GetTrackedImageSize)
    const Vector2i tracked = settings.trackerType == ITMLibSettings::TRACKER_COLOR ? imgSize_rgb : imgSize_d;
    const size_t PT = (size_t)tracked.x * tracked.y;
    check(itm_dev_malloc(&depthBuf, P * 4), "malloc"); check(itm_dev_malloc(&scratchBuf, P * 4), "malloc");
    check(itm_dev_malloc(&normalBuf, P * 16), "malloc"); check(itm_dev_malloc(&sigmaBuf, P * 4), "malloc");
    check(itm_dev_malloc(&pointsBuf, PT * 16), "malloc"); check(itm_dev_malloc(&coloursBuf, PT * 16), "malloc");
    trackingState.pointCloud_locations = (float*)pointsBuf; trackingState.pointCloud_colours = (float*)coloursBuf;
    denseMapper.ResetScene(&scene);
    viewBuilder = new ITMViewBuilder_HIP(&view.calib, calibType, c0, c1);
    if (settings.trackerType == ITMLibSettings::TRACKER_ICP) {
      depthTracker = new ITMDepthTracker_HIP(settings.trackingRegime, settings.noHierarchyLevels, settings.noICPRunTillLevel,
                                             settings.depthTrackerICPThreshold, settings.depthTrackerTerminationThreshold);
      tracker = new ITMDepthTrackerAdapter(depthTracker);
    } else {
      tracker = new ITMExternalTracker();          // TRACKER_EXTERNAL, and TRACKER_COLOR with poses from outside (the colour tracker itself is not part of the path)
    }
    trackingController = new ITMTrackingController_HIP<TVoxel, TIndex>(tracker, &visualisationEngine, &settings);
    renderState_live = visualisationEngine.CreateRenderState(tracked);
  }
  ~ITMMainEngine_HIP() {
    delete renderState_live; delete trackingController; delete tracker; delete depthTracker; delete viewBuilder;
    for (void* p : {depthBuf, scratchBuf, normalBuf, sigmaBuf, pointsBuf, coloursBuf}) itm_dev_free(p);
  }
  ITMMainEngine_HIP(const ITMMainEngine_HIP&) = delete;
  ITMMainEngine_HIP& operator=(const ITMMainEngine_HIP&) = delete;

  ITMView* GetView() { return &view; }
  ITMTrackingState* GetTrackingState() { return &trackingState; }
  ITMScene<TVoxel, TIndex>* GetScene() { return &scene; }
  ITMRenderState* GetRenderState() { return renderState_live; }
  const ITMVisualisationEngine_HIP<TVoxel, TIndex>* GetVisualisationEngine() const { return &visualisationEngine; }
  const ITMViewBuilder_HIP* GetViewBuilder() const { return viewBuilder; }

  // rgbImage: device uchar4 (may be null without colour), rawDepthImage: device short
  void ProcessFrame(const uint8_t* rgbImage, const int16_t* rawDepthImage) {
    // prepare image and turn it into a depth image
    view.rgb = rgbImage;
    viewBuilder->UpdateView(&view, rawDepthImage, (float*)depthBuf, (float*)scratchBuf, settings.useBilateralFilter, settings.modelSensorNoise,
                            (float*)normalBuf, (float*)sigmaBuf);
    ProcessView();
  }
  // The reference's own signature takes the raw frame in HOST memory (Engine/ITMMainEngine.cpp:111): rawDepthHost in page-locked memory
  // (itm_host_malloc); nextRawDepthHost, when the image source already has it, is uploaded while this frame is tracked and fused.
  // The uploads are asynchronous: a buffer handed over here may be REWRITTEN only once GetViewBuilder()->HostFramesRead() says so
  // (WaitHostFramesRead() waits; ~25 us after the call for a frame that was not announced).  The reference's CUDA build blocks in the
  // copy instead (ITMViewBuilder_CUDA.cu:53); an image source with two buffers never has to wait here.
  void ProcessFrameFromHost(const uint8_t* rgbImage, const int16_t* rawDepthHost, const int16_t* nextRawDepthHost = nullptr) {
    view.rgb = rgbImage;
    viewBuilder->UpdateViewFromHost(&view, rawDepthHost, (float*)depthBuf, (float*)scratchBuf, settings.useBilateralFilter, settings.modelSensorNoise,
                                    (float*)normalBuf, (float*)sigmaBuf);
    if (nextRawDepthHost) viewBuilder->Prefetch(nextRawDepthHost, view.depthSize);
    ProcessView();
  }

 private:
  void ProcessView() {
    if (!mainProcessingActive) return;
    // pose of this frame against the maps of the last ray cast
    trackingController->Track(&trackingState, &view);
    // allocate + integrate (the library records both; they launch with the two calls of Prepare as ONE fused frame, pending.hip)
    if (fusionActive) denseMapper.ProcessFrame(&view, &trackingState, &scene, renderState_live);
    // expected depths + ICP maps (or the forward projection) from the new pose: what the next Track call and the UI read
    trackingController->Prepare(&trackingState, &view, renderState_live);
  }

 public:
  void turnOnIntegration() { fusionActive = true; }
  void turnOffIntegration() { fusionActive = false; }
  void turnOnMainProcessing() { mainProcessingActive = true; }
  void turnOffMainProcessing() { mainProcessingActive = false; }
};

// ITMMesh (Objects/ITMMesh.h:14-124): the triangle buffer lives in HBM; WriteOBJ / WriteSTL produce the reference's files.
class ITMMesh {
 public:
  itm_mesh* handle = nullptr;
  uint32_t noTotalTriangles = 0, noMaxTriangles = 0;
  itm_stream stream = nullptr;
  template <class TVoxel, class TIndex>
  explicit ITMMesh(const ITMScene<TVoxel, TIndex>* scene, uint32_t maxTriangles = 0) {
    check(itm_mesh_create(scene->handle, maxTriangles, &handle), "itm_mesh_create");
    check(itm_mesh_info(handle, nullptr, &noMaxTriangles, nullptr, nullptr), "itm_mesh_info");
  }
  ~ITMMesh() { itm_mesh_destroy(handle); }
  ITMMesh(const ITMMesh&) = delete;
  ITMMesh& operator=(const ITMMesh&) = delete;
  const float* triangles() const { const float* p = nullptr; check(itm_mesh_info(handle, nullptr, nullptr, &p, stream), "itm_mesh_info"); return p; }   // device pointer
  void WriteOBJ(const char* fileName) const { check(itm_mesh_write_obj(handle, fileName, stream), "WriteOBJ"); }
  void WriteSTL(const char* fileName) const { check(itm_mesh_write_stl(handle, fileName, stream), "WriteSTL"); }
};

// ITMMeshingEngine<TVoxel,TIndex>::MeshScene (Engine/ITMMeshingEngine.h:19-26)
template <class TVoxel, class TIndex>
class ITMMeshingEngine_HIP {
 public:
  itm_stream stream = nullptr;
  void MeshScene(ITMMesh* mesh, const ITMScene<TVoxel, TIndex>* scene) {
    check(itm_mesh_scene(scene->handle, mesh->handle, stream), "MeshScene");
    check(itm_mesh_info(mesh->handle, &mesh->noTotalTriangles, nullptr, nullptr, stream), "itm_mesh_info");
  }
};

// One rank's end of the multi-stream exchange: the record of this stream goes out with every frame, the table of all
// streams comes back every `batch` frames.  Rank 0 creates the id, the host distributes it.
class ITMStreamExchange_HIP {
  itm_exchange* handle = nullptr;
  int world, maxIds, batch;

 public:
  static void UniqueId(unsigned char id[128]) { check(itm_exchange_unique_id(id), "itm_exchange_unique_id"); }
  ITMStreamExchange_HIP(int world_, int rank, const unsigned char id[128], int maxIds_ = 16384, int batch_ = 8) : world(world_), maxIds(maxIds_), batch(batch_) {
    check(itm_exchange_create(world, rank, id, maxIds, batch, &handle), "itm_exchange_create");
  }
  ~ITMStreamExchange_HIP() { itm_exchange_destroy(handle); }
  ITMStreamExchange_HIP(const ITMStreamExchange_HIP&) = delete;
  ITMStreamExchange_HIP& operator=(const ITMStreamExchange_HIP&) = delete;
  void Publish(const ITMRenderState* renderState, const ITMTrackingState* trackingState, itm_stream stream = nullptr) {
    check(itm_exchange_step(handle, renderState->handle, trackingState->pose_d.GetM(), stream), "itm_exchange_step");
  }
  size_t TableWords() const { return (size_t)world * (size_t)batch * (size_t)(17 + maxIds); }
  void Table(int32_t* hostTable) { check(itm_exchange_table(handle, hostTable, TableWords()), "itm_exchange_table"); }
};

}  // namespace itmhip
