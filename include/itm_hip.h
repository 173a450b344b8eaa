/*
 * itm_hip.h -- C-ABI of the MI355X-native TSDF allocate / integrate / raycast path.
 *
 * This is the drop-in boundary: one flat `extern "C"` surface (plain pointers, sizes and POD
 * structs, no C++/torch types) that a fourth InfiniTAM back-end ("HIP", next to CPU/CUDA/Metal in
 * InfiniTAM/ITMLib/Engine/DeviceSpecific/) binds to.  Every entry point names the reference
 * method it replaces (paths relative to /root/reference/InfiniTAM/ITMLib).
 *
 * Conventions
 *  - All image / map pointers are DEVICE pointers for libitmhip.so (HBM), valid on `stream`'s
 *    device.  Nothing is synchronised implicitly except itm_get_counters / itm_download_* /
 *    itm_stream_synchronize; every other call only enqueues work on `stream` (a hipStream_t
 *    passed as void*; NULL = the default stream).
 *  - Matrices are 4x4 float, column-major `m[col*4+row]` exactly as ORUtils::Matrix4
 *    (ORUtils/Matrix.h:23-33,78); poses are world->camera.
 *  - Intrinsics are (fx, fy, cx, cy) = ITMIntrinsics::projectionParamsSimple.all
 *    (Objects/ITMIntrinsics.h:20-40).
 *  - Memory layouts of ITMHashEntry, ITMVoxel_{s,f,s_rgb,f_rgb}, Vector2f/4f/4u images are
 *    byte-identical to the reference (Utils/ITMLibDefines.h:71-199), so dumps compare 1:1.
 *  - Return value: 0 = ok, negative = ITM_ERR_*.  The reference's engines are `void` and
 *    exit(-1) on device errors (ORUtils/CUDADefines.h:27-36); the C++ adapter maps non-zero to
 *    std::runtime_error.  Pool exhaustion is NOT an error: blocks are silently skipped as in
 *    DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:189,206.
 *
 * The same header is compiled with a different ITM_FN prefix by the test oracle (oracle/) so that
 * the parity tests can drive the oracle and the product through one binding; the product library
 * exports the `itm_` names only.
 */
#ifndef ITM_HIP_H_
#define ITM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifndef ITM_FN
#define ITM_FN(name) itm_##name
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes -------------------------------------------------------------------------- */
#define ITM_OK 0
#define ITM_ERR_INVALID -1      /* bad argument / unsupported configuration            */
#define ITM_ERR_DEVICE -2       /* HIP runtime error (message via itm_last_error)      */
#define ITM_ERR_UNSUPPORTED -3  /* valid in the reference, not available in this build */

/* ---- compile-time constants of the reference kept as defaults ---------------------------- */
#define ITM_SDF_BLOCK_SIZE 8            /* Utils/ITMLibDefines.h:37 */
#define ITM_SDF_BLOCK_SIZE3 512         /* :38 */
#define ITM_DEFAULT_LOCAL_BLOCK_NUM 0x10000 /* :40  (fork value; upstream 0x40000)     */
#define ITM_DEFAULT_BUCKET_NUM 0x100000 /* :51 */
#define ITM_DEFAULT_EXCESS_NUM 0x20000  /* :60 */
#define ITM_MAX_RENDERING_BLOCKS (65536 * 4) /* DeviceAgnostic/ITMVisualisationEngine.h:24 */

/* TVoxel / TIndex template parameters of the reference become runtime enums. */
enum itm_voxel_type {
  ITM_VOXEL_S = 0,     /* ITMVoxel_s      4 B  {i16 sdf; u8 w_depth; pad}           :153-174 */
  ITM_VOXEL_F = 1,     /* ITMVoxel_f      8 B  {f32 sdf; u8 w_depth; pad[3]}        :176-197 */
  ITM_VOXEL_S_RGB = 2, /* ITMVoxel_s_rgb  8 B  {i16 sdf; u8 w; u8 clr[3]; u8 w_color; pad}   */
  ITM_VOXEL_F_RGB = 3  /* ITMVoxel_f_rgb 12 B  {f32 sdf; u8 w; u8 clr[3]; u8 w_color; pad[3]} */
};
enum itm_index_type {
  ITM_INDEX_HASH = 0,  /* ITMVoxelBlockHash   Objects/ITMVoxelBlockHash.h:22-100  */
  ITM_INDEX_DENSE = 1  /* ITMPlainVoxelArray  Objects/ITMPlainVoxelArray.h:18-75  */
};
/* IITMVisualisationEngine::RenderImageType, Engine/ITMVisualisationEngine.h:21-26 */
enum itm_render_type {
  ITM_RENDER_SHADED_GREYSCALE = 0,
  ITM_RENDER_COLOUR_FROM_VOLUME = 1,
  ITM_RENDER_COLOUR_FROM_NORMAL = 2
};

/* ITMSceneParams, Objects/ITMSceneParams.h:14-70 (defaults Utils/ITMLibSettings.cpp:10). */
typedef struct itm_scene_params {
  float voxelSize;
  float mu;
  int32_t maxW;
  float viewFrustum_min;
  float viewFrustum_max;
  int32_t stopIntegratingAtMaxW;
} itm_scene_params;

/* What `new ITMScene<TVoxel,TIndex>(params, useSwapping=false, memoryType)` fixes at compile
 * time in the reference (Objects/ITMScene.h:37-43, ITMVoxelBlockHash.h:35-36,
 * ITMPlainVoxelArray.h:24-37) is a runtime configuration here.  0 = reference default. */
typedef struct itm_scene_config {
  int32_t voxelType;      /* enum itm_voxel_type */
  int32_t indexType;      /* enum itm_index_type */
  int32_t bucketNum;      /* SDF_BUCKET_NUM, power of two   */
  int32_t excessNum;      /* SDF_EXCESS_LIST_SIZE           */
  int32_t localBlockNum;  /* SDF_LOCAL_BLOCK_NUM            */
  int32_t denseSize[3];   /* ITMVoxelArrayInfo::size   (default 512^3)          */
  int32_t denseOffset[3]; /* ITMVoxelArrayInfo::offset (default -256,-256,0)    */
  int32_t denseOffsetSet; /* non-zero: use denseOffset even if it is all zero   */
  int32_t maxRenderingBlocks; /* MAX_RENDERING_BLOCKS (DeviceAgnostic/ITMVisualisationEngine.h:24);
                                 0 = 262144.  Runtime so that the cap path can be tested.       */
  int32_t useSwapping;        /* ITMScene(..., useSwapping, ...) Objects/ITMScene.h:37-43: the scene owns an ITMGlobalCache in host
                                 memory and AllocateSceneFromDepth keeps the swap states (hash index only)                          */
  int32_t transferBlockNum;   /* SDF_TRANSFER_BLOCK_NUM (Utils/ITMLibDefines.h): blocks moved per swapping call; 0 = 0x1000        */
} itm_scene_config;

/* The inputs an engine method takes from `const ITMView*` and `const ITMTrackingState*`:
 * view->depth, view->rgb, view->calib->{intrinsics_d, intrinsics_rgb, trafo_rgb_to_depth} and
 * trackingState->pose_d->GetM()  (Objects/ITMView.h:16-52, Objects/ITMTrackingState.h:18-75). */
typedef struct itm_view {
  const float* depth;      /* float[h*w] metres, <=0 invalid (ITMViewBuilder output)   */
  const uint8_t* rgb;      /* uchar4[h_rgb*w_rgb] RGBA; may be NULL for colourless voxels */
  int32_t w, h;            /* view->depth->noDims */
  int32_t w_rgb, h_rgb;    /* view->rgb->noDims   */
  float M_d[16];           /* trackingState->pose_d->GetM()             */
  float intr_d[4];         /* calib->intrinsics_d.projectionParamsSimple.all   */
  float intr_rgb[4];       /* calib->intrinsics_rgb.projectionParamsSimple.all */
  float rgb_to_depth[16];     /* calib->trafo_rgb_to_depth.calib      (Objects/ITMExtrinsics.h) */
  float rgb_to_depth_inv[16]; /* calib->trafo_rgb_to_depth.calib_inv                            */
} itm_view;

/* Host-visible scalars the reference keeps as members: ITMLocalVBA::lastFreeBlockId,
 * ITMVoxelBlockHash::lastFreeExcessListId, ITMRenderState_VH::noVisibleEntries,
 * ITMRenderState::noFwdProjMissingPoints, ITMPointCloud::noTotalPoints. */
typedef struct itm_counters {
  int32_t lastFreeBlockId;
  int32_t lastFreeExcessListId;
  int32_t noVisibleEntries;
  int32_t noFwdProjMissingPoints;
  int32_t noTotalPoints;
  int32_t noRenderingBlocks;   /* numRenderingBlocks of the last CreateExpectedDepths */
  int32_t noAllocRequests;     /* blocks requested by the last AllocateSceneFromDepth */
  int32_t statusFlags;         /* FATAL conditions: the scene is no longer what the reference would hold.  bit0: a depth pixel needed more ray
                                  steps than the allocation key can number (mu / voxelSize beyond ~2 000), blocks were not requested;
                                  bit1: a bounded wait between workgroups of the visible-list launch expired (a device that does not
                                  run what it was given), the frame was not fused.  Once raised, every call that names the scene
                                  returns ITM_ERR_DEVICE (itm_last_error says which) until itm_reset_scene; itm_get_counters still
                                  fills its output before it returns the error.  Never raised by a healthy frame. */
} itm_counters;

/* buffers addressable by itm_download / itm_upload (parity dumps, checkpoints) */
enum itm_buffer {
  ITM_BUF_HASH_ENTRIES = 0,     /* ITMHashEntry[bucketNum+excessNum]   scene->index.GetEntries() */
  ITM_BUF_EXCESS_LIST = 1,      /* int[excessNum]                      GetExcessAllocationList() */
  ITM_BUF_VOXEL_BLOCKS = 2,     /* TVoxel[localBlockNum*512 | sx*sy*sz] localVBA.GetVoxelBlocks() */
  ITM_BUF_ALLOCATION_LIST = 3,  /* int[localBlockNum]                  localVBA.GetAllocationList() */
  ITM_BUF_VISIBLE_IDS = 4,      /* int[localBlockNum]   renderState_vh->GetVisibleEntryIDs()   */
  ITM_BUF_VISIBLE_TYPE = 5,     /* uchar[noTotalEntries] renderState_vh->GetEntriesVisibleType() */
  ITM_BUF_RANGE_IMAGE = 6,      /* Vector2f[h*w]  renderState->renderingRangeImage  */
  ITM_BUF_RAYCAST_RESULT = 7,   /* Vector4f[h*w]  renderState->raycastResult        */
  ITM_BUF_RAYCAST_IMAGE = 8,    /* Vector4u[h*w]  renderState->raycastImage         */
  ITM_BUF_FORWARD_PROJECTION = 9, /* Vector4f[h*w] renderState->forwardProjection   */
  ITM_BUF_MISSING_POINTS = 10,  /* int[h*w]       renderState->fwdProjMissingPoints */
  ITM_BUF_SWAP_STATES = 11      /* uchar[noTotalEntries] globalCache->GetSwapStates(): ITMHashSwapState::state (scenes with useSwapping) */
};

typedef struct itm_scene itm_scene;               /* ITMScene<TVoxel,TIndex> + engine scratch */
typedef struct itm_render_state itm_render_state; /* ITMRenderState / ITMRenderState_VH       */
typedef void* itm_stream;                         /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------ */
const char* ITM_FN(version)(void);
const char* ITM_FN(last_error)(void);
/* 1 for the product (device pointers), 0 for a host-memory implementation of this ABI. */
int ITM_FN(uses_device_memory)(void);
size_t ITM_FN(voxel_size_bytes)(int voxelType);

/* ---- device memory helpers (so a C / ctypes caller needs no other runtime) ---------------- */
int ITM_FN(dev_malloc)(void** ptr, size_t bytes);
int ITM_FN(dev_free)(void* ptr);
/* page-locked host memory (hipHostMalloc): the source of asynchronous uploads, see itm_depth_stager */
int ITM_FN(host_malloc)(void** ptr, size_t bytes);
int ITM_FN(host_free)(void* ptr);
/* Page-locks host memory the HOST owns (hipHostRegister): an image buffer of the reference (ORUtils::MemoryBlock allocates with `new`)
 * becomes a source / target of asynchronous copies at PCIe speed -- from pageable memory itm_memcpy_h2d is staged through the runtime's
 * bounce buffer and blocks the calling thread.  Registering a range that is registered already is not an error but is TOLD: the call
 * returns ITM_ALREADY_REGISTERED (> 0) and the caller does not own that registration -- it must not unregister it (someone else did the
 * registering, possibly for a different length).  Unregister what you registered before the memory is freed. */
#define ITM_ALREADY_REGISTERED 1
int ITM_FN(host_register)(void* ptr, size_t bytes);
int ITM_FN(host_unregister)(void* ptr);
int ITM_FN(memcpy_h2d)(void* dst_dev, const void* src_host, size_t bytes, itm_stream stream);
int ITM_FN(memcpy_d2h)(void* dst_host, const void* src_dev, size_t bytes, itm_stream stream);
int ITM_FN(stream_synchronize)(itm_stream stream);
/* A stream of the runtime THIS library runs on (hipStreamCreateWithFlags(hipStreamNonBlocking)), for hosts that have no HIP binding of
 * their own -- or whose process holds a second copy of the runtime (a framework that bundles one): a stream must come from the runtime
 * that launches on it.  Every `itm_stream` argument also accepts a hipStream_t the host created itself, or NULL (the null stream). */
int ITM_FN(stream_create)(itm_stream* out);
int ITM_FN(stream_destroy)(itm_stream stream);
int ITM_FN(set_device)(int device);

/* (Test hooks -- itm_debug_set and its keys, the division and cull probes -- are declared in itm_debug.h, which only tests include.) */

/* ---- scene -------------------------------------------------------------------------------- */
/* new ITMScene<TVoxel,TIndex>(sceneParams,false,memType)  Objects/ITMScene.h:37-43 ; also
 * allocates the engine scratch of ITMSceneReconstructionEngine_CPU ctor
 * (DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:9-15). Does NOT reset. */
int ITM_FN(scene_create)(const itm_scene_config* cfg, const itm_scene_params* params,
                         itm_scene** out);
int ITM_FN(scene_destroy)(itm_scene* scene);
int ITM_FN(scene_get_config)(const itm_scene* scene, itm_scene_config* cfg,
                             itm_scene_params* params);

/* ITMSceneReconstructionEngine::ResetScene            Engine/ITMSceneReconstructionEngine.h:35
 * (CPU: DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:24-45 hash, :301-312 dense) */
int ITM_FN(reset_scene)(itm_scene* scene, itm_stream stream);

/* ITMVisualisationEngine::CreateRenderState(imgSize)   Engine/ITMVisualisationEngine.h:79
 * (CPU: DeviceSpecific/CPU/ITMVisualisationEngine_CPU.cpp:18-32; ITMRenderState ctor fills the
 * range image with (vf_min, vf_max), Objects/ITMRenderState.h:51-75) */
int ITM_FN(render_state_create)(const itm_scene* scene, int w, int h, itm_render_state** out);
int ITM_FN(render_state_destroy)(itm_render_state* rs);

/* ---- the four calls of a frame ----------------------------------------------------------------------------------------------
 * ITMMainEngine::ProcessFrame reaches the engines as AllocateSceneFromDepth -> IntegrateIntoScene (Engine/ITMDenseMapper.cpp:50-57)
 * -> CreateExpectedDepths -> CreateICPMaps (Engine/ITMTrackingController.cpp:30-46).  By DEFAULT each of the entry points below
 * enqueues its kernels on `stream` before it returns (8 launches per frame), like the reference's CUDA engines on their stream.
 *
 * itm_scene_set_deferred_fusion(scene, 1) lets the library fuse the sequence (hash scenes): the first three calls then RECORD their
 * arguments (after checking everything they could be refused for) and return with NOTHING enqueued; itm_create_icp_maps for the same
 * view, pose and stream completes the sequence and launches the fused frame of itm_process_frame (5 launches instead of 8).  Any
 * other call of this library that names the scene or the render state, any copy / view-builder call of this library that writes an
 * image the recorded view reads, itm_stream_synchronize on the recording stream, itm_flush and itm_render_state_destroy launch what
 * was recorded first, call by call.  Results are identical either way.  The contract a host accepts by switching it on:
 *   - the images of the view are read when the sequence is LAUNCHED: they stay valid and unchanged until then -- a host that
 *     overwrites them with its OWN kernels or copies between AllocateSceneFromDepth and CreateICPMaps calls itm_flush first;
 *   - stream order only holds from the launching call on: before a hipEventRecord / hipStreamWaitEvent / kernel of the host's own
 *     that is meant to run behind one of the first three calls, and before it uses a raw pointer from itm_buffer_ptr, the host calls
 *     itm_flush;
 *   - the four calls of one frame come from one thread (calls for different scenes may come from different threads).
 * The reference's callers meet all three (the view is built before the tracker runs, Engine/ITMMainEngine.cpp:111-127, and results
 * are read through the engines), so the adapters (include/itm_hip_engines.hpp, integration/ITMEngines_HIP.h) switch it on.
 * Environment: ITM_DEFERRED_FUSION=1 switches it on for every new scene, ITM_NO_DEFERRED_FUSION=1 off whatever the host asked for. */
int ITM_FN(scene_set_deferred_fusion)(itm_scene* scene, int on);
int ITM_FN(flush)(itm_scene* scene, itm_render_state* rs, itm_stream stream);   /* scene == rs == NULL: everything recorded on `stream` */

/* ITMSceneReconstructionEngine::AllocateSceneFromDepth(scene, view, trackingState, renderState,
 * onlyUpdateVisibleList)                               Engine/ITMSceneReconstructionEngine.h:40
 * (CPU hash :116-291, dense :314-317 no-op) */
int ITM_FN(allocate_scene_from_depth)(itm_scene* scene, const itm_view* view,
                                      itm_render_state* rs, int onlyUpdateVisibleList,
                                      itm_stream stream);

/* ITMSceneReconstructionEngine::IntegrateIntoScene     Engine/ITMSceneReconstructionEngine.h:46
 * (CPU hash :47-114, dense :319-369) */
int ITM_FN(integrate_into_scene)(itm_scene* scene, const itm_view* view, itm_render_state* rs,
                                 itm_stream stream);

/* IITMVisualisationEngine::FindVisibleBlocks(pose, intrinsics, renderState)          :39
 * (CPU :39-77; dense: no-op :34-37) */
int ITM_FN(find_visible_blocks)(const itm_scene* scene, const float M[16], const float intr[4],
                                itm_render_state* rs, itm_stream stream);

/* IITMVisualisationEngine::CreateExpectedDepths(pose, intrinsics, renderState)       :45
 * (CPU hash :93-152, dense :79-91) */
int ITM_FN(create_expected_depths)(const itm_scene* scene, const float M[16],
                                   const float intr[4], itm_render_state* rs, itm_stream stream);

/* IITMVisualisationEngine::RenderImage(pose, intrinsics, renderState, outputImage, type) :49
 * (CPU :190-240, :356-368).  out_rgba: uchar4[h*w] device pointer; NULL = rs->raycastImage. */
int ITM_FN(render_image)(const itm_scene* scene, const float M[16], const float intr[4],
                         itm_render_state* rs, uint8_t* out_rgba, int type, itm_stream stream);

/* IITMVisualisationEngine::FindSurface(pose, intrinsics, renderState)               :53
 * (CPU :370-381): raycast only, result in rs->raycastResult. */
int ITM_FN(find_surface)(const itm_scene* scene, const float M[16], const float intr[4],
                         itm_render_state* rs, itm_stream stream);

/* IITMVisualisationEngine::CreatePointCloud(view, trackingState, renderState, skipPoints) :58
 * (CPU :242-264, :424-462).  locations/colours: Vector4f[h*w] = trackingState->pointCloud;
 * count is left in itm_counters::noTotalPoints. */
int ITM_FN(create_point_cloud)(const itm_scene* scene, const itm_view* view,
                               itm_render_state* rs, int skipPoints, float* locations,
                               float* colours, itm_stream stream);

/* IITMVisualisationEngine::CreateICPMaps(view, trackingState, renderState)          :63
 * (CPU :266-287).  points/normals: Vector4f[h*w] = trackingState->pointCloud->locations /
 * ->colours.  Grey rendering goes to rs->raycastImage. */
int ITM_FN(create_icp_maps)(const itm_scene* scene, const itm_view* view, itm_render_state* rs,
                            float* points, float* normals, itm_stream stream);

/* IITMVisualisationEngine::ForwardRender(view, trackingState, renderState)          :70
 * (CPU :289-354) */
int ITM_FN(forward_render)(const itm_scene* scene, const itm_view* view, itm_render_state* rs,
                           itm_stream stream);

/* The per-frame call sequence of ITMMainEngine::ProcessFrame after the view is built
 * (Engine/ITMMainEngine.cpp:123-126): ITMDenseMapper::ProcessFrame (Engine/ITMDenseMapper.cpp:
 * 50-65: AllocateSceneFromDepth + IntegrateIntoScene) followed by
 * ITMTrackingController::Prepare with requiresFullRendering (Engine/ITMTrackingController.cpp:
 * 31-35: CreateExpectedDepths + CreateICPMaps).  Results are identical to issuing the four calls
 * one after another; this entry point only removes launch overhead. */
int ITM_FN(process_frame)(itm_scene* scene, const itm_view* view, itm_render_state* rs,
                          float* points, float* normals, itm_stream stream);
/* The same frame, for a host that already HAS the next frame (an offline sequence, a buffered sensor): `next` (may be NULL) is the view
 * the next call will be made for.  Its per-pixel block requests -- the first stage of its AllocateSceneFromDepth, which only reads the
 * table as this frame's allocation leaves it -- ride in this frame's last launch, beside the ICP maps; the next frame then starts with
 * its visible-list launch (4 launches per frame instead of 5).  Results are identical to itm_process_frame, frame by frame.  Contract:
 * the next allocation on this render state (itm_process_frame[_ahead] or itm_allocate_scene_from_depth) must be for exactly that view
 * (same depth pointer, size, pose, intrinsics), and the scene must not be reset or have its table uploaded in between, else that
 * allocation returns ITM_ERR_INVALID (the render state's visible types then hold the marks of the abandoned requests: recreate it). */
int ITM_FN(process_frame_ahead)(itm_scene* scene, const itm_view* view, const itm_view* next, itm_render_state* rs,
                                float* points, float* normals, itm_stream stream);
/* While the requests of `next` are pending the render state's visible types carry their marks and the scene's request keys are in
 * use: a download / upload / save of the visible list, itm_find_visible_blocks on that render state and an allocation through ANOTHER
 * render state of the scene return ITM_ERR_INVALID.  itm_cancel_ahead abandons the requests: keys and counters return to "no
 * request", the marks are cleared and the entries of the visible list read 3 -- the state of the reference's AllocateSceneFromDepth
 * right after its "previous list -> 3" loop (CPU :160-161) -- so that any allocation may follow.  itm_reset_scene cancels by itself. */
int ITM_FN(cancel_ahead)(itm_scene* scene, itm_render_state* rs, itm_stream stream);

/* ---- view builder (the step before the path) -------------------------------- */
/* convertDepthAffineToFloat  DeviceAgnostic/ITMViewBuilder.h:22-28 */
int ITM_FN(convert_depth_affine)(const int16_t* raw, float* depth_out, int w, int h, float a,
                                 float b, itm_stream stream);
/* convertDisparityToDepth    DeviceAgnostic/ITMViewBuilder.h:7-20 */
int ITM_FN(convert_disparity)(const int16_t* raw, float* depth_out, int w, int h, float c0,
                              float c1, float fx_depth, itm_stream stream);

/* ITMViewBuilder::DepthFiltering  Engine/ITMViewBuilder.h:32, DeviceSpecific/CPU/ITMViewBuilder_CPU.cpp:119-130,
 * filterDepth DeviceAgnostic/ITMViewBuilder.h:30-52: one 5x5 bilateral pass; `out` is cleared, then pixels
 * 2 <= x < w-2, 2 <= y < h-2 are written (invalid input pixel -> -1).  in != out. */
int ITM_FN(filter_depth)(const float* in, float* out, int w, int h, itm_stream stream);
/* ITMViewBuilder::ComputeNormalAndWeights  Engine/ITMViewBuilder.h:33, ..._CPU.cpp:132-145, computeNormalAndWeight
 * DeviceAgnostic/ITMViewBuilder.h:55-117.  normals: float4[h*w], sigmaZ: float[h*w]; only the interior
 * (2-pixel border excluded) is written, as in the reference.  `intr` = projectionParamsSimple.all (fx,fy,cx,cy). */
int ITM_FN(compute_normal_and_weights)(const float* depth, float* normals, float* sigmaZ, int w, int h,
                                       const float intr[4], itm_stream stream);
/* ITMViewBuilder::UpdateView(view, rgb, rawDepth, useBilateralFilter, modelSensorNoise)  ..._CPU.cpp:14-63, for a raw
 * depth frame already in device memory: conversion by calibration type (0 = TRAFO_KINECT disparity, 1 = TRAFO_AFFINE,
 * Objects/ITMDisparityCalib.h:24-29), optionally the five bilateral passes (ping-pong with `scratch`, float[h*w]) and
 * the normal / uncertainty images (normals, sigmaZ may be NULL when modelSensorNoise == 0). */
int ITM_FN(update_view)(const int16_t* raw, int w, int h, int calibType, float c0, float c1, const float intr_d[4],
                        int useBilateralFilter, int modelSensorNoise, float* depth_out, float* scratch,
                        float* normals, float* sigmaZ, itm_stream stream);

/* ---- raw frames from the host: the first statement of ITMViewBuilder_CUDA::UpdateView, shortImage->SetFrom(rawDepthImage, CPU_TO_CUDA)
 * (Engine/DeviceSpecific/CUDA/ITMViewBuilder_CUDA.cu:53), a synchronous copy there.  A stager owns `slots` device images of w x h
 * shorts and a copy stream: itm_depth_stager_upload puts a frame (PINNED host memory) on the copy stream and returns at once -- up to
 * slots - 1 frames ahead of the one being fused --, itm_depth_stager_acquire makes the frame's stream wait for the OLDEST uploaded frame
 * and returns its device image (the `raw` argument of itm_update_view), itm_depth_stager_release says that everything submitted to
 * that stream so far is what read it: the slot is overwritten only behind that work.  One acquire / release pair at a time, frames leave
 * in the order they were uploaded.  Host-side object, not thread safe. */
typedef struct itm_depth_stager itm_depth_stager;
int ITM_FN(depth_stager_create)(int w, int h, int slots, itm_depth_stager** out);
int ITM_FN(depth_stager_destroy)(itm_depth_stager* g);
int ITM_FN(depth_stager_upload)(itm_depth_stager* g, const int16_t* pinned_host);
int ITM_FN(depth_stager_acquire)(itm_depth_stager* g, itm_stream stream, const int16_t** device_image);
/* itm_depth_stager_upload returns BEFORE the copy stream has read the host buffer: a pinned buffer may be rewritten only once
 * *busy == 0 here (no upload still in flight; a query, never a wait), or after its frame has been acquired and the acquiring stream
 * synchronised.  *waiting = uploaded frames not acquired yet.  Either pointer may be NULL. */
int ITM_FN(depth_stager_pending)(itm_depth_stager* g, int* waiting, int* busy);
int ITM_FN(depth_stager_release)(itm_depth_stager* g, itm_stream stream);
/* The copy may also CONVERT: after itm_depth_stager_set_conversion (calibType, c0, c1 and fx = intr_d[0] as in itm_update_view; call it
 * while no frame waits in the ring) every uploaded frame is also turned into the float depth image of itm_update_view's first step
 * (convertDisparityToDepth / convertDepthAffineToFloat, DeviceAgnostic/ITMViewBuilder.h:7-28) by the copy itself, and
 * itm_depth_stager_acquire_depth hands out that image (and, if asked, the raw one) instead of itm_depth_stager_acquire.  For a view
 * without bilateral filter and noise model it IS view->depth -- no conversion launch on the frame's stream --; it stays valid until
 * the release, which therefore comes after the LAST launch that reads the frame's depth. */
int ITM_FN(depth_stager_set_conversion)(itm_depth_stager* g, int calibType, float c0, float c1, float fx);
int ITM_FN(depth_stager_acquire_depth)(itm_depth_stager* g, itm_stream stream, const int16_t** raw_image, const float** depth_image);

/* ---- on-disk input formats of the view builder's sources (host memory, no device work) -------------------------
 * Depth: PGM "P5" (or ASCII "P2") with maxval > 256: 16-bit samples stored BIG-endian, swapped on load
 * (Utils/FileUtils.cpp:377-421); colour: PPM "P6" / "P3" -> RGBA with alpha 255 (:324-375); writers :251-322
 * (the float writer stores millimetres WITHOUT the byte swap, as the reference does).  Calibration text:
 * ITMLib/Utils/ITMCalibIO.cpp:10-101 (rgb intrinsics, depth intrinsics, 3x4 rgb->depth extrinsics, disparity calib;
 * "0 0" selects affine 1/1000).  All pointers are host pointers.  Return ITM_OK or ITM_ERR_INVALID. */
typedef struct itm_rgbd_calib {
  float size_rgb[2], intr_rgb[4];          /* sizeX sizeY ; fx fy cx cy                                  */
  float size_d[2], intr_d[4];
  float rgb_to_depth[16];                  /* trafo_rgb_to_depth.calib, column-major                    */
  float rgb_to_depth_inv[16];              /* .calib_inv (transpose / -R^T t, Objects/ITMExtrinsics.h:32-42) */
  int32_t disparityType;                   /* 0 = TRAFO_KINECT, 1 = TRAFO_AFFINE                        */
  float disparityParams[2];
} itm_rgbd_calib;
int ITM_FN(read_depth_image)(const char* path, int16_t* dst, int capacityPixels, int* w, int* h);
int ITM_FN(read_rgb_image)(const char* path, uint8_t* dst_rgba, int capacityPixels, int* w, int* h);
int ITM_FN(write_depth_image)(const char* path, const int16_t* src, int w, int h);
int ITM_FN(write_rgb_image)(const char* path, const uint8_t* src_rgba, int w, int h);
int ITM_FN(write_float_depth_image)(const char* path, const float* src, int w, int h);
int ITM_FN(read_rgbd_calib)(const char* path, itm_rgbd_calib* out);

/* ---- ICP depth tracker (the step after the path) -------------------------------------
 * Consumes the points / normals maps CreateICPMaps writes.  Device-specific half of the reference tracker:
 *   ITMLowLevelEngine::FilterSubsampleWithHoles (float)  DeviceAgnostic/ITMLowLevelEngine.h:26-47,
 *                                                         DeviceSpecific/CPU/ITMLowLevelEngine_CPU.cpp:50-62
 *   ITMDepthTracker_CPU::ComputeGandH                     DeviceSpecific/CPU/ITMDepthTracker_CPU.cpp:15-79,
 *                                                         DeviceAgnostic/ITMDepthTracker.h:8-106
 * and the host Levenberg-Marquardt loop around it:
 *   ITMDepthTracker::TrackCamera                          Engine/ITMDepthTracker.cpp:149-200 (+ :79-147, ITMPose::Coerce) */
enum itm_tracker_iteration {               /* TrackerIterationType, Utils/ITMLibDefines.h:277-283 */
  ITM_TRACKER_ITERATION_ROTATION = 1,
  ITM_TRACKER_ITERATION_TRANSLATION = 2,
  ITM_TRACKER_ITERATION_BOTH = 3,
  ITM_TRACKER_ITERATION_NONE = 4
};
typedef struct itm_tracker_config {        /* ITMLibSettings fields the tracker factory passes on */
  int32_t noHierarchyLevels;               /* <= 8; default 5                                  */
  int32_t trackingRegime[8];               /* per level, level 0 = full resolution; default B B R R R */
  int32_t noICPRunTillLevel;               /* default 0                                        */
  float distThresh;                        /* depthTrackerICPThreshold, default 0.1*0.1        */
  float terminationThreshold;              /* depthTrackerTerminationThreshold, default 1e-3   */
} itm_tracker_config;
typedef struct itm_tracker_gh {
  float f;                                 /* sqrt(sum b^2)/n, or 1e5 when n <= 100            */
  float nabla[6];
  float hessian[36];                       /* r + c*6; entries outside the active block are 0  */
  int32_t noValidPoints;
} itm_tracker_gh;
/* out: float[(h_in/2)*(w_in/2)] */
int ITM_FN(filter_subsample_with_holes)(const float* in, int w_in, int h_in, float* out, itm_stream stream);
/* Synchronises `stream` (the host loop needs the sums).  Sums are accumulated by a fixed-order reduction
 * tree in double precision: deterministic, and within float rounding of the reference's sequential sum. */
int ITM_FN(tracker_compute_g_and_h)(const float* depth, int w, int h, const float viewIntr[4],
                                    const float* pointsMap, const float* normalsMap, int sceneW, int sceneH,
                                    const float sceneIntr[4], const float approxInvPose[16],
                                    const float scenePose[16], float distThresh, int iterationType,
                                    itm_tracker_gh* out, itm_stream stream);
/* ITMDepthTracker::TrackCamera: view->depth / intr_d / M_d (initial pose_d), the ICP maps of the previous
 * frame and the pose they were rendered from (pose_pointCloud); writes the refined pose_d to M_d_out. */
int ITM_FN(track_camera)(const itm_tracker_config* cfg, const itm_view* view, const float* pointsMap,
                         const float* normalsMap, const float scenePose[16], float M_d_out[16],
                         itm_stream stream);

/* The tracker as an object (ITMDepthTracker owns its hierarchy and reduction buffers, Engine/ITMDepthTracker.cpp:18-44): one
 * handle = the device partials, the pinned result records and the depth pyramid of ONE tracker.  Calls on a handle are
 * serialised by the handle; different handles (e.g. one per depth stream / HIP stream / host thread) are independent.  The
 * two handle-less entry points above use a handle private to the calling host thread.  A reduction that does not arrive
 * (hung or failed kernel) ends the call with ITM_ERR_DEVICE after a bounded wait instead of stalling the host. */
typedef struct itm_tracker itm_tracker;
int ITM_FN(tracker_create)(itm_tracker** out);
int ITM_FN(tracker_destroy)(itm_tracker* tracker);
int ITM_FN(tracker_g_and_h)(itm_tracker* tracker, const float* depth, int w, int h, const float viewIntr[4],
                            const float* pointsMap, const float* normalsMap, int sceneW, int sceneH,
                            const float sceneIntr[4], const float approxInvPose[16], const float scenePose[16],
                            float distThresh, int iterationType, itm_tracker_gh* out, itm_stream stream);
int ITM_FN(tracker_track_camera)(itm_tracker* tracker, const itm_tracker_config* cfg, const itm_view* view,
                                 const float* pointsMap, const float* normalsMap, const float scenePose[16],
                                 float M_d_out[16], itm_stream stream);

/* ---- state access ------------------------------------------------------------------------- */
/* Blocks until `stream` has drained, then reads the device-side counters. */
int ITM_FN(get_counters)(const itm_scene* scene, const itm_render_state* rs, itm_counters* out,
                         itm_stream stream);
/* Restores counters (checkpoint / test set-up).  Only lastFreeBlockId, lastFreeExcessListId,
 * noVisibleEntries are taken from `in`. */
int ITM_FN(set_counters)(itm_scene* scene, itm_render_state* rs, const itm_counters* in,
                         itm_stream stream);
size_t ITM_FN(buffer_bytes)(const itm_scene* scene, const itm_render_state* rs, int which);
/* Synchronous copies of whole buffers between host memory and the scene / render state. */
int ITM_FN(download)(const itm_scene* scene, const itm_render_state* rs, int which, void* dst_host,
                     size_t bytes, itm_stream stream);
int ITM_FN(upload)(itm_scene* scene, itm_render_state* rs, int which, const void* src_host,
                   size_t bytes, itm_stream stream);
/* Scene checkpoint: one file per memory block in the layout of ORUtils/MemoryBlockPersister.h:17-130
 * (int32 element count, then the raw elements): hash.dat (ITMHashEntry), excess.dat (int), alloc.dat (int),
 * voxel.dat (TVoxel), counters.dat (itm_counters as 8 ints), config.dat (itm_scene_config + itm_scene_params as bytes,
 * checked on load), and for a render state visible_ids.dat (int) / visible_type.dat (uchar).  Loading into a scene of
 * the same configuration and continuing gives bit-identical results to the uninterrupted run.  `dir` must exist. */
int ITM_FN(scene_save)(const itm_scene* scene, const itm_render_state* rs, const char* dir, itm_stream stream);
int ITM_FN(scene_load)(itm_scene* scene, itm_render_state* rs, const char* dir, itm_stream stream);
/* ---- scene export: marching-cubes mesh ----------------------------------------------------------------
 * ITMMesh (Objects/ITMMesh.h:14-124): a triangle buffer of noMaxTriangles = SDF_LOCAL_BLOCK_NUM * 32 entries {p0, p1, p2} (nine
 * floats, metres) in device memory; max_triangles == 0 selects that default for the scene's pool size.
 * ITMMeshingEngine::MeshScene(mesh, scene) (Engine/ITMMeshingEngine.h:19-26, CPU: DeviceSpecific/CPU/ITMMeshingEngine_CPU.cpp:
 * 19-58): clears the buffer, then appends the triangles of every cell of every allocated block in table-slot order, voxels in
 * z, y, x order -- the same ORDER as the reference, and the same behaviour when the buffer is full (count stops at
 * noMaxTriangles - 1, the last slot holds the last triangle generated).  Dense scenes produce no triangles, as in the
 * reference (:70-72). */
typedef struct itm_mesh itm_mesh;
int ITM_FN(mesh_create)(const itm_scene* scene, uint32_t max_triangles, itm_mesh** out);
int ITM_FN(mesh_destroy)(itm_mesh* mesh);
int ITM_FN(mesh_scene)(const itm_scene* scene, itm_mesh* mesh, itm_stream stream);
/* mesh->noTotalTriangles / noMaxTriangles / the triangle buffer; synchronises `stream`.  Any output pointer may be NULL. */
int ITM_FN(mesh_info)(const itm_mesh* mesh, uint32_t* noTotalTriangles, uint32_t* noMaxTriangles,
                      const float** triangles, itm_stream stream);
/* copies min(noTotalTriangles, capacity) triangles (9 floats each) to host memory; synchronises `stream` */
int ITM_FN(mesh_download)(const itm_mesh* mesh, float* dst_host, uint32_t capacityTriangles,
                          uint32_t* noTotalTriangles, itm_stream stream);
/* ITMMesh::WriteOBJ (Objects/ITMMesh.h:34-62) and ITMMesh::WriteSTL (:64-110), byte-identical files */
int ITM_FN(mesh_write_obj)(const itm_mesh* mesh, const char* path, itm_stream stream);
int ITM_FN(mesh_write_stl)(const itm_mesh* mesh, const char* path, itm_stream stream);

/* ---- multi-stream exchange (BASELINE configs[3]) ------------------------------------------------------------------
 * One depth stream per GPU; fusion needs no collective.  Per frame every rank publishes the record of itm_export_visible_record
 * ({M_d[16], noVisibleEntries, ids[max_ids]}); every `batch` frames the records are all-gathered with RCCL on a side stream owned by
 * the exchange, so that each rank holds the pose and the live block list of every stream (input of a shared-map merger).  The frame
 * stream never waits for a collective of the current batch, and the collectives are put on the side stream by a thread the exchange owns
 * (enqueueing an all-gather costs a host thread 60-80 us).  RCCL is loaded on first use (dlopen; ITM_RCCL_LIBRARY in the environment
 * names another collective library -- a site's own build, the tests' stand-in for several ranks on one GPU -- and says so on stderr).
 * Every world size, one rank included, gets an RCCL communicator and runs ncclAllGather.  Bootstrap: rank 0 calls itm_exchange_unique_id, the host hands the 128 bytes
 * to every rank; `id` may be NULL for world == 1 (the library then makes the id itself). */
typedef struct itm_exchange itm_exchange;
int ITM_FN(exchange_unique_id)(unsigned char id[128]);
int ITM_FN(exchange_create)(int world, int rank, const unsigned char id[128], int max_ids, int batch, itm_exchange** out);
int ITM_FN(exchange_destroy)(itm_exchange* exchange);
/* frame f of this rank's stream: record copy on `frame_stream`; on the last frame of a batch also the collective (side stream) */
int ITM_FN(exchange_step)(itm_exchange* exchange, const itm_render_state* rs, const float M_d[16], itm_stream frame_stream);
int ITM_FN(exchange_info)(const itm_exchange* exchange, int* world, int* rank, int* max_ids, int* batch, const void** gathered_device);
/* Hand-off of a gathered table to a DEVICE-side consumer (the per-GPU global visibility table of SURVEY 8e; a shared-map merger's kernel).
 * Every slot of the exchange's ring of eight has its own table.  itm_exchange_acquire makes `consumer_stream` wait for the newest
 * collective issued so far and returns its table (world x batch records, rank-major; *first_frame = the frame number of the batch's
 * first record; NULL / -1 before the first collective).  The table belongs to the consumer until itm_exchange_release -- or the next
 * itm_exchange_acquire, which releases first -- both given the stream the consumer's reads were put on: the ring skips a held slot,
 * and the collective that next writes a released slot waits (on the exchange's side stream) behind those reads.  One holder at a time.
 * Frames and collectives go on meanwhile; nothing here blocks the host.  Threads: the slot is chosen and marked as held under one lock,
 * so a consumer thread other than the one calling itm_exchange_step is safe; two consumer threads are not (one holder at a time). */
int ITM_FN(exchange_acquire)(itm_exchange* exchange, itm_stream consumer_stream, const int32_t** table, long long* first_frame);
int ITM_FN(exchange_release)(itm_exchange* exchange, itm_stream consumer_stream);
/* host copy of the gathered table, world x batch records of (17 + max_ids) int32 words, rank-major; synchronises the side stream */
int ITM_FN(exchange_table)(itm_exchange* exchange, int32_t* dst_host, size_t words);
/* Self-check, on by default at every world size (ITM_EXCHANGE_SELF_CHECK=0 in the environment switches it off): behind each collective,
 * on the side stream, this rank's own block of the gathered table is compared with the batch buffer it sent; words that differ make the
 * next itm_exchange_step / itm_exchange_table return ITM_ERR_DEVICE.  Here: collectives checked so far (-1 = self-check off) and words
 * found different; synchronises the side stream. */
int ITM_FN(exchange_self_check)(itm_exchange* exchange, int* collectives_checked, int* mismatched_words);

/* ---- swapping (Engine/ITMSwappingEngine.h:19-36, DeviceSpecific/CPU/ITMSwappingEngine_CPU.cpp, Objects/ITMGlobalCache.h) ----
 * Scenes created with useSwapping keep an ITMGlobalCache in HOST memory (one block slot + one `hasStoredData` flag per table entry) and
 * a swap state per entry on the device (0 host has the newest data / never loaded, 1 visible and still to be combined with the host's
 * copy, 2 the device has the newest data).  ITMDenseMapper::ProcessFrame calls the two methods after the integration
 * (Engine/ITMDenseMapper.cpp:59-64):
 *   IntegrateGlobalIntoLocal  the first transferBlockNum entries in state 1, in table order: the host's stored block (if any) is
 *                             combined voxel by voxel into the device block (weighted by w_depth / w_color), state -> 2
 *   SaveToGlobalMemory        the first transferBlockNum entries in state 2 that hold a block and are not visible: the block goes to
 *                             the host cache, the entry's ptr becomes -1, the voxel block returns to the allocation list, state -> 0
 * Both synchronise with the host (the reference's CUDA twin does, through cudaMemcpy).  Return ITM_ERR_INVALID for scenes without swapping. */
int ITM_FN(swap_integrate_global_into_local)(itm_scene* scene, itm_render_state* rs, itm_stream stream);
int ITM_FN(swap_save_to_global_memory)(itm_scene* scene, itm_render_state* rs, itm_stream stream);
/* ITMGlobalCache::HasStoredData / GetStoredVoxelBlock: copies the stored block of table entry `entry` (512 voxels) to dst_host if there
 * is one; *has = 1 / 0 */
int ITM_FN(global_cache_get)(const itm_scene* scene, int entry, void* dst_host, int* has);
/* hasStoredData[noTotalEntries] as bytes */
int ITM_FN(global_cache_flags)(const itm_scene* scene, uint8_t* dst_host, size_t bytes);

/* The acceleration structures a hash scene carries beside the reference's table (none of them part of the reference's state, all
 * derived from it): a block directory and a slot directory over a cube of 512^3 blocks, an sdf mirror over 256^3 blocks for the
 * short voxel types.  The cubes are NOT tied to the world origin: the first frame places them around its camera (the reference's
 * table has no spatial limit, Objects/ITMVoxelBlockHash.h:22-100, and an external pose source chooses the world frame), a frame
 * whose view leaves a cube moves it (emptied and refilled from the table on the frame's stream, O(allocated blocks)).  Blocks outside
 * a cube are found through the table as the reference finds them.  origin_*: block coordinates of cell (0, 0, 0); moves: cube moves
 * since creation; *_bytes: device memory of each structure (0 = absent). */
typedef struct itm_accel_info {
  int64_t directory_bytes, slot_directory_bytes, mirror_bytes;
  int32_t origin_directory[3], origin_mirror[3];
  int32_t placed;
  int64_t moves;
  int32_t mirror_pages, mirror_pages_mapped;   /* PAGED mirror (ITM_MIRROR=paged in the environment, or a device without 3 x 17 GB to spare): 4 MB pages (16 x 16 x 16 blocks of
                                                  int16 sdf) the pool holds / has handed out (reading this synchronises the device); both 0 for the DENSE form, whose
                                                  mirror_bytes are the whole cube's 17 GB */
} itm_accel_info;
int ITM_FN(scene_accel_info)(const itm_scene* scene, itm_accel_info* out);

/* Device address of a buffer (zero-copy hand-off to e.g. a collective); NULL if absent. */
void* ITM_FN(buffer_ptr)(const itm_scene* scene, const itm_render_state* rs, int which);

/* ---- per-kernel timers (the reference wraps ProcessFrame in a stopwatch: Engine/CLIEngine.cpp:44-86,
 * Utils/NVTimer.h; here hipEvents bracket individual kernels on their stream) ------------------- */
enum itm_timed_kernel {
  ITM_TK_REQUEST = 0,       /* per-pixel block requests (buildHashAllocAndVisibleTypePP)   */
  ITM_TK_ALLOC_SWEEP = 1,   /* ordered allocation sweep                                    */
  ITM_TK_VISIBLE_LIST = 2,  /* frustum re-test + ordered compaction (2 launches)           */
  ITM_TK_INTEGRATE = 3,     /* IntegrateIntoScene kernel                                   */
  ITM_TK_RANGE = 4,         /* CreateExpectedDepths kernels                                */
  ITM_TK_RAYCAST = 5,       /* GenericRaycast kernel                                       */
  ITM_TK_ICP_MAPS = 6,      /* processPixelICP kernel                                      */
  ITM_TK_EMPTY = 7,         /* itm_profile_calibrate: event pairs with nothing between them          */
  ITM_TK_COUNT = 8
};
typedef struct itm_profile {
  int32_t calls[ITM_TK_COUNT];
  double total_ms[ITM_TK_COUNT];
} itm_profile;
/* kernel_mask: bit i enables timing of kernel i (0 disables everything).  Each timed launch costs
 * two hipEventRecord calls on the frame stream. */
int ITM_FN(profile_enable)(itm_scene* scene, uint32_t kernel_mask);
/* Time only every `every`-th launch of each enabled kernel (1 = every launch, the default): the
 * event pair costs a few microseconds of stream time, which a throughput run should not pay on
 * every frame.  The average launch duration is over the sampled launches. */
int ITM_FN(profile_sample)(itm_scene* scene, int every);
/* Records n EMPTY brackets on `stream` (slot ITM_TK_EMPTY of itm_profile): the interval an event pair measures when nothing lies
 * between the two records, i.e. what the pair itself adds to every timed kernel's figure on this machine. */
int ITM_FN(profile_calibrate)(itm_scene* scene, int n, itm_stream stream);
/* Waits for the recorded events, accumulates and returns the totals; reset != 0 clears them. */
int ITM_FN(profile_read)(itm_scene* scene, itm_profile* out, int reset);

/* Fixed-size record for the multi-stream exchange: writes
 * {float M_d[16]; int32 noVisibleEntries; int32 ids[max_ids]} (padded with -1) to `dst`
 * (device memory, (17+max_ids)*4 bytes) on `stream`, without a host round-trip. */
int ITM_FN(export_visible_record)(const itm_render_state* rs, const float M_d[16], int max_ids,
                                  void* dst, itm_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* ITM_HIP_H_ */
