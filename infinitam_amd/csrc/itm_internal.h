// itm_internal.h -- objects behind the opaque handles of include/itm_hip.h and shared helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/itm_hip.h"
#include "../../include/itm_debug.h"      // the library implements the test hooks; hosts never see them
#include "itm_types.h"

namespace itm {

// Device-resident scalars.  Nothing on the frame path reads them back to the host; grids are
// sized from static upper bounds and kernels loop up to these counts.
struct SceneCounters {
  int32_t lastFreeBlockId;       // ITMLocalVBA::lastFreeBlockId
  int32_t lastFreeExcessListId;  // ITMVoxelBlockHash::lastFreeExcessListId
  int32_t noAllocRequests;       // requests seen by the last allocation sweep
  int32_t statusFlags;           // bit0: allocation key overflow
};
struct RenderCounters {
  int32_t noVisibleEntries;        // ITMRenderState_VH::noVisibleEntries
  int32_t noFwdProjMissingPoints;  // ITMRenderState::noFwdProjMissingPoints
  int32_t noTotalPoints;           // ITMPointCloud::noTotalPoints
  int32_t noRenderingBlocks;       // numRenderingBlocks of CreateExpectedDepths
  int32_t rawVisibleCount;         // visible slots before clamping to the list capacity
  int32_t renderingBlocksAccepted; // -1, or the count after the cap replay when noRenderingBlocks reached MAX_RENDERING_BLOCKS
  int32_t listInvalid;             // this frame's visible list could not be ordered (a bounded wait of the one-launch list expired):
                                   // the integration skips the frame instead of fusing through a wrong list; cleared by the next request
  int32_t pad[1];
};

// hipEvent pairs around selected kernels (itm_profile_enable / itm_profile_read)
struct Profiler {
  uint32_t mask = 0;
  int every = 1;                 // time every `every`-th launch of an enabled kernel
  uint32_t tick[ITM_TK_COUNT] = {};
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  double total_ms[ITM_TK_COUNT] = {};
  int32_t calls[ITM_TK_COUNT] = {};
  hipEvent_t get();
  void flush();
};

constexpr int kSweepChunk = 2048;  // hash slots handled by one workgroup in the ordered sweeps

}  // namespace itm

struct itm_scene {
  itm_scene_config cfg;
  itm_scene_params prm;
  int device = 0;
  size_t voxBytes = 0;
  size_t numVoxels = 0;
  int noTotalEntries = 0;
  int numChunks = 0;             // ceil(noTotalEntries / kSweepChunk)
  // ITMScene members, all in HBM
  uint4* hash = nullptr;          // ITMHashEntry[noTotalEntries]
  int32_t* excessList = nullptr;  // int[excessNum]
  void* vba = nullptr;            // TVoxel[numVoxels]
  int32_t* allocList = nullptr;   // int[localBlockNum]
  itm::SceneCounters* counters = nullptr;
  // allocation scratch (replaces entriesAllocType / blockCoords of the reference engines):
  // per-slot winner key of this frame's block requests, zero between frames
  uint32_t* allocKey = nullptr;   // uint32[noTotalEntries]
  int32_t* chunkReq = nullptr;    // int2[2][numChunks]: (requests, excess requests) per sweep chunk, double-buffered
  int32_t* chunkVis = nullptr;    // int[numChunks]: visible slots per sweep chunk (two-pass path, FindVisibleBlocks)
  unsigned long long* chunkGran = nullptr;  // u64[numChunks]: {epoch, visible count} granules of the one-pass visible list
  uint32_t* chunkSweepDone = nullptr;        // u32[numChunks]: epoch stamps of chunks whose excess allocations are in place (fused sweep)
  unsigned long long* chunkKeptGran = nullptr;      // u64 per 32 slots: {epoch, kept bits} of the excess region's shared frustum re-tests (alloc.hip)
  uint32_t listEpoch = 0;
  // occupancy bitmap of the ordered part of the table: bit b set <=> hash[b].ptr >= -1 (an entry lives there: allocated, or swapped
  // out with its chain possibly still resident).  A clear bit
  // proves that no block hashing to bucket b is allocated (excess entries hang off occupied heads),
  // which lets the ray caster skip empty space without touching the 16-byte entries.
  uint32_t* headBits = nullptr;   // uint32[bucketNum / 32]
  // Block directory (itm_types.h): dirPtr[cell] = voxel-block index of the block at that position or -1, cells in brick-major
  // order.  Maintained by the allocation sweep, rebuilt after uploads; an exact mirror of the table entries with ptr >= 0.
  int32_t* dirPtr = nullptr;      // int32[kDirCells]  (512 MB)
  int32_t* dirSlot = nullptr;     // int32[kDirCells]  (512 MB): table slot of the block at that position or -1 (request kernel)
  void* sdfMirror = nullptr;      // the mirror's page pool: int16 / uint32 [mirrorPages * 64 * 512] (512 MB; hash scenes, itm_types.h) or nullptr;
                                  // its page table and page counter travel to the kernels inside `org` (AccelOrigin::mTable / mPages / mMaxPages)
  int mirrorPages = 0;
  // Where the two cubes lie (scene.hip, accel_place): re-placed around the camera when the view leaves them.  Invariant: the only
  // non-empty cells of dirPtr / dirSlot / sdfMirror are those of table entries with ptr >= 0 at `org` -- every path that replaces
  // the table or moves the origin empties exactly those cells first (O(allocated blocks), no 18 GB memset)
  itm::AccelOrigin org = {-itm::kDirHalf, -itm::kDirHalf, -itm::kDirHalf, -itm::kMirrorHalf, -itm::kMirrorHalf, -itm::kMirrorHalf + itm::kMirrorShift, nullptr, nullptr, 0};
  // swapping (scenes with cfg.useSwapping; swapping.hip): ITMHashSwapState per entry on the device, the ITMGlobalCache in host memory
  uint8_t* swapStates = nullptr;           // uchar[noTotalEntries]
  struct SwapHost* swapHost = nullptr;
  unsigned tableEpoch = 0;                 // bumped whenever the table is reset or replaced (requests issued ahead are then void)
  bool countedLive = false;       // this scene is in the per-device count of live hash scenes
  bool orgPlaced = false;         // false until the first frame (or an upload) has placed the cubes
  long long accelMoves = 0;       // times the cubes were re-placed (itm_scene_accel_info)
  uint32_t frameParity = 0;
  // dense integration: min / max tiles of the frame's depth image (integrate.hip), allocated on first use
  float2* depthTiles = nullptr;
  size_t depthTilesCap = 0;
  itm::Profiler* prof = nullptr;
  // engine calls recorded but not yet launched (pending.hip): the render state that holds them, or nullptr
  mutable itm_render_state* deferredRs = nullptr;
  // itm_scene_set_deferred_fusion: the four per-frame engine calls may be recorded and launched as one fused frame (pending.hip).
  // Off unless the host asks for it (or ITM_DEFERRED_FUSION=1 in the environment at scene creation): a recorded call has enqueued
  // nothing on its stream when it returns, which a host that orders its own work by stream must know about.
  bool deferredFusion = false;
  // every render state created for this scene and not yet destroyed (scene.hip): a scene that goes away first takes its render
  // states' back pointers with it, so that destroying them afterwards does not touch freed memory
  mutable std::vector<itm_render_state*> renderStates;
  // itm_process_frame_ahead: the render state whose NEXT frame's block requests are in the table's request keys (one per scene)
  itm_render_state* aheadRs = nullptr;
  // Conditions after which the scene is no longer what the reference would hold (itm_counters::statusFlags): kernels raise them in
  // device memory AND in this word of page-locked host memory, which every entry point reads (a plain host load): the next call on
  // the scene fails with ITM_ERR_DEVICE instead of going on with a state the reference can never be in.  Cleared by ResetScene.
  volatile int32_t* fatalHost = nullptr;   // hipHostMalloc (mapped)
  int32_t* fatalDev = nullptr;             // the same word as the device sees it
};

struct itm_render_state {
  const itm_scene* scene = nullptr;
  int w = 0, h = 0;
  bool hash = false;
  int capIds = 0;
  // true while visibleEntryIDs holds exactly the slots with a non-zero entriesVisibleType (the state
  // AllocateSceneFromDepth leaves behind); lets the next allocation skip the mark-previous launch
  bool listCoherent = true;
  bool lazyThisFrame = false;
  float2* range = nullptr;     // renderingRangeImage  Vector2f[h*w]
  float4* raycast = nullptr;   // raycastResult        Vector4f[h*w]
  float4* fwdProj = nullptr;   // forwardProjection    Vector4f[h*w]
  int32_t* missing = nullptr;  // fwdProjMissingPoints int[h*w]
  uchar4* image = nullptr;     // raycastImage         Vector4u[h*w]
  int32_t* visibleIds = nullptr;   // int[localBlockNum]
  uint8_t* visibleType = nullptr;  // uchar[noTotalEntries]
  itm::RenderCounters* counters = nullptr;
  // scratch
  uint4* projBuf = nullptr;    // per visible entry: projected bounding box + z range (2 x uint4)
  uint2* rangePartials = nullptr;  // [32][ceil(w/8)*ceil(h/8)] partial range images (LDS path)
  int32_t* pixScratch = nullptr;  // int[h*w] (forward projection winners, ordered compaction flags)
  int32_t* pixChunk = nullptr;    // int[ceil(h*w / kSweepChunk)]
  uint8_t* viewFlags = nullptr;   // FindVisibleBlocks: per-slot flags (uchar[numChunks * kSweepChunk]), allocated on first use
  int32_t* viewChunkVis = nullptr; // FindVisibleBlocks: visible slots per chunk
  // dense scenes: the expected-depth image is the constant (0.2, 3.0) of ITMVisualisationEngine_CPU<TVoxel, ITMPlainVoxelArray>::
  // CreateExpectedDepths and the rendering-block counters are its constants too; true once a launch has written them and nothing
  // else has touched them since (uploads, set_counters and loads clear it), so later frames skip the refill
  bool denseRangeReady = false;
  // itm_process_frame on images too large for the fused projection: the projection runs beside the integration (visualise.hip)
  hipStream_t sideStream = nullptr;
  hipEvent_t listReady = nullptr, projectionDone = nullptr;
  // itm_process_frame_ahead: the block requests of the NEXT frame were issued beside this frame's ICP maps; the next allocation
  // must be for exactly this view and skips its request launch
  struct { bool valid = false; const float* depth = nullptr; int w = 0, h = 0; float M_d[16] = {}, intr_d[4] = {}; bool lazy = false; unsigned tableEpoch = 0; } ahead;
  // Deferred fusion (pending.hip): AllocateSceneFromDepth -> IntegrateIntoScene -> CreateExpectedDepths recorded here; CreateICPMaps
  // for the same view then launches the fused frame of itm_process_frame, anything else launches what was recorded one by one.
  // stage: 0 nothing, 1 allocation recorded, 2 + integration, 3 + expected depths
  struct { int stage = 0; itm_view view = {}; hipStream_t st = nullptr; } deferred;
};

namespace itm {

int set_error(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define ITM_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return itm::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define ITM_LAUNCH_CHECK()                                                    \
  do {                                                                        \
    hipError_t _e = hipGetLastError();                                        \
    if (_e != hipSuccess) return itm::hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
  } while (0)

inline hipStream_t as_stream(itm_stream s) { return (hipStream_t)s; }

// RAII bracket: records an event before and after the launches in its scope when kernel `id` is timed
struct KernelTimer {
  Profiler* p; hipStream_t st; int id; hipEvent_t a;
  KernelTimer(const itm_scene* s, int id_, hipStream_t st_) : p(s->prof), st(st_), id(id_), a(nullptr) {
    if (p && ((p->mask >> id) & 1u) && (p->tick[id]++ % (uint32_t)p->every) == 0) { a = p->get(); (void)hipEventRecord(a, st); } else p = nullptr;
  }
  ~KernelTimer() {
    if (p) { hipEvent_t b = p->get(); (void)hipEventRecord(b, st); p->pending.push_back({id, a, b}); }
  }
};

// host-side matrix helpers (host_math.cpp): same operation order as ORUtils/Matrix.h
bool invert4(const float* m, float* out);
void matmul4(const float* lhs, const float* rhs, float* out);

// per-voxel-type dispatch
template <class F>
inline int dispatch_voxel(int voxelType, F&& f) {
  switch (voxelType) {
    case ITM_VOXEL_S: return f(VoxelS{});
    case ITM_VOXEL_F: return f(VoxelF{});
    case ITM_VOXEL_S_RGB: return f(VoxelSRgb{});
    case ITM_VOXEL_F_RGB: return f(VoxelFRgb{});
  }
  return set_error(ITM_ERR_INVALID, "unknown voxel type");
}

// ---- pending work of the entry points (pending.hip) --------------------------------------------------------------------------------
// Launches, unfused and in call order, what the render state has recorded (no-op when nothing is).
int flush_deferred(itm_render_state* rs);
// The prologue of every entry point that reads or writes a scene / render state: fails with ITM_ERR_DEVICE once the scene has raised a
// fatal status, then launches recorded calls of the scene (whichever render state holds them) and of `rs`.
int enter_scene(const itm_scene* s, const itm_render_state* rs);
// Entry points without a handle that write device memory (copies, the view builder) or wait for a stream: recorded calls that read
// [p, p + bytes) -- or, p == nullptr, that were recorded on `st` -- are launched first.
int flush_overlapping(const void* p, size_t bytes, hipStream_t st);
void forget_deferred(itm_render_state* rs);      // the render state is going away
extern int g_debug_no_deferred_fusion;
bool deferred_fusion_default();               // ITM_DEFERRED_FUSION=1 in the environment: new scenes record without being asked
extern int g_debug_force_list_stuck;
extern int g_debug_exchange_device_copy, g_debug_exchange_corrupt_word;      // exchange.hip
// true when rs holds the block requests of a frame issued ahead (itm_process_frame_ahead): `what` is refused with ITM_ERR_INVALID
int refuse_while_ahead(const itm_scene* s, const itm_render_state* rs, const char* what);

// entry points implemented per translation unit
extern int g_debug_explicit_mark;
extern int g_debug_two_pass_visible_list;
extern int g_debug_integrate_wgs;
extern int g_debug_dense_group_cull;
extern int g_debug_dense_classify;
extern int g_debug_dense_no_strips;
extern int g_debug_tracker_launch_per_evaluation;
extern int g_debug_tracker_host_command;
extern int g_debug_tracker_session_unusable;
extern int g_debug_no_sdf_mirror;
extern int g_debug_separate_sweep;
int rebuild_sdf_mirror(itm_scene* s, hipStream_t st);
int launch_swap_after_allocation(itm_scene* s, itm_render_state* rs, hipStream_t st);   // swapping.hip
void free_swap_state(itm_scene* s);
int create_swap_state(itm_scene* s);
int live_hash_scenes(int device);                                     // hash scenes alive on a device (scene.hip)
int accel_unfill(itm_scene* s, hipStream_t st);                      // empties the cubes through the table that filled them
int accel_place(itm_scene* s, const float* invM, hipStream_t st);    // (re-)places the cubes around a view
extern int g_debug_no_fused_projection;
int rebuild_head_bits(itm_scene* s, hipStream_t st);   // occupancy bitmap AND block directory, from the table
extern int g_debug_no_directory;
int launch_request_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, bool fuseRangeInit, hipStream_t st);
int launch_sweep_stage(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, hipStream_t st);
int launch_allocate(itm_scene* s, const itm_view* v, itm_render_state* rs, bool onlyVisible, bool fuseRangeInit, hipStream_t st);
int validate_allocate(const itm_scene* s, const itm_view* v, const itm_render_state* rs, bool onlyVisible);
int validate_integrate(const itm_scene* s, const itm_view* v);
int cancel_ahead(itm_scene* s, itm_render_state* rs, hipStream_t st);
int launch_integrate(itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st, bool fuseProjection = false);
bool can_fuse_projection(const itm_scene* s, const itm_render_state* rs);
int launch_find_visible(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, hipStream_t st);
int launch_expected_depths(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, bool rangeAlreadyInit, hipStream_t st, bool projected = false);
int launch_raycast(const itm_scene* s, const float* invM, const float* intr, itm_render_state* rs, float4* dst, hipStream_t st, bool reduceRange = false);
int launch_icp_maps(const itm_scene* s, const itm_view* v, itm_render_state* rs, float4* points, float4* normals, hipStream_t st, bool reduceRange = false,
                    itm_scene* sceneForNext = nullptr, const itm_view* next = nullptr);
int launch_render_image(const itm_scene* s, const float* M, const float* intr, itm_render_state* rs, uchar4* out, int type, hipStream_t st);
int launch_forward_render(const itm_scene* s, const itm_view* v, itm_render_state* rs, hipStream_t st);
int launch_point_cloud(const itm_scene* s, const itm_view* v, itm_render_state* rs, bool skip, float4* loc, float4* col, hipStream_t st);

}  // namespace itm
