// pending.hip -- the reference's four per-frame engine calls as ONE fused frame, and the bookkeeping every entry point shares.
//
// ITMMainEngine::ProcessFrame reaches the engines as four calls (Engine/ITMDenseMapper.cpp:50-57: AllocateSceneFromDepth,
// IntegrateIntoScene; Engine/ITMTrackingController.cpp:30-46: CreateExpectedDepths, CreateICPMaps).  Launched one by one they are
// eight launches; itm_process_frame does the same work in five (the range-image initialisation rides in the request launch, the
// projection of the visible blocks in the integration launch, the range reduction in the ray cast).  A drop-in back-end is called
// through the four virtuals, so the fusion has to happen BEHIND them:
//
//   * on a scene whose host has asked for it (itm_scene_set_deferred_fusion; never by default: a recorded call has put nothing on its
//     stream when it returns) itm_allocate_scene_from_depth, itm_integrate_into_scene and itm_create_expected_depths RECORD their
//     arguments in the render state (after checking everything they could be refused for) and launch nothing;
//   * itm_create_icp_maps for the same view, pose and stream completes the sequence and launches the fused frame;
//   * every other entry point that could observe or disturb the recorded calls -- any call naming the scene or the render state, a
//     copy or a view-builder call that writes an image the recorded view reads, itm_stream_synchronize on the recording stream,
//     itm_flush -- first launches what was recorded, one launch per call, in call order (the unfused kernels, same results).
//
// Results are those of the four calls launched one by one, bit for bit (tests/test_deferred_fusion.py).  The contract this adds: the
// images of the recorded view are read when the sequence is launched, so a host that overwrites them with its OWN kernels or copies
// (not through this library) between AllocateSceneFromDepth and CreateICPMaps, or orders its own work behind one of the first three
// calls through the stream (an event, a kernel of its own, a raw pointer from itm_buffer_ptr), must call itm_flush first.  The
// reference's callers never do (the view is built before the tracker runs, Engine/ITMMainEngine.cpp:111-127; the engines' results are
// read through the engines), which is why the adapters switch the recording on and a bare C host has to ask for it.
// ITM_NO_DEFERRED_FUSION=1 in the environment (or debug key 19) launches every call at once whatever the scene says;
// ITM_DEFERRED_FUSION=1 makes new scenes record without being asked (a host that cannot be recompiled).
//
// Also here: the fatal-status check (a scene that raised statusFlags refuses further calls with ITM_ERR_DEVICE) and the guards of
// itm_process_frame_ahead's pending requests.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "itm_internal.h"

namespace itm {

int g_debug_no_deferred_fusion = 0;

// One lock for the recorded calls of the whole process: a record is a few stores, a flush happens at most once per frame and render
// state, and a flush triggered from another thread (a copy into an image someone else's recorded view reads) must not race with the
// owner's own calls.
static std::recursive_mutex g_pendingMutex;
static std::vector<itm_render_state*> g_pending;      // render states with deferred.stage > 0

static void unregister(itm_render_state* rs) {
  for (size_t i = 0; i < g_pending.size(); ++i)
    if (g_pending[i] == rs) { g_pending[i] = g_pending.back(); g_pending.pop_back(); break; }
  if (rs->scene && rs->scene->deferredRs == rs) rs->scene->deferredRs = nullptr;
}

int flush_deferred(itm_render_state* rs) {
  std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
  if (!rs || !rs->deferred.stage) return ITM_OK;
  const int stage = rs->deferred.stage;
  rs->deferred.stage = 0;
  unregister(rs);
  itm_scene* s = const_cast<itm_scene*>(rs->scene);
  if (!s) return set_error(ITM_ERR_INVALID, "the render state's scene has been destroyed");      // (free_scene forgets what was recorded: belt and braces)
  const itm_view* v = &rs->deferred.view;
  hipStream_t st = rs->deferred.st;
  int rc = launch_allocate(s, v, rs, false, false, st);
  if (!rc && stage >= 2) rc = launch_integrate(s, v, rs, st, false);
  if (!rc && stage >= 3) rc = launch_expected_depths(s, v->M_d, v->intr_d, rs, false, st, false);
  return rc;
}

void forget_deferred(itm_render_state* rs) {
  std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
  if (!rs) return;
  rs->deferred.stage = 0;
  unregister(rs);
}

static bool fatal_raised(const itm_scene* s) { return s && s->fatalHost && *s->fatalHost != 0; }
static int fatal_error(const itm_scene* s) {
  const int f = (int)*s->fatalHost;
  return set_error(ITM_ERR_DEVICE, std::string("the scene raised a fatal status (itm_counters::statusFlags ") + std::to_string(f) +
                   ((f & 2) ? ": a wait between workgroups of the visible-list launch expired, the frame was not fused" : "") +
                   ((f & 1) ? ": more ray steps than the allocation key can number, blocks were not requested" : "") +
                   "); its state is not the reference's any more -- itm_reset_scene clears the condition");
}

int enter_scene(const itm_scene* s, const itm_render_state* rs) {
  if (!s && rs) s = rs->scene;
  if (fatal_raised(s)) return fatal_error(s);
  // (the records are read under the lock: flush_overlapping may rewrite them from another thread)
  std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
  int rc = ITM_OK;
  if (s && s->deferredRs) rc = flush_deferred(s->deferredRs);
  if (!rc && rs && rs->deferred.stage) rc = flush_deferred(const_cast<itm_render_state*>(rs));
  return rc;
}

int flush_overlapping(const void* p, size_t bytes, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
  int rc = ITM_OK;
  for (size_t i = 0; i < g_pending.size() && !rc;) {
    itm_render_state* rs = g_pending[i];
    const itm_view& v = rs->deferred.view;
    bool hit;
    if (!p) hit = rs->deferred.st == st;
    else {
      auto overlaps = [&](const void* q, size_t n) { return q && (const char*)q < (const char*)p + bytes && (const char*)p < (const char*)q + n; };
      hit = overlaps(v.depth, (size_t)v.w * v.h * 4) || overlaps(v.rgb, (size_t)v.w_rgb * v.h_rgb * 4);
    }
    if (hit) rc = flush_deferred(rs);      // (removes rs from the list: look at position i again)
    else ++i;
  }
  return rc;
}

int refuse_while_ahead(const itm_scene* s, const itm_render_state* rs, const char* what) {
  if (rs && rs->ahead.valid)
    return set_error(ITM_ERR_INVALID, std::string(what) + ": the render state holds the block requests of a frame issued ahead (itm_process_frame_ahead); "
                     "its visible types carry their marks until that frame is fused or itm_cancel_ahead is called");
  (void)s;
  return ITM_OK;
}

static bool deferral_enabled(const itm_scene* s) {
  static const bool off = [] { const char* e = getenv("ITM_NO_DEFERRED_FUSION"); return e && atoi(e) != 0; }();
  // dense scenes launch the same kernels either way; with swapping the mapper calls the swapping engine between the integration and
  // the ray cast (Engine/ITMDenseMapper.cpp:59-64), which would flush every frame
  return s->deferredFusion && !off && !g_debug_no_deferred_fusion && s->cfg.indexType == ITM_INDEX_HASH && !s->cfg.useSwapping;
}

bool deferred_fusion_default() {
  static const bool on = [] { const char* e = getenv("ITM_DEFERRED_FUSION"); return e && atoi(e) != 0; }();
  return on;
}

static bool same_images(const itm_view& a, const itm_view& b) { return memcmp(&a, &b, sizeof(itm_view)) == 0; }

}  // namespace itm

using namespace itm;

extern "C" {

int itm_allocate_scene_from_depth(itm_scene* s, const itm_view* v, itm_render_state* rs, int onlyUpdateVisibleList, itm_stream stream) {
  if (!s || !v || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  int rc = enter_scene(s, rs);
  if (rc) return rc;
  if (s->cfg.indexType == ITM_INDEX_DENSE) return ITM_OK;  // _CPU.cpp:314-317
  if (!v->depth) return set_error(ITM_ERR_INVALID, "null depth image");
  if (rs->scene != s || v->w != rs->w || v->h != rs->h) return set_error(ITM_ERR_INVALID, "view / render state mismatch");
  if (onlyUpdateVisibleList || !deferral_enabled(s)) return launch_allocate(s, v, rs, onlyUpdateVisibleList != 0, false, as_stream(stream));
  if ((rc = validate_allocate(s, v, rs, false))) return rc;
  std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
  rs->deferred.view = *v;
  rs->deferred.st = as_stream(stream);
  rs->deferred.stage = 1;
  s->deferredRs = rs;
  g_pending.push_back(rs);
  return ITM_OK;
}

int itm_integrate_into_scene(itm_scene* s, const itm_view* v, itm_render_state* rs, itm_stream stream) {
  if (!s || !v || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (!v->depth) return set_error(ITM_ERR_INVALID, "null depth image");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  {
    // (the lock covers the bookkeeping only: kernels are launched outside it, so that the frames of several scenes submitted from
    // several host threads do not queue up behind each other's launches)
    std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
    if (rs->deferred.stage == 1 && s->deferredRs == rs && rs->deferred.st == as_stream(stream) && same_images(rs->deferred.view, *v)) {
      if (s->fatalHost && *s->fatalHost) return enter_scene(s, nullptr);
      const int rc = validate_integrate(s, v);
      if (rc) return rc;                       // (the recorded allocation stays recorded: the call that failed did nothing)
      rs->deferred.stage = 2;
      return ITM_OK;
    }
  }
  const int rc = enter_scene(s, rs);
  if (rc) return rc;
  return launch_integrate(s, v, rs, as_stream(stream), false);
}

int itm_create_expected_depths(const itm_scene* s, const float M[16], const float intr[4], itm_render_state* rs, itm_stream stream) {
  if (!s || !M || !intr || !rs) return set_error(ITM_ERR_INVALID, "null argument");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  {
    std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
    if (rs->deferred.stage == 2 && s->deferredRs == rs && rs->deferred.st == as_stream(stream) &&
        memcmp(M, rs->deferred.view.M_d, 64) == 0 && memcmp(intr, rs->deferred.view.intr_d, 16) == 0) {
      if (s->fatalHost && *s->fatalHost) return enter_scene(s, nullptr);
      rs->deferred.stage = 3;
      return ITM_OK;
    }
  }
  const int rc = enter_scene(s, rs);
  if (rc) return rc;
  return launch_expected_depths(s, M, intr, rs, false, as_stream(stream), false);
}

int itm_create_icp_maps(const itm_scene* s, const itm_view* v, itm_render_state* rs, float* points, float* normals, itm_stream stream) {
  if (!s || !v || !rs || !points || !normals) return set_error(ITM_ERR_INVALID, "null argument");
  if (rs->scene != s) return set_error(ITM_ERR_INVALID, "render state belongs to another scene");
  bool complete = false;
  itm_view view;
  {
    std::lock_guard<std::recursive_mutex> lock(g_pendingMutex);
    if (rs->deferred.stage == 3 && s->deferredRs == rs && rs->deferred.st == as_stream(stream) && same_images(rs->deferred.view, *v)) {
      view = rs->deferred.view;
      forget_deferred(rs);
      complete = true;
    }
  }
  // the sequence is complete: the fused frame (visualise.hip), launched outside the lock
  if (complete) return itm_process_frame_ahead(const_cast<itm_scene*>(s), &view, nullptr, rs, points, normals, stream);
  const int rc = enter_scene(s, rs);
  if (rc) return rc;
  return launch_icp_maps(s, v, rs, (float4*)points, (float4*)normals, as_stream(stream));
}

int itm_scene_set_deferred_fusion(itm_scene* s, int on) {
  if (!s) return set_error(ITM_ERR_INVALID, "null scene");
  const int rc = enter_scene(s, nullptr);      // what was recorded under the old setting is launched under it
  if (rc) return rc;
  s->deferredFusion = on != 0;
  return ITM_OK;
}

int itm_flush(itm_scene* s, itm_render_state* rs, itm_stream stream) {
  (void)stream;
  if (!s && !rs) return flush_overlapping(nullptr, 0, as_stream(stream));
  return enter_scene(s, rs);
}

int itm_cancel_ahead(itm_scene* s, itm_render_state* rs, itm_stream stream) {
  if (!s || !rs || rs->scene != s) return set_error(ITM_ERR_INVALID, "scene / render state mismatch");
  const int rc = enter_scene(s, rs);
  if (rc) return rc;
  return cancel_ahead(s, rs, as_stream(stream));
}

}  // extern "C"
