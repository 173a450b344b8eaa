/*
 * itm_debug.h -- test hooks of libitmhip.so.  NOT part of the drop-in boundary (include/itm_hip.h): nothing here changes a result;
 * the keys select alternative code paths of the library so that every one of them has a parity test, and the probes expose device
 * routines (the reduced division sequences, the dense cull) and the tracker's host iteration to tests that check them in isolation.
 * Hosts do not include this header.
 */
#ifndef ITM_DEBUG_H_
#define ITM_DEBUG_H_

#include "itm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hooks (no effect on results): select alternative code paths so that they can be covered. */
#define ITM_DEBUG_FORCE_GLOBAL_RANGE_ATOMICS 1 /* range image via global atomics even if it fits LDS */
#define ITM_DEBUG_EXPLICIT_MARK_PREVIOUS 2     /* always run the separate mark-previous-list launch   */
#define ITM_DEBUG_INTEGRATE_WORKGROUPS 3      /* tuning: persistent workgroups of the hash integration (0 = default) */
#define ITM_DEBUG_NO_FUSED_PROJECTION 4       /* process_frame: keep integration and projection as two launches */
#define ITM_DEBUG_NO_DIRECTORY 5              /* ray casting / free-view reads walk the hash table instead of the block directory */
#define ITM_DEBUG_NO_FUSED_RANGE_REDUCE 6     /* process_frame: reduce the partial range images in their own launch, not in the ray cast */
#define ITM_DEBUG_TWO_PASS_VISIBLE_LIST 7     /* AllocateSceneFromDepth: visible list by a count launch and a compaction launch */
#define ITM_DEBUG_SINGLE_PASS_RAYCAST 8       /* ray casting: every ray start to finish in one launch (no parked-ray pass) */
#define ITM_DEBUG_DENSE_GROUP_CULL 9          /* dense integration: frustum test per 4-voxel group instead of the per-column row interval */
#define ITM_DEBUG_TRACKER_LAUNCH_PER_EVALUATION 10 /* TrackCamera: one launch per cost evaluation instead of one resident kernel per call */
#define ITM_DEBUG_TRACKER_HOST_COMMAND 11     /* TrackCamera session: commands through pinned host memory (set before the tracker's first call) */
#define ITM_DEBUG_SEPARATE_SWEEP 13           /* AllocateSceneFromDepth: allocation sweep as its own launch, not inside the visible-list launch */
#define ITM_DEBUG_NO_SIDE_PROJECTION 14       /* itm_process_frame on large images: projection of the visible blocks after the integration on the frame's stream, not beside it on the render state's own */
#define ITM_DEBUG_DENSE_RANGE_REFILL 15       /* dense scenes: write the constant expected-depth image on every frame, as the reference does, although it already holds it */
#define ITM_DEBUG_NO_SDF_MIRROR 12            /* ray casting: voxels through the directory / table although the scene has an sdf mirror; set before itm_scene_create: no mirror is allocated */
#define ITM_DEBUG_DENSE_CLASSIFY 16            /* dense integration: 0 = 4-voxel groups classified against the depth tiles before the fetch (default), 1 = no classification, 2 = classified after the fetch, 3 = check mode (itm_debug_dense_classify_check) */
#define ITM_DEBUG_DENSE_NO_STRIPS 17           /* dense integration: the launch shape of rounds 1-2 (four groups per lane, 131 072 short waves) instead of the strip kernel */
#define ITM_DEBUG_TRACKER_SESSION_UNUSABLE 18  /* TrackCamera: the resident evaluation kernel reports itself unusable at the n-th evaluation of a handle (n = value): the call must finish through one launch per evaluation with the same pose */
#define ITM_DEBUG_NO_DEFERRED_FUSION 19         /* the four per-frame engine calls launch at once, one by one, instead of being recorded and fused (see "the four calls" below) */
#define ITM_DEBUG_FORCE_LIST_STUCK 20           /* AllocateSceneFromDepth, one-launch visible list: chunk n - 1 behaves as if its bounded wait for another workgroup had expired (0 = off): the scene must raise statusFlags bit 1 and refuse further calls */
#define ITM_DEBUG_EXCHANGE_DEVICE_COPY 24        /* exchanges created while it is set: a ONE-rank exchange performs its collective as a device copy instead of ncclAllGather (a test pins both to the same table) */
#define ITM_DEBUG_EXCHANGE_CORRUPT_WORD 25       /* exchanges created while it is >= 0: the self-check behind every collective sees this word of the rank's own block flipped (the check must fire); -1 = off */
int ITM_FN(debug_set)(int key, int value);
/* A stand-in for a device-side consumer of the exchange's table (tests of itm_exchange_acquire): dst[0] = sum over `rounds` passes of a
 * position-weighted checksum of src[0 .. words), computed by ONE workgroup on `stream` -- slow on purpose, so that collectives of later
 * batches run while it reads. */
int ITM_FN(debug_checksum)(const int32_t* src, size_t words, int rounds, unsigned long long* dst, itm_stream stream);
/* dense integration, check mode of key 16: {free groups, shadow groups, mixed groups, violations}; reset != 0 clears */
int ITM_FN(debug_dense_classify_check)(int32_t out[4], int reset);
/* Test hook (host only): rows [rlo, rhi] of the column of 4-voxel groups (x0 .. x0 + 3, slice z) that the dense integration visits for
 * a volume of `size` voxels at `offset` seen from M_d; every voxel of the column outside that interval must fail the exact
 * projection test of computeUpdatedVoxelDepthInfo.  Returns 1 when no cull planes can be formed (the kernel then tests per group). */
int ITM_FN(debug_column_cull_rows)(const float M_d[16], const float intr[4], int w, int h, float voxelSize, const int size[3],
                                   const int offset[3], int x0, int z, int* rlo, int* rhi);
/* out[i] = SDF_valueToFloat(in[i]) of the short voxel types, i.e. in[i] / 32767.0f, through the same
 * device routine the kernels use (a 3-instruction correctly rounded division; test hook). */
int ITM_FN(debug_div32767)(const float* in, float* out, int n, itm_stream stream);
/* out[i] = a[i] / b[i] through the reduced division sequences of the integration kernel (mode 1: shared
 * refined reciprocal; 2: small-integer divisor; 3: the refined reciprocal of b; 4: reciprocal given in r). */
int ITM_FN(debug_divide)(int mode, const float* a, const float* b, const float* r, float* out, int n, itm_stream stream);


/* Test hook (host only, no device work): the tracker's host-side iteration (level schedule, accept / reject damping, SE(3)
 * update) driven by a caller-supplied evaluator of cost / gradient / Hessian, so that it can be checked on a machine without
 * a GPU against ITMDepthTracker::TrackCamera with the same evaluator.  `evaluate` returns 0 on success. */
typedef int (*itm_icp_evaluate_fn)(void* user, int level, int iterationType, const float approxInvPose[16],
                                   float distThresh, itm_tracker_gh* out);
int ITM_FN(debug_icp_track)(const itm_tracker_config* cfg, const float M_d[16], itm_icp_evaluate_fn evaluate,
                            void* user, float M_d_out[16]);

#ifdef __cplusplus
}
#endif
#endif /* ITM_DEBUG_H_ */
