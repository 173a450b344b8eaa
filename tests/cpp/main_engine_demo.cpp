// Drives ITMMainEngine_HIP (include/itm_hip_engines.hpp) -- the reference's ITMMainEngine::ProcessFrame with its switches, its
// tracking controller and dense mapper (Engine/ITMMainEngine.cpp:111-127,194-197, ITMTrackingController.cpp:11-46,
// ITMDenseMapper.cpp:50-58) -- over a sequence of raw depth frames read from a file, and prints one JSON line per frame
// (age_pointCloud, requiresFullRendering, pose, digests of the tracking maps); tests/test_main_engine.py compares them with
// the same sequence run on the reference's objects.
//   main_engine_demo <sequence file>          parity run
//   main_engine_demo --bench <frames>         closed tracking + mapping loop on the 640x480 bench scene, frames/s
//   main_engine_demo --bench-host <frames>    the same with every raw frame arriving from page-locked host memory (ProcessFrameFromHost,
//                                             the next frame uploaded while the current one is tracked and fused)
//   main_engine_demo --bench-map[-host] <frames>   THIS FORK'S DEFAULT pipeline: TRACKER_EXTERNAL (Utils/ITMLibSettings.cpp:44) -- the pose source
//                                             writes pose_d before every frame (Engine/RosPoseSourceEngine.cpp:112-118), ITMMainEngine::ProcessFrame
//                                             builds the view from the raw frame, the tracker is a no-op, mapper + Prepare fuse and ray-cast
// sequence file: int32 {w, h, n, trackerType, useApproximateRaycast, skipPoints, hasPoses}, float intr[4], int16 raw[n*h*w],
//                float poses[n*16] (if hasPoses), uint8 fusion[n], uint8 mainProcessing[n]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "itm_hip_engines.hpp"

using namespace itmhip;
typedef ITMVoxel_s V;
typedef ITMVoxelBlockHash I;

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
template <class T> static bool rd(FILE* f, T* dst, size_t n) { return fread(dst, sizeof(T), n, f) == n; }

static ITMLibSettings settings_for(int trackerType, bool approx, bool skip) {
  ITMLibSettings st;
  st.trackerType = trackerType == 0 ? ITMLibSettings::TRACKER_COLOR : trackerType == 1 ? ITMLibSettings::TRACKER_ICP : ITMLibSettings::TRACKER_EXTERNAL;
  st.useApproximateRaycast = approx; st.skipPoints = skip;
  return st;
}

static int bench(int frames, bool fromHost, bool externalPoses = false) {
  // the bench scene: sphere of radius 0.5 m at (0, 0, 1.5) in front of a wall at 2.5 m, triangle-wave trajectory
  const int W = 640, H = 480, P = W * H, distinct = 100;
  std::vector<int16_t> raw((size_t)distinct * P);
  auto tri = [](int k) { return std::abs(((k + 25) % 100) - 50) - 25; };
  for (int k = 0; k < distinct; ++k) {
    const float tx = 0.004f * (float)tri(k), ty = 0.002f * (float)tri(2 * k);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      const float dx = ((float)x - 320.f) / 580.f, dy = ((float)y - 240.f) / 580.f, ox = tx, oy = ty, oz = -1.5f;
      const float A = dx * dx + dy * dy + 1.f, B = 2.f * (ox * dx + oy * dy + oz), C = ox * ox + oy * oy + oz * oz - 0.25f, disc = B * B - 4.f * A * C;
      float z = 2.5f;
      if (disc > 0) { const float t = (-B - std::sqrt(disc)) / (2.f * A); if (t > 0) z = t; }
      raw[(size_t)k * P + y * W + x] = (int16_t)(z * 1000.f);
    }
  }
  void* dRaw; check(itm_dev_malloc(&dRaw, raw.size() * 2), "malloc");
  check(itm_memcpy_h2d(dRaw, raw.data(), raw.size() * 2, nullptr), "h2d");
  ITMLibSettings st = settings_for(externalPoses ? 2 : 1, false, true);
  ITMSceneParams params(0.02f, 100, 0.004f, 0.35f, 3.0f, false);
  ITMRGBDCalib calib;
  ITMMainEngine_HIP<V, I> engine(st, params, calib, Vector2i{W, H}, Vector2i{W, H}, 1, 0.001f, 0.0f, 0x40000);
  void* hRaw = nullptr;
  if (fromHost) { check(itm_host_malloc(&hRaw, raw.size() * 2), "host malloc"); memcpy(hRaw, raw.data(), raw.size() * 2); }
  auto frame = [&](int k, bool more) {
    if (externalPoses) {      // the pose source of this fork: world -> camera for a camera at t, identity rotation
      float M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      M[12] = -(0.004f * (float)tri(k)); M[13] = -(0.002f * (float)tri(2 * k));
      engine.GetTrackingState()->pose_d.SetM(M);
    }
    if (!fromHost) { engine.ProcessFrame(nullptr, (const int16_t*)dRaw + (size_t)(k % distinct) * P); return; }
    engine.ProcessFrameFromHost(nullptr, (const int16_t*)hRaw + (size_t)(k % distinct) * P, more ? (const int16_t*)hRaw + (size_t)((k + 1) % distinct) * P : nullptr);
  };
  for (int k = 0; k < 5; ++k) frame(k, true);
  check(itm_stream_synchronize(nullptr), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 5; k < 5 + frames; ++k) frame(k, k + 1 < 5 + frames);
  check(itm_stream_synchronize(nullptr), "sync");
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const float* M = engine.GetTrackingState()->pose_d.GetM();
  const int last = 4 + frames;
  const float ex = -0.004f * (float)tri(last), ey = -0.002f * (float)tri(2 * last);
  printf("{\"frames\": %d, \"tracker\": \"%s\", \"raw_frames_from\": \"%s\", \"pose\": [%.9g, %.9g, %.9g], \"fps\": %.1f, \"ms_per_frame\": %.4f, \"final_translation_error_m\": %.5f}\n", frames, externalPoses ? "external poses (TRACKER_EXTERNAL, the fork's default): mapping only" : "ICP (closed loop)", fromHost ? "pinned host memory (stager)" : "device memory", M[12], M[13], M[14], frames / dt, 1e3 * dt / frames,
         std::fmax(std::fabs(M[12] - ex), std::fmax(std::fabs(M[13] - ey), std::fabs(M[14]))));
  itm_dev_free(dRaw);
  if (hRaw) itm_host_free(hRaw);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "--bench")) return bench(atoi(argv[2]), false);
  if (argc >= 3 && !strcmp(argv[1], "--bench-host")) return bench(atoi(argv[2]), true);
  if (argc >= 3 && !strcmp(argv[1], "--bench-map")) return bench(atoi(argv[2]), false, true);
  if (argc >= 3 && !strcmp(argv[1], "--bench-map-host")) return bench(atoi(argv[2]), true, true);
  if (argc < 2) { fprintf(stderr, "usage: %s <sequence file> | --bench <frames>\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int32_t hd[7]; float intr[4];
  if (!rd(f, hd, 7) || !rd(f, intr, 4)) return 2;
  const int W = hd[0], H = hd[1], N = hd[2], P = W * H;
  std::vector<int16_t> raw((size_t)N * P); std::vector<float> poses((size_t)N * 16); std::vector<uint8_t> fusion(N), mainOn(N);
  if (!rd(f, raw.data(), raw.size()) || (hd[6] && !rd(f, poses.data(), poses.size())) || !rd(f, fusion.data(), (size_t)N) || !rd(f, mainOn.data(), (size_t)N)) return 2;
  fclose(f);

  ITMLibSettings st = settings_for(hd[3], hd[4] != 0, hd[5] != 0);
  ITMSceneParams params(0.02f, 100, 0.005f, 0.35f, 3.0f, false);      // ITMLibSettings.cpp:10
  ITMRGBDCalib calib;
  calib.intrinsics_d.SetFrom(intr[0], intr[1], intr[2], intr[3]);
  calib.intrinsics_rgb = calib.intrinsics_d;
  ITMMainEngine_HIP<V, I> engine(st, params, calib, Vector2i{W, H}, Vector2i{W, H});
  void *dRaw, *dRgb;
  check(itm_dev_malloc(&dRaw, (size_t)P * 2), "malloc"); check(itm_dev_malloc(&dRgb, (size_t)P * 4), "malloc");
  std::vector<uint8_t> rgb((size_t)P * 4);
  for (int i = 0; i < P; ++i) { rgb[4 * i] = (uint8_t)(i % W); rgb[4 * i + 1] = (uint8_t)(i / W); rgb[4 * i + 2] = (uint8_t)((i % W) ^ (i / W)); rgb[4 * i + 3] = 255; }
  check(itm_memcpy_h2d(dRgb, rgb.data(), rgb.size(), nullptr), "h2d");
  std::vector<float> pts((size_t)P * 4), col((size_t)P * 4); std::vector<uint8_t> img((size_t)P * 4); std::vector<int32_t> ids;
  for (int k = 0; k < N; ++k) {
    check(itm_memcpy_h2d(dRaw, raw.data() + (size_t)k * P, (size_t)P * 2, nullptr), "h2d");
    ITMTrackingState* ts = engine.GetTrackingState();
    if (hd[6]) ts->pose_d.SetM(poses.data() + 16 * k);            // the pose source of this fork writes it before the frame
    if (fusion[k]) engine.turnOnIntegration(); else engine.turnOffIntegration();
    if (mainOn[k]) engine.turnOnMainProcessing(); else engine.turnOffMainProcessing();
    engine.ProcessFrame((const uint8_t*)dRgb, (const int16_t*)dRaw);
    itm_counters c;
    check(itm_get_counters(engine.GetScene()->handle, engine.GetRenderState()->handle, &c, nullptr), "counters");
    const size_t nPts = hd[3] == 0 ? (size_t)c.noTotalPoints : (size_t)P;
    check(itm_memcpy_d2h(pts.data(), ts->pointCloud_locations, nPts * 16, nullptr), "d2h");
    check(itm_memcpy_d2h(col.data(), ts->pointCloud_colours, nPts * 16, nullptr), "d2h");
    check(itm_download(engine.GetScene()->handle, engine.GetRenderState()->handle, ITM_BUF_RAYCAST_IMAGE, img.data(), img.size(), nullptr), "download");
    ids.resize((size_t)c.noVisibleEntries);
    if (c.noVisibleEntries) check(itm_download(engine.GetScene()->handle, engine.GetRenderState()->handle, ITM_BUF_VISIBLE_IDS, ids.data(), ids.size() * 4, nullptr), "download");
    const int32_t nv = c.noVisibleEntries;
    printf("{\"k\": %d, \"age\": %d, \"full\": %d, \"pose\": [", k, ts->age_pointCloud, ts->requiresFullRendering ? 1 : 0);
    for (int i = 0; i < 16; ++i) printf("%s%.9g", i ? ", " : "", ts->pose_d.GetM()[i]);
    printf("], \"digest\": [\"%016llx\", \"%016llx\", \"%016llx\", \"%016llx\"], \"visible\": %d, \"points\": %zu}\n",
           (unsigned long long)fnv(pts.data(), nPts * 16), (unsigned long long)fnv(col.data(), nPts * 16), (unsigned long long)fnv(img.data(), img.size()),
           (unsigned long long)fnv(ids.data(), ids.size() * 4, fnv(&nv, 4)), nv, nPts);
  }
  itm_dev_free(dRaw); itm_dev_free(dRgb);
  return 0;
}
