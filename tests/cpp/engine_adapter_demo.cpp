// Drives the HIP back-end through the C++ adapter classes exactly like ITMMainEngine::ProcessFrame
// drives the reference engines (Engine/ITMMainEngine.cpp:111-127), on a flat wall at 1.5 m.
// Prints counters and checksums as JSON; tests/test_cpp_adapter.py compares them with the oracle.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "itm_hip_engines.hpp"

using namespace itmhip;
typedef ITMVoxel_s V;
typedef ITMVoxelBlockHash I;

int main() {
  const int W = 160, H = 120, P = W * H;
  ITMSceneParams params(0.02f, 100, 0.01f, 0.35f, 3.0f, false);
  ITMScene<V, I> scene(&params);
  ITMDenseMapper_HIP<V, I> mapper;
  ITMVisualisationEngine_HIP<V, I> vis(&scene);
  ITMTrackingController_HIP<V, I> controller(&vis);
  mapper.ResetScene(&scene);
  ITMRenderState* rs = vis.CreateRenderState(Vector2i{W, H});

  std::vector<float> depth(P, 1.5f);
  void *dDepth, *dPts, *dNrm;
  check(itm_dev_malloc(&dDepth, P * 4), "malloc"); check(itm_dev_malloc(&dPts, P * 16), "malloc"); check(itm_dev_malloc(&dNrm, P * 16), "malloc");
  check(itm_memcpy_h2d(dDepth, depth.data(), P * 4, nullptr), "h2d");

  ITMView view;
  view.calib.intrinsics_d.SetFrom(145.f, 145.f, 80.f, 60.f);
  view.calib.intrinsics_rgb = view.calib.intrinsics_d;
  view.depth = (const float*)dDepth; view.depthSize = Vector2i{W, H}; view.rgbSize = Vector2i{W, H};
  ITMTrackingState ts;
  ts.pointCloud_locations = (float*)dPts; ts.pointCloud_colours = (float*)dNrm;

  for (int k = 0; k < 2; ++k) {
    float M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -0.01f * k, 0, 0, 1};
    ts.pose_d.SetM(M);                       // external pose, as RosPoseSourceEngine does
    controller.Track(&ts, &view);
    mapper.ProcessFrame(&view, &ts, &scene, rs);
    controller.Prepare(&ts, &view, rs);
  }
  itm_counters c;
  check(itm_get_counters(scene.handle, rs->handle, &c, nullptr), "counters");
  std::vector<float> pts(P * 4);
  check(itm_memcpy_d2h(pts.data(), dPts, P * 16, nullptr), "d2h");
  check(itm_stream_synchronize(nullptr), "sync");
  double sx = 0, sz = 0; long valid = 0;
  for (int i = 0; i < P; ++i) if (pts[4 * i + 3] > 0) { ++valid; sx += pts[4 * i]; sz += pts[4 * i + 2]; }
  // ITMMainEngine::UpdateMesh (Engine/ITMMainEngine.cpp:129-135): mesh the scene, read the triangles back
  ITMMesh mesh(&scene);
  ITMMeshingEngine_HIP<V, I> mesher;
  mesher.MeshScene(&mesh, &scene);
  std::vector<float> tri((size_t)mesh.noTotalTriangles * 9);
  uint32_t n = 0;
  check(itm_mesh_download(mesh.handle, tri.data(), mesh.noTotalTriangles, &n, nullptr), "mesh download");
  double tz = 0;
  for (size_t i = 2; i < tri.size(); i += 3) tz += tri[i];
  printf("{\"lastFreeBlockId\": %d, \"noVisibleEntries\": %d, \"valid\": %ld, \"sum_x\": %.9g, \"sum_z\": %.9g, \"age\": %d, "
         "\"triangles\": %u, \"maxTriangles\": %u, \"tri_sum_z\": %.9g}\n",
         c.lastFreeBlockId, c.noVisibleEntries, valid, sx, sz, ts.age_pointCloud, mesh.noTotalTriangles, mesh.noMaxTriangles, tz);
  delete rs;
  itm_dev_free(dDepth); itm_dev_free(dPts); itm_dev_free(dNrm);
  return 0;
}
