// icp_solver.h -- host side of the ICP depth tracker: damped Gauss-Newton over SE(3).  Plain C++ (no HIP), shared by
// tracker.hip (cost / gradient / Hessian evaluated on the GPU) and by the CPU check in tests/cpp/icp_solver_check.cpp.
#pragma once

#include <cmath>
#include <cstring>

#include "../../include/itm_hip.h"
#include "../../include/itm_debug.h"      // itm_icp_evaluate_fn (the host-only test hook drives this solver)
#include "se3.h"

namespace itm {

// Behaviour of ITMDepthTracker::TrackCamera (Engine/ITMDepthTracker.cpp:149-200): per hierarchy level, coarse to fine, up
// to 2(l+1) iterations of { evaluate cost / gradient / Hessian at the current pose; if the cost rose or nothing was valid,
// go back to the last accepted pose and multiply the damping by 10, else accept and divide it by 10; solve
// (H + damping diag H) x = g for the active 3 or 6 parameters; left-multiply the inverse pose by the first-order motion
// I - [x_rot]x, x_trans and project the result back onto SE(3); stop when |x| / 6 < threshold }.
// Own formulation: poses and the solve in double (se3.h), the projection as exp(log(.)).
struct IcpSolver {
  se3::Rigid pose, accepted;          // world -> camera: current estimate, last estimate that lowered the cost
  double H[36], g[6];                 // normal equations of the accepted estimate (per valid point), kept across levels
  double damping = 1.0;
  float acceptedCost = 1e20f;

  void begin_level() { accepted = pose; damping = 1.0; acceptedCost = 1e20f; }

  void inverse_pose(float out16[16]) const {
    se3::Rigid inv;
    if (!se3::invert(pose, inv)) inv = pose;
    se3::to_matrix(inv, out16);
  }

  // returns the length criterion sqrt(sum x^2) / 6 of the step taken
  double iterate(const itm_tracker_gh& e, int mode) {
    if (e.noValidPoints <= 0 || e.f > acceptedCost) {
      pose = accepted;
      damping *= 10.0;
    } else {
      accepted = pose;
      acceptedCost = e.f;
      const double inv = 1.0 / (double)e.noValidPoints;
      for (int i = 0; i < 36; ++i) H[i] = (double)e.hessian[i] * inv;
      for (int i = 0; i < 6; ++i) g[i] = (double)e.nabla[i] * inv;
      damping /= 10.0;
    }
    const int n = (mode == ITM_TRACKER_ITERATION_BOTH) ? 6 : 3;
    double A[36], x[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 36; ++i) A[i] = H[i];
    for (int i = 0; i < 6; ++i) A[7 * i] *= 1.0 + damping;
    se3::solve_spd(A, 6, n, g, x);
    double rot[3] = {0, 0, 0}, trans[3] = {0, 0, 0};
    if (mode == ITM_TRACKER_ITERATION_ROTATION) { rot[0] = x[0]; rot[1] = x[1]; rot[2] = x[2]; }
    else if (mode == ITM_TRACKER_ITERATION_TRANSLATION) { trans[0] = x[0]; trans[1] = x[1]; trans[2] = x[2]; }
    else { for (int i = 0; i < 3; ++i) { rot[i] = x[i]; trans[i] = x[3 + i]; } }
    // first-order motion applied on the left of the inverse pose ...
    se3::Rigid motion;
    motion.R[0] = 1.0;     motion.R[1] = rot[2];  motion.R[2] = -rot[1];
    motion.R[3] = -rot[2]; motion.R[4] = 1.0;     motion.R[5] = rot[0];
    motion.R[6] = rot[1];  motion.R[7] = -rot[0]; motion.R[8] = 1.0;
    for (int i = 0; i < 3; ++i) motion.t[i] = trans[i];
    se3::Rigid inv, moved, back;
    if (se3::invert(pose, inv)) {
      moved = se3::compose(motion, inv);
      // ... and the result, which is no longer a rigid motion, projected back onto SE(3)
      if (se3::invert(moved, back)) pose = se3::exp(se3::log(back));
    }
    double len = 0.0;
    for (int i = 0; i < 6; ++i) len += x[i] * x[i];
    return std::sqrt(len) / 6.0;
  }
};


// The level schedule of the ITMDepthTracker constructor (Engine/ITMDepthTracker.cpp:18-44): levels coarse to fine down to
// noICPRunTillLevel, 2 (l + 1) iterations on level l, distance threshold falling linearly from the coarsest level.
// `evaluate(level, mode, inversePose16, distThresh, out)` returns the cost / gradient / Hessian at a pose (0 = ok).
template <class Evaluate>
inline int icp_track(const itm_tracker_config* cfg, const float M_d_in[16], float M_d_out[16], Evaluate&& evaluate) {
  const int levels = cfg->noHierarchyLevels;
  IcpSolver solver;
  solver.pose = se3::from_matrix(M_d_in);
  std::memset(solver.H, 0, sizeof solver.H); std::memset(solver.g, 0, sizeof solver.g);
  for (int level = levels - 1; level >= cfg->noICPRunTillLevel; --level) {
    const int mode = cfg->trackingRegime[level];
    if (mode == ITM_TRACKER_ITERATION_NONE) continue;
    const int maxIterations = 2 * (level + 1);
    float distThresh = cfg->distThresh;
    for (int l = levels - 1; l > level; --l) distThresh -= cfg->distThresh / (float)levels;
    solver.begin_level();
    for (int k = 0; k < maxIterations; ++k) {
      float invPose[16];
      solver.inverse_pose(invPose);
      itm_tracker_gh e;
      const int rc = evaluate(level, mode, invPose, distThresh, &e);
      if (rc) return rc;
      if (solver.iterate(e, mode) < (double)cfg->terminationThreshold) break;
    }
  }
  se3::to_matrix(solver.pose, M_d_out);
  return 0;
}

}  // namespace itm
