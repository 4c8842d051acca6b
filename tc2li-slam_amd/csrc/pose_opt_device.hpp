// Shared between the host orchestration and the pose-optimisation kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_math.hpp"

namespace tc2li {

struct PoseProblem { int32_t edge_off, n; };  // one frame: its correspondences are edges[edge_off .. edge_off + n)

void launch_pose_optimization(const PoseProblem* probs, int nprobs, const double* Xw, const BaEdge* edges, const CameraD& cam,
                              double* poses7, uint8_t* outlier, double* chi2_scratch, int* inliers, int max_edges /* upper bound of PoseProblem::n */,
                              hipStream_t st);

}  // namespace tc2li
