// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the per-map-point refresh local mapping runs after every local BA and after creating / fusing points
// (SURVEY.md section 8f item 3):
//   MapPoint::ComputeDistinctiveDescriptors   SF/src/MapPoint.cc:338-412  (the observed descriptor with the least median
//                                             Hamming distance to the others; median = sorted[0.5 * (N - 1)], first minimum wins)
//   MapPoint::UpdateNormalAndDepth            SF/src/MapPoint.cc:444-503  (mean viewing direction, float accumulation in observation
//                                             order; scale-invariance distances from the reference keyframe)
//   ORBmatcher::DescriptorDistance            SF/src/ORBmatcher.cc (bit count of the XOR of the 256-bit descriptors)
// The observations arrive flattened in the iteration order of the point's std::map<KeyFrame*, tuple<int, int>> (left then right
// index of every keyframe that is not bad).
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <cstdint>

namespace oracle {

// descriptors: [n][32]; returns BestIdx (0 when n == 0 is never asked: the reference returns early)
int ComputeDistinctiveDescriptor(const uint8_t* descriptors, int n);
// centres: [n][3] camera centre of every observation; pos, ref_centre: [3]; level_scale = mvScaleFactors[level of the reference
// observation], last_scale = mvScaleFactors[nLevels - 1].  out: normal[3], min_distance, max_distance
void UpdateNormalAndDepth(const float* centres, int n, const float pos[3], const float ref_centre[3], float level_scale, float last_scale,
                          float normal[3], float* min_distance, float* max_distance);

}  // namespace oracle
