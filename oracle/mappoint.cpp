// TEST INFRASTRUCTURE ONLY -- see mappoint.hpp.
#include "mappoint.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace oracle {

static int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
        dist += __builtin_popcount(x ^ y);
    }
    return dist;
}

int ComputeDistinctiveDescriptor(const uint8_t* d, int N) {  // MapPoint.cc:372-404
    std::vector<float> D((size_t)N * N);
    for (int i = 0; i < N; ++i) {
        D[(size_t)i * N + i] = 0;
        for (int j = i + 1; j < N; ++j) {
            const int dij = descriptor_distance(d + 32 * (size_t)i, d + 32 * (size_t)j);
            D[(size_t)i * N + j] = (float)dij;
            D[(size_t)j * N + i] = (float)dij;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; ++i) {
        std::vector<int> v(D.begin() + (size_t)i * N, D.begin() + (size_t)(i + 1) * N);
        std::sort(v.begin(), v.end());
        const int median = v[(size_t)(0.5 * (N - 1))];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    return BestIdx;
}

void UpdateNormalAndDepth(const float* c, int n, const float pos[3], const float ref[3], float level_scale, float last_scale, float normal[3],
                          float* min_distance, float* max_distance) {  // MapPoint.cc:462-501
    float acc[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
        const float v[3] = {pos[0] - c[3 * k], pos[1] - c[3 * k + 1], pos[2] - c[3 * k + 2]};
        const float nr = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int a = 0; a < 3; ++a) acc[a] = acc[a] + v[a] / nr;
    }
    const float pc[3] = {pos[0] - ref[0], pos[1] - ref[1], pos[2] - ref[2]};
    const float dist = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    *max_distance = dist * level_scale;
    *min_distance = *max_distance / last_scale;
    for (int a = 0; a < 3; ++a) normal[a] = acc[a] / (float)n;
}

}  // namespace oracle
