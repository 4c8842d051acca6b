// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of Frame::ComputeStereoMatches (SF/src/Frame.cc:841-1011), ORBmatcher::DescriptorDistance
// (SF/src/ORBmatcher.cc:2067-2083), Frame::PosInGrid / AssignFeaturesToGrid (SF/src/Frame.cc:412-443,755-765) and
// Frame::GetFeaturesInArea (:687-753).  PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <climits>
#include <utility>

#include "orb.hpp"

namespace oracle {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // SF/src/ORBmatcher.cc:44-46

// SF/src/ORBmatcher.cc:2067-2083
inline int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct StereoResult {
    std::vector<float> uRight, depth;
    std::vector<int> bestDist;  // SAD of the accepted match, -1 if none (diagnostic)
};

// SF/src/Frame.cc:841-1011.  mbf, mb as in the Frame (mb = mbf / fx, :197).
StereoResult ComputeStereoMatches(const ORBextractor& left, const ORBextractor& right, const std::vector<KeyPoint>& keysL,
                                  const std::vector<uint8_t>& descL, const std::vector<KeyPoint>& keysR,
                                  const std::vector<uint8_t>& descR, float mbf, float mb);

// 64 x 48 feature grid (SF/include/Frame.h FRAME_GRID_COLS/ROWS; SF/src/Frame.cc:412-443, 755-765)
constexpr int FRAME_GRID_COLS = 64, FRAME_GRID_ROWS = 48;
struct FeatureGrid {
    float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<size_t> cell[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    void init(int cols, int rows);                           // ComputeImageBounds (:816-833) + Frame.cc:180-181
    void assign(const std::vector<KeyPoint>& keys);          // AssignFeaturesToGrid
    std::vector<size_t> GetFeaturesInArea(const std::vector<KeyPoint>& keys, float x, float y, float r, int minLevel,
                                          int maxLevel) const;  // :687-753
};

}  // namespace oracle
