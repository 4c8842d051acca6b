// Host-side keypoint distribution of the ORB extractor: same selection and output order as
// ORBextractor::DistributeOctTree (SF/src/ORBextractor.cc:529-753), built on flat index pools instead of
// std::list<ExtractorNode> with per-node keypoint vectors.
#pragma once
#include <cstdint>
#include <vector>

namespace tc2li {

struct QuadtreeScratch {
    struct Node {
        int ulx, uly, brx, bry;  // UL and BR corners; UR = (brx, uly), BL = (ulx, bry)
        int begin, count;        // key index range in `keys`
        int prev, next;          // list links (node ids), -1 = none
        bool no_more;
    };
    std::vector<Node> nodes;
    std::vector<int32_t> keys;  // pool of candidate indices, children ranges appended
    std::vector<std::pair<int, int>> expand, prev_expand;  // (size, node id)
};

// Candidates are packed y<<20 | x<<8 | response (border-free level coordinates), in FAST emission order.
// Appends the index of the retained candidate of every final node to `out`, in final list order.
void distribute_quadtree(const uint32_t* cand, int ncand, int min_x, int max_x, int min_y, int max_y, int n_target,
                         QuadtreeScratch& scratch, std::vector<int32_t>& out);

}  // namespace tc2li
