// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of TC2LI_SLAM::ORBextractor (SF/src/ORBextractor.cc, SF/include/ORBextractor.h).
// PARITY UNPINNED: the reference holds no test or golden vector for this class and cannot be
// built here (needs OpenCV 4.2); the OpenCV calls are restated in cv_restate.hpp.
#pragma once
#include <list>
#include <utility>

#include "cv_restate.hpp"

namespace oracle {

extern const int8_t kOrbPattern[256 * 4];  // SF/src/ORBextractor.cc:123-381

constexpr int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;  // ORBextractor.cc:45-47

struct Pt2i { int x = 0, y = 0; };

// ExtractorNode  (ORBextractor.h:32-44, ORBextractor.cc:454-510)
struct ExtractorNode {
    std::vector<KeyPoint> vKeys;
    Pt2i UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

class ORBextractor {
public:
    // ORBextractor.cc:383-443
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    // ORBextractor.cc:1060-1141.  Returns monoIndex, or -1 for an empty image.
    int extract(const Img& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors,
                const int vLappingArea[2]);

    std::vector<Img> mvImagePyramid;   // level images WITHOUT the 19-px apron (views of mvBordered)
    std::vector<Img> mvBordered;       // the apron-carrying buffers (ORBextractor.cc:1149-1151)
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;  // declared double in ORBextractor.h:105

    // exposed for unit tests
    void ComputePyramid(const Img& image);                                     // :1143-1168
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& all);     // :755-870
    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& keys, int minX, int maxX, int minY,
                                            int maxY, int N, int level);     // :529-753
    std::vector<std::vector<KeyPoint>> mvCandidates;  // per level FAST output before the quadtree (diagnostic)
};

float IC_Angle(const Img& image, float ptx, float pty, const std::vector<int>& u_max);       // :50-77
void computeOrbDescriptor(const KeyPoint& kpt, const Img& img, const int8_t* pattern, uint8_t* desc);  // :81-120

}  // namespace oracle
