// The extractor handle behind tc2li_orb_* (shared by the ORB and stereo translation units).
#pragma once
#include <vector>

#include "common.hpp"
#include "orb_device.hpp"
#include "quadtree.hpp"

namespace tc2li {

struct LevelGeom {
    int w = 0, h = 0, pitch = 0;
    size_t img_stride = 0;
    int cell_begin = 0, cell_end = 0;
    int dense_off = 0;  // offset of this level's region inside the per-image slab
    int min_bx = 0, max_bx = 0, min_by = 0, max_by = 0;
};

}  // namespace tc2li

struct tc2li_orb {
    tc2li_orb_params prm{};
    int max_w = 0, max_h = 0, max_images = 0;
    bool gauss_rounded_taps = false;  // TC2LI_GAUSS_TAPS=rounded at creation: the per-tap rounded 8.8 Gaussian (orb_kernels.hip, k_blur7_strips)
    int max_cell_w = 0, max_cell_h = 0;  // largest FAST cell window of the current geometry
    tc2li::DevBuf<int> d_cell_ids;       // cells whose window fits 48 x 48, then the others (one kernel variant each)
    int n_small_cells = 0, n_large_cells = 0;
    // ctor tables (SF/src/ORBextractor.cc:388-442)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> features_per_level;
    int umax[16];

    // geometry of the current image size
    int cur_w = 0, cur_h = 0;
    std::vector<tc2li::LevelGeom> geom;
    std::vector<tc2li::FastCell> cells;
    int slab_per_image = 0;
    int kp_cap_per_image = 0;

    // device memory
    tc2li::DevBuf<uint8_t> d_level0;               // used by the host-image entry point
    std::vector<tc2li::DevBuf<uint8_t>> d_levels;  // [1..nlevels)
    std::vector<tc2li::DevBuf<uint8_t>> d_blur;    // [0..nlevels)
    std::vector<tc2li::DevBuf<int>> d_xofs, d_yofs;
    std::vector<tc2li::DevBuf<short>> d_ialpha, d_ibeta;
    tc2li::DevBuf<tc2li::FastCell> d_cells;
    tc2li::DevBuf<int> d_level_cell_begin, d_level_dense_off, d_cell_counts;
    tc2li::DevBuf<uint32_t> d_slab;
    // pinned, device-mapped host buffers: the compaction kernel writes candidates and counts straight into them and
    // the descriptor kernel reads keypoints from / writes angles and descriptors to them (no staging copies)
    // keypoint distribution on the device (k_quadtree): dense candidates per (image, level), one job each, its work space and picks;
    // the keypoints of image i occupy kp_cap_per_image slots from i * kp_cap_per_image in d_kps and in every per-keypoint array
    tc2li::DevBuf<uint32_t> d_dense, d_picked;
    tc2li::DevBuf<int> d_level_counts, d_picked_count, d_nkp, d_status;
    tc2li::DevBuf<tc2li::QuadJob> d_jobs;
    tc2li::DevBuf<uint8_t> d_qscratch;
    tc2li::DevBuf<int32_t> d_qclass;  // per chunk: the keypoint distribution's per-class job lists (launch_quadtree)
    tc2li::DevBuf<tc2li::DevKeypoint> d_kps;
    tc2li::PinnedBuf<int> h_level_counts, h_nkp, h_status;
    tc2li::PinnedBuf<tc2li::DevKeypoint> h_kps;
    tc2li::PinnedBuf<float> h_angles;
    tc2li::PinnedBuf<uint8_t> h_desc;
    // device-resident copy of the last call's features for the matchers (valid for lapping area {0,0})
    tc2li::DevBuf<tc2li::MatchKey> d_mkeys;
    tc2li::DevBuf<uint8_t> d_desc;
    tc2li::DevBuf<float> d_angles;  // keypoint angles (the matcher's rotation histogram)
    std::vector<int> last_kp_off, last_kp_cnt;  // [nimg] first slot and number of each image's keypoints in d_mkeys / d_desc
    bool last_plain_order = false;  // true when the device order equals the output order
    tc2li::ScaleTable scale_tab{};

    hipStream_t side_stream = nullptr;
    hipStream_t chunk_stream = nullptr;  // the odd chunks of a chunked batch call (round 4): their kernels overlap the even chunks' bus-bound tails
    hipEvent_t ev_fork = nullptr;
    // per chunk of a batch call (tc2li_orb_extract_batch pipelines chunks of images): 0/1 pyramid, 8/3 FAST, 3/2 compaction, 4/5 blur, 6/7 descriptors
    // 9/10 keypoint distribution
    static constexpr int kMaxChunks = 4, kEvPerChunk = 11;
    hipEvent_t ev[kMaxChunks * kEvPerChunk] = {};
    int last_chunks = 1;
    bool profiling = false;  // serialise all kernels on the caller's stream so that per-kernel event times are clean
    tc2li::LevelTable raw_tab{}, blur_tab{};
    int last_nimg = 0;
    std::vector<int> last_level_counts;  // [nimg][nlevels] candidates
    float timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    ~tc2li_orb() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (chunk_stream) (void)hipStreamDestroy(chunk_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};


namespace tc2li {
void stereo_release_workspace(const tc2li_orb* o);  // stereo_host.cpp: drops the matcher workspace of a handle
}
