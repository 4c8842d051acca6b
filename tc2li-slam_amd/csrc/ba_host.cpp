// Host side of the optimisation entry points (include/tc2li_hip.h): tc2li_pose_optimization[_batch] replaces
// Optimizer::PoseOptimization (SF/src/Optimizer.cc:816-1116).
#include <algorithm>
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "pose_opt_device.hpp"

using namespace tc2li;

static_assert(sizeof(tc2li_ba_edge) == sizeof(BaEdge), "ABI layout");
static_assert(sizeof(tc2li_camera) == sizeof(CameraD), "ABI layout");

namespace {

struct PoseOptWorkspace {
    DevBuf<PoseProblem> d_probs;
    DevBuf<double> d_Xw, d_poses, d_chi2;
    DevBuf<BaEdge> d_edges;
    DevBuf<uint8_t> d_outlier;
    DevBuf<int> d_inliers;
    std::mutex mu;
};
PoseOptWorkspace& po_ws() { static PoseOptWorkspace w; return w; }

}  // namespace

extern "C" {

int tc2li_pose_optimization_batch(int n_frames, double* poses7, const int32_t* edge_offsets, const double* Xw,
                                  const tc2li_ba_edge* edges, const tc2li_camera* cam, uint8_t* outlier, int32_t* n_inliers,
                                  void* stream_) {
    if (n_frames < 0 || !poses7 || !edge_offsets || !cam || !n_inliers) { set_error("tc2li_pose_optimization_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n_frames == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    const int total = edge_offsets[n_frames];
    if (total < 0 || (total > 0 && (!Xw || !edges || !outlier))) { set_error("tc2li_pose_optimization_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream_;
    PoseOptWorkspace& w = po_ws();
    std::lock_guard<std::mutex> lk(w.mu);
    std::vector<PoseProblem> probs(n_frames);
    for (int f = 0; f < n_frames; ++f) {
        probs[f] = PoseProblem{edge_offsets[f], edge_offsets[f + 1] - edge_offsets[f]};
        if (probs[f].n < 0) { set_error("edge offsets must be non-decreasing"); return TC2LI_ERR_INVALID; }
    }
    TC2LI_HIP_CHECK(w.d_probs.ensure(n_frames));
    TC2LI_HIP_CHECK(w.d_poses.ensure((size_t)7 * n_frames));
    TC2LI_HIP_CHECK(w.d_inliers.ensure(n_frames));
    TC2LI_HIP_CHECK(w.d_Xw.ensure(std::max(3 * (size_t)total, (size_t)1)));
    TC2LI_HIP_CHECK(w.d_edges.ensure(std::max(total, 1)));
    TC2LI_HIP_CHECK(w.d_outlier.ensure(std::max(total, 1)));
    TC2LI_HIP_CHECK(w.d_chi2.ensure(std::max(total, 1)));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_probs.p, probs.data(), n_frames * sizeof(PoseProblem), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_poses.p, poses7, (size_t)7 * n_frames * sizeof(double), hipMemcpyHostToDevice, st));
    if (total) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_Xw.p, Xw, 3 * (size_t)total * sizeof(double), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_edges.p, edges, (size_t)total * sizeof(BaEdge), hipMemcpyHostToDevice, st));
    }
    CameraD c;
    memcpy(&c, cam, sizeof(c));
    launch_pose_optimization(w.d_probs.p, n_frames, w.d_Xw.p, w.d_edges.p, c, w.d_poses.p, w.d_outlier.p, w.d_chi2.p, w.d_inliers.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(poses7, w.d_poses.p, (size_t)7 * n_frames * sizeof(double), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(n_inliers, w.d_inliers.p, n_frames * sizeof(int), hipMemcpyDeviceToHost, st));
    if (total) TC2LI_HIP_CHECK(hipMemcpyAsync(outlier, w.d_outlier.p, total, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    return n_frames;
}

int tc2li_pose_optimization(double pose7[7], const double* Xw, const tc2li_ba_edge* edges, int n, const tc2li_camera* cam,
                            uint8_t* outlier) {
    if (n < 0) { set_error("tc2li_pose_optimization: invalid argument"); return TC2LI_ERR_INVALID; }
    const int32_t offs[2] = {0, n};
    int32_t inl = 0;
    int rc = tc2li_pose_optimization_batch(1, pose7, offs, Xw, edges, cam, outlier, &inl, nullptr);
    if (rc < 0) return rc;
    return inl;
}

}  // extern "C"
