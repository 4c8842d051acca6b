// Host side of the optimisation entry points (include/tc2li_hip.h): tc2li_pose_optimization[_batch] replaces
// Optimizer::PoseOptimization (SF/src/Optimizer.cc:816-1116).
#include "ba_internal.hpp"

using namespace tc2li;
using namespace tc2li::ba_detail;

static_assert(sizeof(tc2li_ba_edge) == sizeof(BaEdge), "ABI layout");
static_assert(sizeof(tc2li_camera) == sizeof(CameraD), "ABI layout");

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
    int max_edges = 0;
    for (int f = 0; f < n_frames; ++f) max_edges = std::max(max_edges, probs[f].n);
    launch_pose_optimization(w.d_probs.p, n_frames, w.d_Xw.p, w.d_edges.p, c, w.d_poses.p, w.d_outlier.p, w.d_chi2.p, w.d_inliers.p, max_edges, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(poses7, w.d_poses.p, (size_t)7 * n_frames * sizeof(double), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(n_inliers, w.d_inliers.p, n_frames * sizeof(int), hipMemcpyDeviceToHost, st));
    if (total) TC2LI_HIP_CHECK(hipMemcpyAsync(outlier, w.d_outlier.p, total, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
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

// OptimizerWithLidar::LocalLVBundleAdjustment / Optimizer::LocalBundleAdjustment, visual part: the optimisation between
// "Setup optimizer" and "Check inlier observations" (SF/src/OptimizerWithLidar.cc:132-400).  The Levenberg-Marquardt
// control flow of g2o (optimization_algorithm_levenberg.cpp:61-169) runs here on the host; every numerical step is a
// kernel of ba_kernels.hip; the reduced camera system (6 x free poses) is factorised on the host (LDL^T), as g2o's
// LinearSolverEigen does.
//
// shard != NULL: this rank holds the landmarks (points3, edges) it owns of a window split over several GPUs
// (tc2li_local_lv_bundle_adjustment_sharded below); used_all marks the poses any rank's edges touch, so that the free-pose
// numbering is the same everywhere.  The kernels then write their sums to device memory, the ranks' parts are added by the
// caller's all-reduce, and only the sum comes to the host.
static int lv_ba_impl(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                      const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                      double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                      uint8_t* edge_depth_positive, tc2li_ba_stats* stats, const tc2li_lidar_window* lidar_window,
                      tc2li_lidar_ba_stats* lidar_stats, const tc2li_ba_shard* shard, const uint8_t* used_all, void* stream_,
                      int* shard_exit = nullptr) {
    if (!poses7 || !fixed || !points3 || !edges || !cam || n_poses <= 0 || n_points <= 0 || n_edges <= 0 || iterations < 0) {
        set_error("tc2li_local_bundle_adjustment: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    const BaOptions opt = BaOptions::read();
    const std::string& inj_s = opt.shard_fail;
    // ---- sharded window: a failure is agreed on, no rank leaves alone -------------------------------------------------------
    // Every all-reduce carries one more element, the status word (0 = fine, 1 = this rank failed), and every rank reads it at
    // the synchronisation that follows: a peer's failure makes all ranks return TC2LI_ERR_COMM from the same collective.  A rank
    // that fails between two collectives (a HIP error, VisualProblem::setup, BalmTerm::build, an allocation) does not return at
    // once: it first joins the collective the others are heading for (`next`: same buffer size and operation -- the control
    // flow is replicated, so the failing rank knows it) with its status word set, then returns its own error.  What this cannot
    // cover: a failing all-reduce callback (the communicator itself is broken) and a device so broken that the poisoned
    // collective cannot be enqueued -- the launcher's watchdog has to end those.
    const bool sharded = shard != nullptr;
    struct Collective { double* dev = nullptr; size_t count = 0; int op = TC2LI_REDUCE_SUM; } next;
    // *shard_exit, for a sharded window that returns an error: kShardResultSumNext -- the peers' next collective is the wrapper's result sum
    // (it joins that sum with its status word set); kShardPeersTold -- the peers were told through the collective they were heading for;
    // kShardUnknown -- this rank cannot know what the peers enter next (or could not tell them): it must not enter any collective, the
    // launcher's watchdog ends the job
    enum { kShardResultSumNext = 0, kShardPeersTold = 1, kShardUnknown = 2 };
    bool result_sum_next = false;
    if (shard_exit) *shard_exit = kShardUnknown;
    BaWorkspace& ws = ba_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    double* h_stat = nullptr;  // [0..4): status words read back since the last synchronisation; [6] = 0.0, [7] = 1.0 (sources)
    int n_stat = 0;
    double* d_red = nullptr;
    // One collective of the replicated control flow.  Its own failures leave this rank out of step with the peers (the status word did not go
    // out, or the sum ran and its word could not be read: the peers are already past it): *shard_exit stays kShardUnknown.
    auto reduce = [&](double* dev, size_t count, int op, bool failed = false) -> int {
        next = Collective{};
        result_sum_next = false;
        if (hipMemcpyAsync(dev + count, h_stat + (failed ? 7 : 6), sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("sharded bundle adjustment: the status word could not be written");
            return (int)TC2LI_ERR_HIP;
        }
        const int rc = shard->allreduce(shard->ctx, dev, count + 1, op, st);
        if (rc != 0) { set_error("sharded bundle adjustment: the all-reduce callback returned %d", rc); return (int)TC2LI_ERR_COMM; }
        if (!failed && hipMemcpyAsync(h_stat + n_stat++, dev + count, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) {
            (void)hipGetLastError();
            set_error("sharded bundle adjustment: the status word could not be read");
            return (int)TC2LI_ERR_HIP;
        }
        return 0;
    };
    auto peer_failed = [&] {  // after a synchronisation
        bool bad = false;
        for (int i = 0; i < n_stat; ++i) bad |= h_stat[i] > 0;
        n_stat = 0;
        if (bad) set_error("sharded bundle adjustment: another rank failed; all ranks leave the window");
        return bad;
    };
    auto leave = [&](int rc) {  // a local failure: tell the peers through their next collective, then return the error
        if (sharded && next.dev && h_stat) {
            const std::string why = tc2li_last_error();
            const Collective c = next;
            if (reduce(c.dev, c.count, c.op, true) == 0 && hipStreamSynchronize(st) == hipSuccess) { if (shard_exit) *shard_exit = kShardPeersTold; }
            (void)hipGetLastError();
            set_error("%s", why.c_str());
        } else if (sharded && result_sum_next) {
            if (shard_exit) *shard_exit = kShardResultSumNext;
        }
        return rc;
    };
#define TC2LI_SH_CHECK(call)                                                                                         \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess) {                                                                                      \
            ::tc2li::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);             \
            return leave(TC2LI_ERR_HIP);                                                                             \
        }                                                                                                            \
    } while (0)
    if (sharded) {
        // the reduction buffer before anything that can fail: sized for the largest reduced system this window can have
        const size_t np_max = 6 * (size_t)n_poses;
        TC2LI_HIP_CHECK(ws.d_red.ensure(np_max * np_max + 2 * np_max + 27 * (size_t)n_poses + 16));
        TC2LI_HIP_CHECK(ws.h_stat.ensure(8));
        d_red = ws.d_red.p;
        h_stat = ws.h_stat.p;
        h_stat[6] = 0.0; h_stat[7] = 1.0;
        if (iterations > 0) next = Collective{d_red, 1, TC2LI_REDUCE_SUM};
        else result_sum_next = true;  // the wrapper's result sum comes first
        if (const char* inj = inj_s.empty() ? nullptr : inj_s.c_str()) {  // tests: "<rank>:setup" makes that rank fail before its first collective
            int r = -1; char where[16] = {0};
            if (sscanf(inj, "%d:%15s", &r, where) == 2 && r == shard->rank && !strcmp(where, "setup")) {
                set_error("injected failure (TC2LI_TEST_SHARD_FAIL=%s)", inj);
                return leave(TC2LI_ERR_INVALID);
            }
        }
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (lidar_stats) memset(lidar_stats, 0, sizeof(*lidar_stats));
    std::vector<uint8_t> extra_used;
    if (lidar_window) {
        if (lidar_window->n_keyframes < 1 || !lidar_window->pose_index) { set_error("lidar window: invalid argument"); return TC2LI_ERR_INVALID; }
        extra_used.assign(n_poses, 0);
        if (used_all) extra_used.assign(used_all, used_all + n_poses);
        for (int i = 0; i < lidar_window->n_keyframes; ++i) {
            const int k = lidar_window->pose_index[i];
            if (k < 0 || k >= n_poses) { set_error("lidar window: pose_index[%d] = %d out of range", i, k); return TC2LI_ERR_INVALID; }
            extra_used[k] = 1;
        }
    } else if (used_all) {
        extra_used.assign(used_all, used_all + n_poses);
    }
    BalmTerm* lidar = nullptr;
    if (lidar_window) {
        const int rc = ws.lidar.build(poses7, n_poses, lidar_window, st);
        if (rc < 0) return leave(rc);
        lidar = &ws.lidar;
    }
    VisualProblem vp;
    {
        const int rc = vp.setup(ws, poses7, fixed, n_poses, points3, n_points, edges, n_edges, cam, extra_used.empty() ? nullptr : extra_used.data(), st);
        if (rc < 0) return leave(rc);
    }
    BaProblemDev& pb = vp.pb;
    const std::vector<int>& pose_var = vp.pose_var;
    std::vector<Se3>& poses = vp.poses;
    const int n_free = vp.n_free, np = vp.np, n_slices = vp.n_slices, k_per_slice = vp.k_per_slice;
    auto &h_S = ws.h_S, &h_bs = ws.h_bs, &h_xp = ws.h_xp, &h_scal = ws.h_scal;
    auto &d_Hpp = ws.d_Hpp, &d_chi2 = ws.d_chi2;
    auto& d_depth = ws.d_depth;
    const size_t E = n_edges, P = n_points;

    const bool kTiming = opt.timing;
    double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // sharded window: the sums land in d_red, are all-reduced there and then copied to where the single-GPU path has them;
    // the stop flag is agreed on with the [scale, chi2] sum of every trial, so that all ranks leave the loops together
    bool stop_agreed = false;
    const Collective kTrialSystem{d_red, (size_t)np * np + 2 * (size_t)np, TC2LI_REDUCE_SUM}, kTrialScalars{d_red, 3, TC2LI_REDUCE_SUM};
    auto stopped = [&] { return sharded ? stop_agreed : (stop_flag && *stop_flag); };
    double lambda = -1, ni = 2;
    int n_bad = 0, done = 0, trials_total = 0;
    bool ok = true;
    std::vector<double> Swork((size_t)std::max(np * np, 1)), x(std::max(np, 1));
    std::vector<double> Hl, bl_;  // dense pose-pose contribution of the LiDAR edge
    if (lidar) { Hl.assign((size_t)np * np, 0.0); bl_.assign(np, 0.0); }
    tm[0] = now() - t_begin;
    for (int it = 0; it < iterations && !stopped() && ok; ++it) {
        double t0 = now();
        const bool want_maxdiag = it == 0 && !(lambda_init > 0);
        if (sharded) next = Collective{d_red, 1, TC2LI_REDUCE_SUM};
        // sharded layout of d_red in this phase: [0] chi2, [1] its status word, [4] largest landmark diagonal ([5]: status word, over
        // the pose maximum k_ba_maxdiag leaves there -- the pose diagonal is summed over the ranks below), [8 ..) packed Hpp
        ba_launch_linearize(pb, sharded ? d_red : h_scal.p, sharded ? d_red + 4 : h_scal.p + 1, want_maxdiag, st);
        TC2LI_SH_CHECK(hipGetLastError());
        const bool need_diag = (lidar || sharded) && want_maxdiag && n_free > 0;
        if (need_diag) TC2LI_SH_CHECK(ws.h_Hpp.ensure(27 * (size_t)n_free));
        if (sharded) {
            if (int rc = reduce(d_red, 1, TC2LI_REDUCE_SUM)) return rc;  // chi2 of all edges
            next = np > 0 ? kTrialSystem : kTrialScalars;
            // a failed rank joins ONE collective: before another one follows without a decision in between (first iteration only),
            // the status word of the one just made is read
            auto agreed = [&]() -> int {
                if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); set_error("sharded bundle adjustment: synchronisation failed"); return (int)TC2LI_ERR_HIP; }
                return peer_failed() ? (int)TC2LI_ERR_COMM : 0;
            };
            if (want_maxdiag) {
                next = Collective{d_red + 4, 1, TC2LI_REDUCE_MAX};
                if (int rc = agreed()) return rc;
                if (int rc = reduce(d_red + 4, 1, TC2LI_REDUCE_MAX)) return rc;  // largest landmark diagonal of any rank
                next = np > 0 ? kTrialSystem : kTrialScalars;
                if (need_diag) {  // the pose diagonal is a sum over the ranks' edges before it is a maximum
                    double* d_hpp_sum = d_red + 8;
                    next = Collective{d_hpp_sum, 27 * (size_t)n_free, TC2LI_REDUCE_SUM};
                    if (int rc = agreed()) return rc;
                    TC2LI_SH_CHECK(hipMemcpyAsync(d_hpp_sum, d_Hpp.p, 27 * (size_t)n_free * sizeof(double), hipMemcpyDeviceToDevice, st));
                    if (int rc = reduce(d_hpp_sum, 27 * (size_t)n_free, TC2LI_REDUCE_SUM)) return rc;
                    next = np > 0 ? kTrialSystem : kTrialScalars;
                    TC2LI_SH_CHECK(hipMemcpyAsync(ws.h_Hpp.p, d_hpp_sum, 27 * (size_t)n_free * sizeof(double), hipMemcpyDeviceToHost, st));
                }
            }
            TC2LI_SH_CHECK(hipMemcpyAsync(h_scal.p, d_red, sizeof(double), hipMemcpyDeviceToHost, st));
            TC2LI_SH_CHECK(hipMemcpyAsync(h_scal.p + 1, d_red + 4, sizeof(double), hipMemcpyDeviceToHost, st));
            h_scal.p[2] = 0;  // the pose maximum comes from the summed Hpp (need_diag) or there is no free pose
        } else if (need_diag) {
            TC2LI_HIP_CHECK(hipMemcpyAsync(ws.h_Hpp.p, d_Hpp.p, 27 * (size_t)n_free * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        if (lidar) {  // computeActiveErrors + linearizeOplus of the LiDAR edge ride on the same synchronisation
            // after the first iteration the accepted estimate IS the last trial (a rejected last trial ends the loop): its residual
            // and plane decompositions are already there, computing them again would give the same bits
            if (it == 0) lidar->enqueue_error(pb.poses, st);
            else lidar->eig_at = pb.poses;
            const int rc = lidar->enqueue_linearization(pb.poses, st);
            if (rc < 0) return leave(rc);
        }
        TC2LI_SH_CHECK(hipGetLastError());
        TC2LI_SH_CHECK(hipStreamSynchronize(st));
        if (sharded && peer_failed()) return (int)TC2LI_ERR_COMM;
        tm[1] += now() - t0; t0 = now();
        double currentChi = h_scal.p[0];
        double max_pose_diag = h_scal.p[2];
        if (lidar) {
            lidar->finish_error();
            currentChi = lidar->chi2() + currentChi;
            lidar->finish_linearization();  // constructQuadraticForm uses the stored Jacobian / Hessian when the cost grew
            std::fill(Hl.begin(), Hl.end(), 0.0);
            std::fill(bl_.begin(), bl_.end(), 0.0);
            lidar->add_quadratic_form(pose_var.data(), np, Hl.data(), bl_.data());
        }
        if (need_diag) {
            static const int dpos[6] = {0, 6, 11, 15, 18, 20};  // diagonal of the packed upper triangle
            max_pose_diag = 0;
            for (int j = 0; j < np; ++j)
                max_pose_diag = std::max(max_pose_diag, std::fabs(ws.h_Hpp.p[27 * (size_t)(j / 6) + dpos[j % 6]] + (lidar ? Hl[(size_t)j * np + j] : 0.0)));
        }
        tm[2] += now() - t0;
        double tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) {
            if (stats) stats->initial_chi2 = currentChi;
            lambda = lambda_init > 0 ? lambda_init : 1e-5 * std::max(h_scal.p[1], max_pose_diag);
            ni = 2;
            n_bad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            bool ok2 = true;
            t0 = now();
            // a rank's part of the reduced camera system: its edges' Hpp and b_p, minus its landmarks' W Hll^-1 W^T and
            // W Hll^-1 b_l; lambda goes onto the diagonal once (rank 0)
            if (sharded) TC2LI_SH_CHECK(hipMemsetAsync(d_red, 0, (size_t)np * np * sizeof(double), st));  // above the diagonal: summed, never read
            ba_launch_schur(pb, lambda, sharded && shard->rank != 0 ? 0.0 : lambda, n_slices, k_per_slice, sharded ? d_red : h_S.p,
                            sharded ? d_red + (size_t)np * np : h_bs.p, st);
            TC2LI_SH_CHECK(hipGetLastError());
            if (np > 0) {
                if (sharded) {
                    if (int rc = reduce(d_red, (size_t)np * np + 2 * (size_t)np, TC2LI_REDUCE_SUM)) return rc;
                    next = kTrialScalars;  // what follows when the reduced system can be solved (a rank that fails before it knows cannot tell)
                    TC2LI_SH_CHECK(hipMemcpyAsync(h_S.p, d_red, (size_t)np * np * sizeof(double), hipMemcpyDeviceToHost, st));
                    TC2LI_SH_CHECK(hipMemcpyAsync(h_bs.p, d_red + (size_t)np * np, 2 * (size_t)np * sizeof(double), hipMemcpyDeviceToHost, st));
                }
                TC2LI_SH_CHECK(hipStreamSynchronize(st));
                if (sharded && peer_failed()) return (int)TC2LI_ERR_COMM;
                tm[3] += now() - t0; t0 = now();
                memcpy(Swork.data(), h_S.p, (size_t)np * np * sizeof(double));
                if (lidar) {
                    for (size_t k = 0; k < (size_t)np * np; ++k) Swork[k] += Hl[k];
                    for (int j = 0; j < np; ++j) { h_bs.p[j] += bl_[j]; h_bs.p[np + j] += bl_[j]; }
                }
                ok2 = ldlt_solve_small(Swork.data(), np, h_bs.p, x.data(), false);
                memcpy(h_xp.p, x.data(), np * sizeof(double));
                tm[4] += now() - t0; t0 = now();
            }
            double scale = 0;
            // pose part of computeScale(): b_p is what the finish kernel left in h_bs[np .. 2 np)
            for (int j = 0; j < np; ++j) scale += x[j] * (lambda * x[j] + h_bs.p[np + j]);
            if (ok2) {
                ba_launch_trial(pb, h_xp.p, lambda, sharded ? d_red : h_scal.p + 3, sharded ? d_red + 1 : h_scal.p + 4, st);
                if (sharded) {
                    next = kTrialScalars;
                    h_scal.p[7] = stop_flag && *stop_flag ? 1.0 : 0.0;
                    TC2LI_SH_CHECK(hipMemcpyAsync(d_red + 2, h_scal.p + 7, sizeof(double), hipMemcpyHostToDevice, st));
                    if (int rc = reduce(d_red, 3, TC2LI_REDUCE_SUM)) return rc;
                    next = Collective{};  // trial again, next iteration or the result sum: decided by values this rank has not read yet
                    TC2LI_SH_CHECK(hipMemcpyAsync(h_scal.p + 3, d_red, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
                }
                if (lidar) lidar->enqueue_error(pb.poses_trial, st);
                TC2LI_SH_CHECK(hipGetLastError());
                TC2LI_SH_CHECK(hipStreamSynchronize(st));
                if (sharded && peer_failed()) return (int)TC2LI_ERR_COMM;
                if (sharded)
                    if (const char* inj = inj_s.empty() ? nullptr : inj_s.c_str()) {  // tests: "<rank>:trial" fails here, where the next collective is not known yet
                        int r = -1; char where[16] = {0};
                        if (sscanf(inj, "%d:%15s", &r, where) == 2 && r == shard->rank && !strcmp(where, "trial")) {
                            set_error("injected failure (TC2LI_TEST_SHARD_FAIL=%s)", inj);
                            return leave(TC2LI_ERR_INVALID);
                        }
                    }
                if (sharded) stop_agreed = h_scal.p[5] > 0;
                tempChi = h_scal.p[4];
                scale += h_scal.p[3];
                tm[5] += now() - t0; t0 = now();
                if (lidar) {
                    lidar->finish_error();
                    tempChi = lidar->chi2() + tempChi;
                    tm[6] += now() - t0;
                }
            } else {
                tempChi = std::numeric_limits<double>::max();
            }
            if (sharded) next = np > 0 ? kTrialSystem : kTrialScalars;  // another trial (corrected below when the loops end)
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                lambda = lm_lambda_accepted(lambda, rho);
                ni = 2;
                currentChi = tempChi;
                std::swap(pb.poses, pb.poses_trial);
                std::swap(pb.points, pb.points_trial);
            } else {
                lambda *= ni;
                ni *= 2;
            }
            qmax++;
            trials_total++;
        } while (rho < 0 && qmax < 10 && !stopped());
        ++done;
        if (stats) { stats->final_chi2 = currentChi; stats->final_lambda = lambda; }
        if (qmax == 10 || rho == 0) { ok = false; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) ok = false;
    }
    next = Collective{};  // the wrapper's result sum is the next collective: it tells the peers itself
    result_sum_next = true;
    if (sharded && shard_exit) *shard_exit = kShardResultSumNext;  // also for the plain HIP checks of the result copies below
    if (stats) { stats->iterations = done; stats->trials = trials_total; stats->n_free_poses = n_free; }
    if (lidar && lidar_stats) {
        lidar_stats->n_planes = lidar->n_planes; lidar_stats->hessian_evaluations = lidar->hessian_evaluations;
        lidar_stats->residual = lidar->error; lidar_stats->chi2 = lidar->chi2();
    }
    // ---- results ----
    ba_launch_depth(pb, d_depth.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(poses.data(), pb.poses, n_poses * sizeof(Se3), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(points3, pb.points, 3 * P * sizeof(double), hipMemcpyDeviceToHost, st));
    if (edge_chi2) TC2LI_HIP_CHECK(hipMemcpyAsync(edge_chi2, d_chi2.p, E * sizeof(double), hipMemcpyDeviceToHost, st));
    if (edge_depth_positive) TC2LI_HIP_CHECK(hipMemcpyAsync(edge_depth_positive, d_depth.p, E, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < n_poses; ++k) { memcpy(poses7 + 7 * k, poses[k].q, 4 * sizeof(double)); memcpy(poses7 + 7 * k + 4, poses[k].t, 3 * sizeof(double)); }
    if (kTiming) fprintf(stderr, "BA timing ms: setup %.3f linearize %.3f lidar-lin %.3f schur %.3f solve %.3f trial %.3f lidar-err %.3f total %.3f\n", tm[0], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], now() - t_begin);
    return done;
#undef TC2LI_SH_CHECK
}

int tc2li_local_lv_bundle_adjustment(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                     const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                     double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                     uint8_t* edge_depth_positive, tc2li_ba_stats* stats, const tc2li_lidar_window* lidar_window,
                                     tc2li_lidar_ba_stats* lidar_stats, void* stream) {
    return lv_ba_impl(poses7, fixed, n_poses, points3, n_points, edges, n_edges, cam, iterations, lambda_init, stop_flag, edge_chi2,
                      edge_depth_positive, stats, lidar_window, lidar_stats, nullptr, nullptr, stream);
}

int tc2li_ba_shard_select(const tc2li_ba_edge* edges, int n_edges, int n_points, int rank, int world, uint8_t* landmark_owned,
                          uint8_t* edge_owned) {
    if (n_edges < 0 || n_points < 0 || world < 1 || rank < 0 || rank >= world || (n_edges > 0 && !edges)) {
        set_error("tc2li_ba_shard_select: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (landmark_owned) for (int l = 0; l < n_points; ++l) landmark_owned[l] = l % world == rank;
    int owned = 0;
    for (int e = 0; e < n_edges; ++e) {
        if (edges[e].point < 0 || edges[e].point >= n_points) { set_error("edge %d references point %d out of range", e, edges[e].point); return TC2LI_ERR_INVALID; }
        const bool mine = edges[e].point % world == rank;
        if (edge_owned) edge_owned[e] = mine;
        owned += mine;
    }
    return owned;
}

int tc2li_local_lv_bundle_adjustment_sharded(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                             const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                             double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                             uint8_t* edge_depth_positive, tc2li_ba_stats* stats, const tc2li_lidar_window* lidar_window,
                                             tc2li_lidar_ba_stats* lidar_stats, const tc2li_ba_shard* shard, void* stream_) {
    if (!shard || !shard->allreduce || shard->world < 1 || shard->rank < 0 || shard->rank >= shard->world) {
        set_error("tc2li_local_lv_bundle_adjustment_sharded: invalid shard description");
        return TC2LI_ERR_INVALID;
    }
    if (!poses7 || !fixed || !points3 || !edges || !cam || n_poses <= 0 || n_points <= 0 || n_edges <= 0 || iterations < 0) {
        set_error("tc2li_local_lv_bundle_adjustment_sharded: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    // Everything that can be rejected is rejected here, on the whole window, which every rank sees: a rank that left on its own
    // would leave the others waiting in the collective.
    if (n_points < shard->world) { set_error("a window of %d points cannot be split over %d ranks", n_points, shard->world); return TC2LI_ERR_INVALID; }
    std::vector<uint8_t> used_all(n_poses, 0), has_edge(n_points, 0);
    for (int e = 0; e < n_edges; ++e) {
        if (edges[e].pose < 0 || edges[e].pose >= n_poses || edges[e].point < 0 || edges[e].point >= n_points) {
            set_error("edge %d references pose %d / point %d out of range", e, edges[e].pose, edges[e].point);
            return TC2LI_ERR_INVALID;
        }
        used_all[edges[e].pose] = 1;
        has_edge[edges[e].point] = 1;
    }
    for (int l = 0; l < n_points; ++l) if (!has_edge[l]) { set_error("point %d has no edge", l); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    const int rank = shard->rank, world = shard->world;
    // ---- this rank's landmarks and their edges, renumbered ----
    std::vector<int> local_of(n_points, -1), global_point, global_edge;
    for (int l = rank; l < n_points; l += world) { local_of[l] = (int)global_point.size(); global_point.push_back(l); }
    std::vector<double> pts(3 * global_point.size());
    for (size_t i = 0; i < global_point.size(); ++i) memcpy(&pts[3 * i], points3 + 3 * (size_t)global_point[i], 3 * sizeof(double));
    std::vector<tc2li_ba_edge> mine;
    for (int e = 0; e < n_edges; ++e) {
        if (local_of[edges[e].point] < 0) continue;
        mine.push_back(edges[e]);
        mine.back().point = local_of[edges[e].point];
        global_edge.push_back(e);
    }
    std::vector<double> chi2(mine.size());
    std::vector<uint8_t> depth(mine.size());
    int shard_exit = 2;
    int rc = lv_ba_impl(poses7, fixed, n_poses, pts.data(), (int)global_point.size(), mine.data(), (int)mine.size(), cam, iterations,
                        lambda_init, stop_flag, chi2.data(), depth.data(), stats, lidar_window, lidar_stats, shard, used_all.data(), stream_, &shard_exit);
    // a failed rank joins the result sum only when that sum IS the collective the peers enter next; told inside the loops: every rank is
    // leaving; otherwise (the peers are in a collective of another size, or nothing is known) no collective is entered at all -- a sum of the
    // wrong size would hang or corrupt the others, the launcher's watchdog ends the job instead
    if (rc == TC2LI_ERR_COMM || (rc < 0 && shard_exit != 0)) return rc;
    // ---- every rank receives the whole result: one sum of [points | chi2 | depth flags | status word], zeros where another rank owns the
    // entry.  A rank that failed locally after its last collective joins this sum with its status word set (and zeros), so that the others
    // return TC2LI_ERR_COMM instead of waiting for it. ----
    const size_t P = n_points, E = n_edges, total = 3 * P + 2 * E;
    std::vector<double> all(total + 1, 0.0);
    BaWorkspace& ws = ba_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    if (rc >= 0 && ws.d_red.ensure(total + 1) != hipSuccess) { set_error("sharded bundle adjustment: no memory for the result sum"); rc = TC2LI_ERR_HIP; }
    if (rc >= 0) {
        for (size_t i = 0; i < global_point.size(); ++i) memcpy(&all[3 * (size_t)global_point[i]], &pts[3 * i], 3 * sizeof(double));
        for (size_t i = 0; i < global_edge.size(); ++i) { all[3 * P + global_edge[i]] = chi2[i]; all[3 * P + E + global_edge[i]] = depth[i]; }
    } else {
        all[total] = 1.0;
        if (ws.d_red.n < total + 1) return rc;  // nothing to send from: the peers cannot be told
    }
    const std::string why = rc < 0 ? tc2li_last_error() : "";
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_red.p, all.data(), (total + 1) * sizeof(double), hipMemcpyHostToDevice, st));
    if (shard->allreduce(shard->ctx, ws.d_red.p, total + 1, TC2LI_REDUCE_SUM, st) != 0) { set_error("sharded bundle adjustment: the final all-reduce failed"); return TC2LI_ERR_COMM; }
    TC2LI_HIP_CHECK(hipMemcpyAsync(all.data(), ws.d_red.p, (total + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    if (rc < 0) { set_error("%s", why.c_str()); return rc; }
    if (all[total] > 0) { set_error("sharded bundle adjustment: another rank failed; all ranks leave the window"); return TC2LI_ERR_COMM; }
    memcpy(points3, all.data(), 3 * P * sizeof(double));
    if (edge_chi2) memcpy(edge_chi2, &all[3 * P], E * sizeof(double));
    if (edge_depth_positive) for (size_t e = 0; e < E; ++e) edge_depth_positive[e] = all[3 * P + E + e] != 0;
    return rc;
}

int tc2li_local_bundle_adjustment(double* poses7, const uint8_t* fixed, int n_poses, double* points3, int n_points,
                                  const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, int iterations,
                                  double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                  uint8_t* edge_depth_positive, tc2li_ba_stats* stats, void* stream) {
    return tc2li_local_lv_bundle_adjustment(poses7, fixed, n_poses, points3, n_points, edges, n_edges, cam, iterations, lambda_init,
                                            stop_flag, edge_chi2, edge_depth_positive, stats, nullptr, nullptr, stream);
}


int tc2li_ba_options(char* text, int capacity) {
    const BaOptions o = BaOptions::read();
    char buf[256];
    const int n = snprintf(buf, sizeof(buf), "{\"device_lm\": %d, \"device_solve\": %d, \"fuse_linearize\": %d, \"fuse_trial\": %d, \"lvi_device_solve\": %d, "
                           "\"lockstep\": %d, \"groups\": %d}", (int)o.device_lm, (int)o.device_solve, (int)o.fuse_linearize, (int)o.fuse_trial, (int)o.lvi_device_solve,
                           (int)o.lockstep, o.groups);
    if (text && capacity > 0) { const int m = std::min(capacity - 1, n); memcpy(text, buf, m); text[m] = 0; }
    return n + 1;
}

int tc2li_lidar_window_evaluate(const double* poses7, int n_poses, const tc2li_lidar_window* win, double* residual, double* JacT,
                                double* Hessian, void* stream_) {
    if (!poses7 || n_poses <= 0 || !win) { set_error("tc2li_lidar_window_evaluate: invalid argument"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    BaWorkspace& ws = ba_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    int rc = ws.lidar.build(poses7, n_poses, win, st);
    if (rc < 0) return rc;
    std::vector<Se3> poses(n_poses);
    for (int k = 0; k < n_poses; ++k) { memcpy(poses[k].q, poses7 + 7 * k, 4 * sizeof(double)); memcpy(poses[k].t, poses7 + 7 * k + 4, 3 * sizeof(double)); }
    TC2LI_HIP_CHECK(ws.d_poses.ensure(n_poses));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_poses.p, poses.data(), n_poses * sizeof(Se3), hipMemcpyHostToDevice, st));
    rc = ws.lidar.compute_error(ws.d_poses.p, st);
    if (rc < 0) return rc;
    if (residual) *residual = ws.lidar.error;
    if (JacT && Hessian) {
        ws.lidar.is_calc_hess = true;
        rc = ws.lidar.linearize(ws.d_poses.p, st);
        if (rc < 0) return rc;
        const int n = 6 * ws.lidar.W;
        memcpy(JacT, ws.lidar.JacT.data(), n * sizeof(double));
        memcpy(Hessian, ws.lidar.Hessian.data(), (size_t)n * n * sizeof(double));
    }
    return ws.lidar.n_planes;
}

}  // extern "C"
