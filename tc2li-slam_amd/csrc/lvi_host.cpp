// LocalInertialBA / LocalLVIBA: the one-window entry points, the lock-step batch of such windows and the reduced-system solvers' entry points (include/tc2li_hip.h).
#include "ba_internal.hpp"

using namespace tc2li;
using namespace tc2li::ba_detail;

extern "C" {

static_assert(sizeof(tc2li_imu_calib) == sizeof(ImuCalib), "ABI layout");

int tc2li_local_inertial_bundle_adjustment(tc2li_inertial_keyframe* kfs, const uint8_t* fixed, const uint8_t* has_imu, int n_kfs,
                                           const tc2li_imu_calib* calib, double* points3, int n_points, const tc2li_ba_edge* edges,
                                           int n_edges, const tc2li_inertial_link* links, int n_links, const tc2li_camera* cam,
                                           int iterations, double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2,
                                           uint8_t* edge_depth_positive, tc2li_ba_stats* stats, void* stream_) {
    return tc2li_local_lvi_bundle_adjustment(kfs, fixed, has_imu, n_kfs, calib, points3, n_points, edges, n_edges, links, n_links, cam, iterations,
                                             lambda_init, stop_flag, edge_chi2, edge_depth_positive, stats, nullptr, nullptr, nullptr, stream_);
}

int tc2li_local_lvi_bundle_adjustment(tc2li_inertial_keyframe* kfs, const uint8_t* fixed, const uint8_t* has_imu, int n_kfs,
                                      const tc2li_imu_calib* calib, double* points3, int n_points, const tc2li_ba_edge* edges, int n_edges,
                                      const tc2li_inertial_link* links, int n_links, const tc2li_camera* cam, int iterations,
                                      double lambda_init, const volatile uint8_t* stop_flag, double* edge_chi2, uint8_t* edge_depth_positive,
                                      tc2li_ba_stats* stats, const tc2li_lidar_window* lidar_window, const float* Tbl7,
                                      tc2li_lidar_ba_stats* lidar_stats, void* stream_) {
    if (!kfs || !fixed || !has_imu || !calib || !points3 || !edges || !cam || n_kfs <= 0 || n_points <= 0 || n_edges <= 0 || n_links < 0 ||
        (n_links > 0 && !links) || iterations < 0 || (lidar_window && !Tbl7)) {
        set_error("tc2li_local_inertial_bundle_adjustment: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (lidar_stats) memset(lidar_stats, 0, sizeof(*lidar_stats));
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    if (stats) memset(stats, 0, sizeof(*stats));
    // ---- inertial edges ----
    InertialTerm inertial;
    std::vector<uint8_t> extra_used;
    {
        const int rc = inertial.prepare(links, n_links, has_imu, n_kfs, extra_used);
        if (rc < 0) return rc;
    }
    std::vector<uint8_t> imu_used = extra_used;  // keyframes whose velocity / bias vertices an inertial edge touches
    BaWorkspace& ws = ba_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    static_assert(offsetof(tc2li_inertial_keyframe, Rcw) == 0 && offsetof(tc2li_inertial_keyframe, tcw) == 72, "Rcw, tcw first");
    static_assert(offsetof(ImuPose, Rcw) == 0 && offsetof(ImuPose, tcw) == 72, "Rcw, tcw first");
    BalmTerm* lidar = nullptr;
    if (lidar_window) {
        const int rc = ws.lidar.build_body(kfs, sizeof(tc2li_inertial_keyframe), n_kfs, lidar_window, Tbl7, sizeof(ImuPose), st);
        if (rc < 0) return rc;
        lidar = &ws.lidar;
        for (int i = 0; i < lidar_window->n_keyframes; ++i) extra_used[lidar_window->pose_index[i]] = 1;
    }
    VisualProblem vp;
    {
        const int rc = vp.setup(ws, nullptr, fixed, n_kfs, points3, n_points, edges, n_edges, cam, extra_used.data(), st);
        if (rc < 0) return rc;
    }
    BaProblemDev& pb = vp.pb;
    const std::vector<int>& pose_var = vp.pose_var;
    const int n_free = vp.n_free, np = vp.np;
    inertial.number(fixed, has_imu, imu_used, n_kfs, pose_var, np);
    if (lidar_window)   // the LiDAR term's blocks of the reduced system (BalmTerm::add_quadratic_form)
        for (int i = 0; i < lidar_window->n_keyframes; ++i)
            for (int j = 0; j < lidar_window->n_keyframes; ++j) {
                const int vi = pose_var[lidar_window->pose_index[i]], vj = pose_var[lidar_window->pose_index[j]];
                if (vi >= 0 && vj >= 0) inertial.note_block(6 * vi, 6 * vj, 6, 6);
            }
    const std::vector<int>& imu_var = inertial.imu_var;
    const int n = inertial.n;
    // ---- keyframe states: ImuCamPose on the device (authoritative), a host mirror for the inertial edges ----
    std::vector<ImuPose> hp(n_kfs), hp_trial(n_kfs);
    std::vector<ImuVertexState> sv(n_kfs), sv_trial(n_kfs);
    for (int k = 0; k < n_kfs; ++k) {
        memcpy(hp[k].Rcw, kfs[k].Rcw, 72); memcpy(hp[k].tcw, kfs[k].tcw, 24); memcpy(hp[k].Rwb, kfs[k].Rwb, 72); memcpy(hp[k].twb, kfs[k].twb, 24);
        hp[k].its = 0; hp[k].pad_ = 0;
        memcpy(sv[k].v, kfs[k].velocity, 24); memcpy(sv[k].bg, kfs[k].gyro_bias, 24); memcpy(sv[k].ba, kfs[k].acc_bias, 24);
    }
    TC2LI_HIP_CHECK(ws.d_iposes.ensure(n_kfs)); TC2LI_HIP_CHECK(ws.d_iposes_trial.ensure(n_kfs)); TC2LI_HIP_CHECK(ws.h_iposes.ensure(n_kfs));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_iposes.p, hp.data(), n_kfs * sizeof(ImuPose), hipMemcpyHostToDevice, st));
    pb.inertial = 1; pb.iposes = ws.d_iposes.p; pb.iposes_trial = ws.d_iposes_trial.p;
    vp.decide_trial_fused();
    memcpy(&pb.calib, calib, sizeof(ImuCalib));
    auto &h_S = ws.h_S, &h_bs = ws.h_bs, &h_xp = ws.h_xp, &h_scal = ws.h_scal;
    const size_t E = n_edges, P = n_points;

    std::vector<double>&Hi = inertial.Hi, &bi = inertial.bi;
    auto inertial_cost = [&](const std::vector<ImuPose>& Pz, const std::vector<ImuVertexState>& Sz, bool linearize) { return inertial.cost(Pz, Sz, linearize); };

    auto stopped = [&] { return stop_flag && *stop_flag; };
    double lambda = lambda_init, ni = 2, last_chi = 0;
    int n_bad = 0, done = 0, trials_total = 0;
    bool ok = true;
    std::vector<double> rhs(std::max(n, 1)), bfull(std::max(n, 1)), x(std::max(n, 1), 0.0);
    ReducedSolver solver;
    // the reduced system on the device (k_lvi_solve: the kernel body of the lock-step batch, so a window gives the same bits here and there)
    const bool dev_solve = inertial.device_solve_ok();
    if (dev_solve) {
        TC2LI_HIP_CHECK(ws.lvi.ensure(np, n - np)); TC2LI_HIP_CHECK(ws.d_S.ensure((size_t)np * np)); TC2LI_HIP_CHECK(ws.d_bs.ensure(2 * (size_t)np));
        TC2LI_HIP_CHECK(ws.d_xp.ensure(n)); TC2LI_HIP_CHECK(ws.h_xp.ensure(n)); TC2LI_HIP_CHECK(ws.h_ok.ensure(1));
    }
    for (int it = 0; it < iterations && !stopped() && ok; ++it) {
        ba_launch_linearize(pb, h_scal.p, h_scal.p + 1, it == 0 && !(lambda_init > 0), st);
        TC2LI_HIP_CHECK(hipGetLastError());
        if (lidar) {  // computeActiveErrors + linearizeOplus of the LiDAR edge ride on the same synchronisation
            lidar->enqueue_error(reinterpret_cast<const Se3*>(pb.iposes), st);
            const int rc = lidar->enqueue_linearization(reinterpret_cast<const Se3*>(pb.iposes), st);
            if (rc < 0) return rc;
        }
        double chi_imu = inertial_cost(hp, sv, true);  // overlaps with the kernels
        TC2LI_HIP_CHECK(hipStreamSynchronize(st));
        if (lidar) {
            if (it == 0) lidar->finish_error();  // the computeActiveErrors() before optimize() (OptimizerWithLidar.cc:978)
            lidar->finish_error();
            chi_imu += lidar->chi2();
            lidar->finish_linearization();  // constructQuadraticForm uses the stored Jacobian / Hessian when the cost grew
            lidar->add_quadratic_form(pose_var.data(), n, Hi.data(), bi.data());
        }
        solver.set_pattern(Hi.data(), n, np, !dev_solve);
        if (dev_solve) {
            if (solver.band() > kLviBand) { set_error("tc2li_local_lvi_bundle_adjustment: inertial band wider than the device solve holds"); return TC2LI_ERR_INVALID; }
            const size_t bytes = ws.lvi.pack(solver, Hi.data(), bi.data());
            TC2LI_HIP_CHECK(hipMemcpyAsync(ws.lvi.d_blob.p, ws.lvi.h_blob.p, bytes, hipMemcpyHostToDevice, st));
        }
        double currentChi = chi_imu + h_scal.p[0], tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) {
            if (stats) stats->initial_chi2 = currentChi;
            last_chi = currentChi;
            if (!(lambda_init > 0)) {  // computeLambdaInit over the whole diagonal (not used by the reference's settings)
                double mx = std::max(h_scal.p[1], h_scal.p[2]);
                for (int j = np; j < n; ++j) mx = std::max(mx, std::fabs(Hi[(size_t)j * n + j]));
                lambda = 1e-5 * mx;
            }
            ni = 2;
            n_bad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            if (dev_solve) {  // Schur product, solve, trial estimate and its cost in one queue, one synchronisation
                ba_launch_schur(pb, lambda, lambda, vp.n_slices, vp.k_per_slice, ws.d_S.p, ws.d_bs.p, st);
                lvi_launch_solve(ws.lvi.dev, ws.d_S.p, ws.d_bs.p, lambda, ws.d_xp.p, h_xp.p, ws.h_ok.p, st);
                TC2LI_HIP_CHECK(hipMemcpyAsync(h_bs.p + np, ws.d_bs.p + np, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, st));
                ba_launch_trial(pb, ws.d_xp.p, lambda, h_scal.p + 3, h_scal.p + 4, st);
                TC2LI_HIP_CHECK(hipGetLastError());
                TC2LI_HIP_CHECK(hipMemcpyAsync(ws.h_iposes.p, pb.iposes_trial, n_kfs * sizeof(ImuPose), hipMemcpyDeviceToHost, st));
                if (lidar) lidar->enqueue_error(reinterpret_cast<const Se3*>(pb.iposes_trial), st);
                TC2LI_HIP_CHECK(hipStreamSynchronize(st));
                const bool ok2 = ws.h_ok.p[0] != 0;
                memcpy(x.data(), h_xp.p, (size_t)n * sizeof(double));
                double scale = 0;
                for (int j = 0; j < n; ++j) {
                    const double bf = bi[j] + (j < np ? h_bs.p[np + j] : 0.0);
                    scale += x[j] * (lambda * x[j] + bf);
                }
                if (ok2) {
                    sv_trial = sv;
                    for (int k = 0; k < n_kfs; ++k)
                        if (imu_var[k] >= 0) {
                            const double* u = &x[np + 9 * imu_var[k]];
                            for (int c = 0; c < 3; ++c) { sv_trial[k].v[c] += u[c]; sv_trial[k].bg[c] += u[3 + c]; sv_trial[k].ba[c] += u[6 + c]; }
                        }
                    memcpy(hp_trial.data(), ws.h_iposes.p, n_kfs * sizeof(ImuPose));
                    tempChi = inertial_cost(hp_trial, sv_trial, false) + h_scal.p[4];
                    if (lidar) { lidar->finish_error(); tempChi += lidar->chi2(); }
                    scale += h_scal.p[3];
                    last_chi = tempChi;
                } else {
                    tempChi = std::numeric_limits<double>::max();
                }
                rho = currentChi - tempChi;
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    lambda = lm_lambda_accepted(lambda, rho);
                    ni = 2;
                    currentChi = tempChi;
                    std::swap(pb.iposes, pb.iposes_trial);
                    std::swap(pb.points, pb.points_trial);
                    hp.swap(hp_trial);
                    sv.swap(sv_trial);
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                qmax++;
                trials_total++;
                continue;
            }
            ba_launch_schur(pb, lambda, lambda, vp.n_slices, vp.k_per_slice, h_S.p, h_bs.p, st);
            TC2LI_HIP_CHECK(hipGetLastError());
            TC2LI_HIP_CHECK(hipStreamSynchronize(st));
            // reduced system: [S_visual + H_inertial(poses)   H_inertial(poses, imu) ; ...   H_inertial(imu) + lambda I]
            // (the envelope LDL^T of reduced_solve.hpp: velocity / bias unknowns first, the pose rows after them)
            for (int j = 0; j < n; ++j) {
                bfull[j] = bi[j] + (j < np ? h_bs.p[np + j] : 0.0);
                rhs[j] = bi[j] + (j < np ? h_bs.p[j] : 0.0);
            }
            const bool ok2 = n == 0 ? true : solver.factorise(Hi.data(), h_S.p, lambda);
            if (ok2 && n) solver.solve(rhs.data(), x.data());
            double scale = 0;
            for (int j = 0; j < n; ++j) scale += x[j] * (lambda * x[j] + bfull[j]);
            if (ok2) {
                if (np) memcpy(h_xp.p, x.data(), np * sizeof(double));
                ba_launch_trial(pb, h_xp.p, lambda, h_scal.p + 3, h_scal.p + 4, st);
                TC2LI_HIP_CHECK(hipGetLastError());
                TC2LI_HIP_CHECK(hipMemcpyAsync(ws.h_iposes.p, pb.iposes_trial, n_kfs * sizeof(ImuPose), hipMemcpyDeviceToHost, st));
                if (lidar) lidar->enqueue_error(reinterpret_cast<const Se3*>(pb.iposes_trial), st);
                sv_trial = sv;
                for (int k = 0; k < n_kfs; ++k)
                    if (imu_var[k] >= 0) {
                        const double* u = &x[np + 9 * imu_var[k]];
                        for (int c = 0; c < 3; ++c) { sv_trial[k].v[c] += u[c]; sv_trial[k].bg[c] += u[3 + c]; sv_trial[k].ba[c] += u[6 + c]; }
                    }
                TC2LI_HIP_CHECK(hipStreamSynchronize(st));
                memcpy(hp_trial.data(), ws.h_iposes.p, n_kfs * sizeof(ImuPose));
                tempChi = inertial_cost(hp_trial, sv_trial, false) + h_scal.p[4];
                if (lidar) { lidar->finish_error(); tempChi += lidar->chi2(); }
                scale += h_scal.p[3];
                last_chi = tempChi;
            } else {
                tempChi = std::numeric_limits<double>::max();
            }
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                lambda = lm_lambda_accepted(lambda, rho);
                ni = 2;
                currentChi = tempChi;
                std::swap(pb.iposes, pb.iposes_trial);
                std::swap(pb.points, pb.points_trial);
                hp.swap(hp_trial);
                sv.swap(sv_trial);
            } else {
                lambda *= ni;
                ni *= 2;
            }
            qmax++;
            trials_total++;
        } while (rho < 0 && qmax < 10 && !stopped());
        ++done;
        if (stats) stats->final_lambda = lambda;
        if (qmax == 10 || rho == 0) { ok = false; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) n_bad++; else n_bad = 0;
        if (n_bad >= 3) ok = false;
    }
    if (stats) { stats->iterations = done; stats->trials = trials_total; stats->n_free_poses = n_free; stats->final_chi2 = last_chi; }
    if (lidar && lidar_stats) {
        lidar_stats->n_planes = lidar->n_planes; lidar_stats->hessian_evaluations = lidar->hessian_evaluations;
        lidar_stats->residual = lidar->error; lidar_stats->chi2 = lidar->chi2();
    }
    // ---- results ----
    ba_launch_depth(pb, ws.d_depth.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(points3, pb.points, 3 * P * sizeof(double), hipMemcpyDeviceToHost, st));
    if (edge_chi2) TC2LI_HIP_CHECK(hipMemcpyAsync(edge_chi2, ws.d_chi2.p, E * sizeof(double), hipMemcpyDeviceToHost, st));
    if (edge_depth_positive) TC2LI_HIP_CHECK(hipMemcpyAsync(edge_depth_positive, ws.d_depth.p, E, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < n_kfs; ++k) {
        memcpy(kfs[k].Rcw, hp[k].Rcw, 72); memcpy(kfs[k].tcw, hp[k].tcw, 24); memcpy(kfs[k].Rwb, hp[k].Rwb, 72); memcpy(kfs[k].twb, hp[k].twb, 24);
        memcpy(kfs[k].velocity, sv[k].v, 24); memcpy(kfs[k].gyro_bias, sv[k].bg, 24); memcpy(kfs[k].acc_bias, sv[k].ba, 24);
    }
    return done;
}

}  // extern "C"

namespace {


// ---- lock-step batch of LocalLVIBA windows (tc2li_local_lvi_bundle_adjustment_batch) -------------------------------------------
// The phases of ba_batch_lockstep with the host steps of tc2li_local_lvi_bundle_adjustment between them: the inertial edges'
// normal equations (InertialTerm, overlapping the linearisation kernels), the dense reduced system [6 per free pose | 9 per free
// keyframe with IMU state] = Schur complement of the landmarks + inertial + LiDAR blocks, its LDL^T, and the inertial cost of every
// trial state (the trial ImuCamPose states come back through one copy launch per phase).  Same kernel bodies and host arithmetic
// as the one-window entry point: a window gives the same result alone and in a batch.
struct LviWindow {
    const tc2li_lvi_problem* p = nullptr;
    BaWorkspace* ws = nullptr;
    VisualProblem vp;
    BalmTerm* lidar = nullptr;
    InertialTerm inertial;
    std::vector<uint8_t> extra_used, imu_used;
    std::vector<ImuPose> hp, hp_trial;
    std::vector<ImuVertexState> sv, sv_trial;
    std::vector<double> rhs, bfull, x;
    ReducedSolver solver;
    bool dev_solve = false;    // the reduced system on the device (k_lvi_solve_b)
    size_t blob_bytes = 0;
    double lambda = -1, ni = 2, currentChi = 0, tempChi = 0, iniChi = 0, rho = 0, scale = 0, chi_imu = 0, last_chi = 0;
    int n_bad = 0, done = 0, trials_total = 0, qmax = 0, it = 0, rc = 0;
    int parity = 0;  // 1: the accepted estimate lives in the trial buffers of the slot
    bool ok = true, ok2 = true, want_maxdiag = false;
    bool wants_hpp() const { return false; }
    bool stopped() const { return p->stop_flag && *p->stop_flag; }
    bool wants_iteration() const { return rc >= 0 && it < p->iterations && !stopped() && ok; }
};
LockstepContext& lvi_lockstep_ctx(int group) { return shutdown_owned<LockstepContexts, 1>().c[group]; }

bool lvi_batch_lockstep(const tc2li_lvi_problem* problems, int n, const tc2li_imu_calib* calib, const tc2li_camera* cam, WorkerPool& pool, int32_t* results,
                        int group = 0) {
    LockstepContext& C = lvi_lockstep_ctx(group);
    std::lock_guard<std::mutex> lk(C.mu);
    const BaOptions opt = BaOptions::read();
    for (int i = 0; i < n; ++i)
        if (problems[i].lidar && problems[i].lidar->n_keyframes > 7) return false;
    if (!C.st) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&C.st, hipStreamNonBlocking, hi) != hipSuccess &&
            hipStreamCreateWithFlags(&C.st, hipStreamNonBlocking) != hipSuccess) { C.st = nullptr; return false; }
    }
    hipStream_t st = C.st;
    while ((int)C.ws.size() < n) C.ws.emplace_back(new BaWorkspace());
    // the slot table and the steps' staging area, as in ba_batch_lockstep: the table goes up once, a phase's state in the kernels' arguments
    if (n > 65535) return false;
    constexpr size_t kXpStride = kBaXpStride;
    const size_t table_bytes = (size_t)n * sizeof(BaBatchSlot), xp_bytes = (size_t)n * kXpStride * sizeof(double);
    if (C.d_table.ensure(table_bytes + xp_bytes) != hipSuccess || C.h_table.ensure(table_bytes + xp_bytes) != hipSuccess) return false;
    BaBatchSlot* const h_slots = (BaBatchSlot*)C.h_table.p;
    double* const h_xp_area = (double*)(C.h_table.p + table_bytes);
    const BaBatchSlot* const d_table = (const BaBatchSlot*)C.d_table.p;
    double* const d_xp_area = (double*)(C.d_table.p + table_bytes);
    std::vector<LviWindow> W(n);
    const bool kTiming = opt.timing;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // setup, linearise (device + host edges), host after linearise, schur, solve, trial, trial cost, results
    const double t_begin = kTiming ? now() : 0;
    double t_mark = t_begin;
    auto lap = [&](int k) { if (kTiming) { const double t = now(); tm[k] += t - t_mark; t_mark = t; } };
    // ---- setup: argument checks, inertial links, plane extraction (device, queued first), uploads ----
    std::vector<int> rc_lidar(n, 0);
    std::vector<std::vector<CopyTask>> deferred(2 * (size_t)n);
    if (C.h_cut.ensure(std::max(n, 1)) != hipSuccess) return false;
    for (int i = 0; i < n; ++i) C.h_cut.p[i].n_points = 0;
    auto setup_task = [&](int task) {  // two tasks per window: structure + uploads (even), the LiDAR window (odd)
        CopySink sink(&deferred[task]);
        const int i = task >> 1;
        LviWindow& w = W[i];
        const tc2li_lvi_problem& p = problems[i];
        const bool args_ok = p.keyframes && p.fixed && p.has_imu && p.points3 && p.edges && p.n_keyframes > 0 && p.n_points > 0 && p.n_edges > 0 &&
                             p.n_links >= 0 && (p.n_links == 0 || p.links) && p.iterations >= 0 && (!p.lidar || p.Tbl);
        bool lidar_ok = true;
        if (args_ok && p.lidar) {
            if (p.lidar->n_keyframes < 1 || !p.lidar->pose_index) lidar_ok = false;
            else for (int k = 0; k < p.lidar->n_keyframes; ++k) if (p.lidar->pose_index[k] < 0 || p.lidar->pose_index[k] >= p.n_keyframes) lidar_ok = false;
        }
        if (task & 1) {
            if (!args_ok || !lidar_ok || !p.lidar) return;
            rc_lidar[i] = C.ws[i]->lidar.build_body(p.keyframes, sizeof(tc2li_inertial_keyframe), p.n_keyframes, p.lidar, p.Tbl, sizeof(ImuPose), st, &C.h_cut.p[i]);
            return;
        }
        w.p = &p; w.ws = C.ws[i].get();
        if (!args_ok) { set_error("tc2li_local_lvi_bundle_adjustment_batch: problem %d: invalid argument", i); w.rc = TC2LI_ERR_INVALID; return; }
        if (!lidar_ok) { set_error("lidar window: invalid argument or pose_index out of range"); w.rc = TC2LI_ERR_INVALID; return; }
        if (p.stats) memset(p.stats, 0, sizeof(*p.stats));
        if (p.lidar_stats) memset(p.lidar_stats, 0, sizeof(*p.lidar_stats));
        const int n_kfs = p.n_keyframes;
        w.rc = w.inertial.prepare(p.links, p.n_links, p.has_imu, n_kfs, w.extra_used);
        if (w.rc < 0) return;
        w.imu_used = w.extra_used;
        if (p.lidar) for (int k = 0; k < p.lidar->n_keyframes; ++k) w.extra_used[p.lidar->pose_index[k]] = 1;
        w.rc = w.vp.setup(*w.ws, nullptr, p.fixed, n_kfs, p.points3, p.n_points, p.edges, p.n_edges, cam, w.extra_used.data(), st);
        if (w.rc < 0) return;
        w.inertial.number(p.fixed, p.has_imu, w.imu_used, n_kfs, w.vp.pose_var, w.vp.np);
        if (p.lidar)   // the LiDAR term's blocks of the reduced system (BalmTerm::add_quadratic_form)
            for (int a = 0; a < p.lidar->n_keyframes; ++a)
                for (int b = 0; b < p.lidar->n_keyframes; ++b) {
                    const int vi = w.vp.pose_var[p.lidar->pose_index[a]], vj = w.vp.pose_var[p.lidar->pose_index[b]];
                    if (vi >= 0 && vj >= 0) w.inertial.note_block(6 * vi, 6 * vj, 6, 6);
                }
        BaWorkspace& ws = *w.ws;
        w.hp.resize(n_kfs); w.hp_trial.resize(n_kfs); w.sv.resize(n_kfs); w.sv_trial.resize(n_kfs);
        if (ws.d_iposes.ensure(n_kfs) != hipSuccess || ws.d_iposes_trial.ensure(n_kfs) != hipSuccess || ws.h_iposes.ensure(n_kfs) != hipSuccess ||
            ws.h_iposes_up.ensure(n_kfs) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
        for (int k = 0; k < n_kfs; ++k) {
            const tc2li_inertial_keyframe& kf = p.keyframes[k];
            memcpy(w.hp[k].Rcw, kf.Rcw, 72); memcpy(w.hp[k].tcw, kf.tcw, 24); memcpy(w.hp[k].Rwb, kf.Rwb, 72); memcpy(w.hp[k].twb, kf.twb, 24);
            w.hp[k].its = 0; w.hp[k].pad_ = 0;
            memcpy(w.sv[k].v, kf.velocity, 24); memcpy(w.sv[k].bg, kf.gyro_bias, 24); memcpy(w.sv[k].ba, kf.acc_bias, 24);
        }
        memcpy(ws.h_iposes_up.p, w.hp.data(), n_kfs * sizeof(ImuPose));
        if (upload_or_defer(ws.d_iposes.p, ws.h_iposes_up.p, n_kfs * sizeof(ImuPose), st) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
        BaProblemDev& pb = w.vp.pb;
        pb.inertial = 1; pb.iposes = ws.d_iposes.p; pb.iposes_trial = ws.d_iposes_trial.p;
        w.vp.decide_trial_fused();
        memcpy(&pb.calib, calib, sizeof(ImuCalib));
        const int nn = w.inertial.n;
        w.dev_solve = w.inertial.device_solve_ok();
        if (w.dev_solve) {
            const int np1 = w.vp.np;
            if (ws.lvi.ensure(np1, nn - np1) != hipSuccess || ws.d_S.ensure((size_t)np1 * np1) != hipSuccess || ws.d_bs.ensure(2 * (size_t)np1) != hipSuccess ||
                ws.d_xp.ensure(nn) != hipSuccess || ws.h_xp.ensure(nn) != hipSuccess || ws.h_ok.ensure(1) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
        }
        w.rhs.assign(std::max(nn, 1), 0.0); w.bfull.assign(std::max(nn, 1), 0.0); w.x.assign(std::max(nn, 1), 0.0);
    };
    pool.parallel_for(n, [&](int i) { setup_task(2 * i + 1); });
    if (!plane_extraction_begin(C, deferred, n, st)) { (void)hipStreamSynchronize(st); return false; }
    pool.parallel_for(n, [&](int i) { setup_task(2 * i); });
    if (!plane_extraction_finish(C, n, rc_lidar, st)) return false;
    for (int i = 0; i < n; ++i) {
        if (W[i].rc < 0 || !problems[i].lidar) continue;
        if (rc_lidar[i] < 0) W[i].rc = rc_lidar[i]; else W[i].lidar = &C.ws[i]->lidar;
    }
    for (int i = 0; i < n; ++i)
        if (W[i].rc >= 0 && W[i].lidar && W[i].lidar->n_planes > 2048) { (void)hipStreamSynchronize(st); return false; }
    // the reduced systems on the device or on the host, the whole call one way: a window decides for itself (InertialTerm::device_solve_ok), and a call
    // whose windows disagree is handed back to the one-window entry points -- every window then runs exactly as it would alone
    bool dev_solve = false;
    int max_lvi_np = 0, max_lvi_ni = 0;
    {
        int n_dev = 0, n_ok = 0;
        for (int i = 0; i < n; ++i) if (W[i].rc >= 0) { ++n_ok; if (W[i].dev_solve) { ++n_dev; max_lvi_np = std::max(max_lvi_np, W[i].vp.np); max_lvi_ni = std::max(max_lvi_ni, W[i].inertial.n - W[i].vp.np); } }
        if (n_dev && n_dev != n_ok) {
            // the windows disagree: the majority stays in lock step, the others are handed back one by one (the whole group used to
            // go back -- one window without a velocity vertex sent 43 through the serial path)
            const bool keep_dev = 2 * n_dev >= n_ok;
            max_lvi_np = max_lvi_ni = 0;
            for (int i = 0; i < n; ++i) {
                if (W[i].rc < 0) continue;
                if (W[i].dev_solve != keep_dev) { W[i].rc = kLockstepDeclined; continue; }
                if (keep_dev) { max_lvi_np = std::max(max_lvi_np, W[i].vp.np); max_lvi_ni = std::max(max_lvi_ni, W[i].inertial.n - W[i].vp.np); }
            }
            n_dev = keep_dev ? n_dev : 0;
        }
        dev_solve = n_dev > 0;
    }
    std::vector<int> all_windows(n);
    for (int i = 0; i < n; ++i) all_windows[i] = i;
    bool all_block_parts = true;
    BaBatchExtent X = batch_extent(W, all_windows, &all_block_parts);
    X.inertial = 1;
    // The sums behind a trial's errors (k_ba_trial_reduce_b: two workgroups per window) are taken by the LAST workgroup of the window's
    // error pass (a ticket per window, ba_kernels.hip: ba_last_of): one launch fewer per LM trial -- BA stage alone 15.0-15.2 against 15.2-15.6 ms
    // per 128 windows, the loop 28.3 / 28.7 against 28.4 / 28.9 ms.  (The same for the Schur product's closing sums measured SLOWER, 29.5-29.8
    // against 28.4-28.6 ms: one workgroup adding ten parts of 2 700 values is a longer tail than the 21 workgroups of k_ba_schur_finish_b
    // are a launch; removed.)
    {
        X.fuse_trial = all_block_parts ? 1 : 0;
        // round 5: the linearisation's closing sums (pose blocks, robust cost, largest diagonals) and the plane Hessian's
        // chunk sums the same way -- an iteration's linearisation phase is then two launches instead of four or five.  Built, bit-identical
        // (the same sums in the same order), and measured in the whole loop, three A/B pairs in one call: 26.24 / 26.26 / 26.27 ms per step fused
        // against 26.08 / 25.95 / 26.03 separate (mapping workers 25.1-25.7 against 24.5-25.3): the loop is bound by the kernels' combined
        // throughput, not by the number of launches in a chain, and one workgroup's tail is longer than the small launch it replaces.  Off by
        // default; TC2LI_BA_FUSE_LIN=1 (read per call) switches it on.
        X.fuse_linearize = opt.fuse_linearize ? 1 : 0;
    }
    auto fill_slot = [&](int i) {
        LviWindow& w = W[i];
        BaBatchSlot& s = h_slots[i];
        s.pb = w.vp.pb;
        // (the LM decisions of the inertial windows stay on the host: no device-side state -- the table's memory is reused from call to call)
        s.lm = nullptr; s.lm_host = nullptr; s.stop_host = nullptr; s.lidar_JH = nullptr; s.lambda_init = 0; s.lidar_information = 0; s.iterations = 0; s.lm_pad_ = 0;
        s.n_slices = w.vp.n_slices; s.k_per_slice = w.vp.k_per_slice; s.has_lidar = w.lidar != nullptr; s.pad_ = 0;
        double* sc = w.ws->h_scal.p;
        s.chi_out = sc; s.maxdiag_out = sc + 1; s.scale_out = sc + 3; s.chi_trial_out = sc + 4;
        s.S_out = w.ws->h_S.p; s.bs_out = w.ws->h_bs.p; s.xp = w.ws->h_xp.p; s.depth_out = w.ws->d_depth.p;
        s.hpp_out = nullptr; s.bp_host = nullptr; s.Hl = s.bl_lidar = nullptr; s.x_dev = s.x_host = nullptr; s.ok_host = nullptr;
        s.iposes_host = w.ws->h_iposes.p;  // the trial kernel leaves the trial ImuCamPose states there for the host's inertial cost
        s.lvi = LviSolveDev{};
        if (w.dev_solve) {  // Schur product, solve and trial in one queue: S and b_s stay on the device, b_p and the step (all n unknowns) come back
            s.S_out = w.ws->d_S.p; s.bs_out = w.ws->d_bs.p; s.bp_host = w.ws->h_bs.p + w.vp.np;
            s.xp = s.x_dev = w.ws->d_xp.p; s.x_host = w.ws->h_xp.p; s.ok_host = w.ws->h_ok.p;
            s.lvi = w.ws->lvi.dev;
        }
        if (w.lidar) s.balm = w.lidar->dev; else s.balm = BalmDev{};
    };
    bool failed = false;
    {   // the table and everything the setup deferred: one launch
        for (int i = 0; i < n; ++i) if (W[i].rc >= 0) fill_slot(i); else h_slots[i] = BaBatchSlot{};
        size_t n_tasks = 1, max_bytes = table_bytes;
        for (const auto& d : deferred) n_tasks += d.size();
        if (C.h_tasks.ensure(n_tasks) != hipSuccess) return false;
        size_t at = 0;
        C.h_tasks.p[at++] = CopyTask{C.d_table.p, C.h_table.p, table_bytes};
        for (const auto& d : deferred) for (const CopyTask& t : d) { C.h_tasks.p[at++] = t; max_bytes = std::max(max_bytes, t.bytes); }
        launch_copy_tasks(C.h_tasks.p, (int)n_tasks, max_bytes, st);
    }
    // No upload launch for the steps: the trial kernels read a window's step (<= 1.5 KB, once per workgroup) from the pinned staging area
    // over the bus -- ten launches fewer per call at the same speed (BA stage alone 15.3 / 15.4 ms per 128 windows, the loop 28.5-28.7 /
    // 28.6-29.1 ms per step against the one-entry k_copy_tasks launch of rounds 3-4)
    constexpr bool xp_pinned = true;
    auto stage_steps = [&](const std::vector<int>& step) {
        for (size_t k = 0; k < step.size(); ++k) {
            const LviWindow& w = W[step[k]];
            if (w.vp.np <= 0 || w.vp.np > (int)kXpStride) continue;
            memcpy(h_xp_area + k * kXpStride, w.ws->h_xp.p, (size_t)w.vp.np * sizeof(double));
        }
        if (xp_pinned) return;  // the trial kernels read the steps where they are
        if (C.h_table_task.ensure(1) != hipSuccess) { failed = true; return; }
        C.h_table_task.p[0] = CopyTask{d_xp_area, h_xp_area, step.size() * kXpStride * sizeof(double)};
        launch_copy_tasks(C.h_table_task.p, 1, 4096, st);
    };
    auto pieces = [&](const std::vector<int>& list, const double* xp_area, auto&& fn) { for_phase_pieces(d_table, xp_area, W, list, fn); };
    // (the group's thread spins on its stream between the phases: sleeping on a blocking event instead was measured in round 4 -- the same host
    // CPU time, 13.7 of the 16 CPUs a one-GPU box's cgroup grants, and a step 0.3 ms longer)
    auto sync = [&] { if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) failed = true; };

    // OptimizationAlgorithmLevenberg::solve's gain ratio and damping update for the windows of a trial; returns those that try again
    auto lm_decisions = [&](const std::vector<int>& trial) {
        std::vector<int> again;
        for (int i : trial) {
            LviWindow& w = W[i];
            if (!w.ok2) w.tempChi = std::numeric_limits<double>::max();
            w.rho = w.currentChi - w.tempChi;
            w.scale += 1e-3;
            w.rho /= w.scale;
            if (w.rho > 0 && std::isfinite(w.tempChi)) {
                w.lambda = lm_lambda_accepted(w.lambda, w.rho);
                w.ni = 2;
                w.currentChi = w.tempChi;
                std::swap(w.vp.pb.iposes, w.vp.pb.iposes_trial);
                std::swap(w.vp.pb.points, w.vp.pb.points_trial);
                w.parity ^= 1;
                w.hp.swap(w.hp_trial);
                w.sv.swap(w.sv_trial);
            } else {
                w.lambda *= w.ni;
                w.ni *= 2;
            }
            w.qmax++;
            w.trials_total++;
            if (w.rho < 0 && w.qmax < 10 && !w.stopped()) again.push_back(i);
        }
        return again;
    };
    lap(0);
    for (;;) {
        std::vector<int> active, with_lidar;
        for (int i = 0; i < n; ++i) if (W[i].wants_iteration()) active.push_back(i);
        if (active.empty() || failed) break;
        lap(7);
        // ---- phase A: linearisation at the accepted estimate; the inertial edges on the host meanwhile ----
        bool any_maxdiag = false;
        for (int i : active) {
            LviWindow& w = W[i];
            w.want_maxdiag = w.it == 0 && !(w.p->lambda_init > 0);
            any_maxdiag |= w.want_maxdiag;
            if (w.lidar) with_lidar.push_back(i);
        }
        pieces(active, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_linearize(ph, cnt, X, any_maxdiag, st); });
        // computeActiveErrors + linearizeOplus of the LiDAR edge: the residual at the accepted estimate and the Hessian, every iteration
        // (the one-window path's enqueue_error + enqueue_linearization)
        pieces(with_lidar, nullptr, [&](const BaPhase& ph, int cnt) {
            balm_batch_launch_residual(ph, cnt, false, st);
            balm_batch_launch_hessian(ph, cnt, X, st);
        });
        // (the first trial's Schur product behind the linearisation, as in ba_batch_lockstep: the host's inertial edges, the LiDAR term's change of
        // variables and the upload of the reduced system's inertial part run beside it)
        constexpr bool kPreSchur = true;
        bool pre_schur = kPreSchur && !any_maxdiag;
        if (pre_schur) {
            for (int i : active) if (W[i].it == 0) W[i].lambda = W[i].p->lambda_init;  // (what the host's part sets below)
            pieces(active, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_schur(ph, cnt, X, st); });
        }
        pool.parallel_for((int)active.size(), [&](int k) { LviWindow& w = W[active[k]]; w.chi_imu = w.inertial.cost(w.hp, w.sv, true); });
        lap(8);
        sync();
        if (failed) break;
        lap(1);
        pool.parallel_for((int)active.size(), [&](int k) {
            LviWindow& w = W[active[k]];
            const double* sc = w.ws->h_scal.p;
            const int np = w.vp.np, nn = w.inertial.n;
            double chi_imu = w.chi_imu;
            if (w.lidar) {
                if (w.it == 0) w.lidar->finish_error();  // the computeActiveErrors() before optimize() (OptimizerWithLidar.cc:978)
                w.lidar->finish_error();
                chi_imu += w.lidar->chi2();
                w.lidar->finish_linearization();
                w.lidar->add_quadratic_form(w.vp.pose_var.data(), nn, w.inertial.Hi.data(), w.inertial.bi.data());
            }
            w.solver.set_pattern(w.inertial.Hi.data(), nn, np, !w.dev_solve);
            if (w.dev_solve) {
                if (w.solver.band() > kLviBand) w.rc = TC2LI_ERR_INVALID;  // (device_solve_ok bounds the band by the links: not reached)
                else w.blob_bytes = w.ws->lvi.pack(w.solver, w.inertial.Hi.data(), w.inertial.bi.data());
            }
            w.currentChi = chi_imu + sc[0];
            w.tempChi = w.currentChi;
            w.iniChi = w.currentChi;
            if (w.it == 0) {
                if (w.p->stats) w.p->stats->initial_chi2 = w.currentChi;
                w.last_chi = w.currentChi;
                w.lambda = w.p->lambda_init;
                if (!(w.p->lambda_init > 0)) {  // computeLambdaInit over the whole diagonal (not used by the reference's settings)
                    double mx = std::max(sc[1], sc[2]);
                    for (int j = np; j < nn; ++j) mx = std::max(mx, std::fabs(w.inertial.Hi[(size_t)j * nn + j]));
                    w.lambda = 1e-5 * mx;
                }
                w.ni = 2;
                w.n_bad = 0;
            }
            w.rho = 0;
            w.qmax = 0;
        });
        if (dev_solve) {  // this linearisation's inertial / LiDAR part of the reduced systems goes up: one launch
            if (C.h_tasks.ensure(active.size()) != hipSuccess) { failed = true; break; }
            size_t max_bytes = 0;
            for (size_t k = 0; k < active.size(); ++k) {
                LviWindow& w = W[active[k]];
                if (w.rc < 0) { failed = true; break; }
                C.h_tasks.p[k] = CopyTask{w.ws->lvi.d_blob.p, w.ws->lvi.h_blob.p, w.blob_bytes};
                max_bytes = std::max(max_bytes, w.blob_bytes);
            }
            if (failed) break;
            launch_copy_tasks(C.h_tasks.p, (int)active.size(), max_bytes, st);
        }
        // ---- trials ----
        lap(2);
        std::vector<int> trial = active;
        while (!trial.empty() && !failed && dev_solve) {
            // Schur product, solve, trial estimate and its cost in one queue; the host sees the step, whether the factorisation went through, and
            // the sums at the one synchronisation
            std::vector<int> trial_lidar;
            for (int i : trial) if (W[i].lidar) trial_lidar.push_back(i);
            const bool have_schur = pre_schur;  // (this trial's product came with the linearisation)
            pre_schur = false;
            pieces(trial, nullptr, [&](const BaPhase& ph, int cnt) {
                if (!have_schur) ba_batch_launch_schur(ph, cnt, X, st);
                lvi_batch_launch_solve(ph, cnt, max_lvi_np, max_lvi_ni, st);
                ba_batch_launch_trial(ph, cnt, X, st);
            });
            if (X.any_trial_unfused) pieces(trial_lidar, nullptr, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, true, st); });
            sync();
            if (failed) break;
            lap(3);
            pool.parallel_for((int)trial.size(), [&](int k) {
                LviWindow& w = W[trial[k]];
                BaWorkspace& ws = *w.ws;
                const int np = w.vp.np, nn = w.inertial.n;
                const std::vector<double>& bi = w.inertial.bi;
                w.ok2 = ws.h_ok.p[0] != 0;
                memcpy(w.x.data(), ws.h_xp.p, (size_t)nn * sizeof(double));
                w.scale = 0;
                for (int j = 0; j < nn; ++j) {
                    const double bfull = bi[j] + (j < np ? ws.h_bs.p[np + j] : 0.0);
                    w.scale += w.x[j] * (w.lambda * w.x[j] + bfull);
                }
                if (!w.ok2) return;
                w.sv_trial = w.sv;
                for (int q = 0; q < w.p->n_keyframes; ++q)
                    if (w.inertial.imu_var[q] >= 0) {
                        const double* u = &w.x[np + 9 * w.inertial.imu_var[q]];
                        for (int c = 0; c < 3; ++c) { w.sv_trial[q].v[c] += u[c]; w.sv_trial[q].bg[c] += u[3 + c]; w.sv_trial[q].ba[c] += u[6 + c]; }
                    }
                memcpy(w.hp_trial.data(), ws.h_iposes.p, w.p->n_keyframes * sizeof(ImuPose));
                w.tempChi = w.inertial.cost(w.hp_trial, w.sv_trial, false) + ws.h_scal.p[4];
                if (w.lidar) { w.lidar->finish_error(); w.tempChi += w.lidar->chi2(); }
                w.scale += ws.h_scal.p[3];
                w.last_chi = w.tempChi;
            });
            lap(6);
            trial = lm_decisions(trial);
        }
        while (!trial.empty() && !failed && !dev_solve) {
            if (pre_schur) pre_schur = false;  // (this trial's product came with the linearisation)
            else {
                pieces(trial, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_schur(ph, cnt, X, st); });
                sync();
                if (failed) break;
            }
            lap(3);
            pool.parallel_for((int)trial.size(), [&](int k) {
                LviWindow& w = W[trial[k]];
                BaWorkspace& ws = *w.ws;
                const int np = w.vp.np, nn = w.inertial.n;
                const std::vector<double>&Hi = w.inertial.Hi, &bi = w.inertial.bi;
                // reduced system: [S_visual + H_inertial(poses)   H_inertial(poses, imu) ; ...   H_inertial(imu) + lambda I] (reduced_solve.hpp)
                for (int j = 0; j < nn; ++j) {
                    w.bfull[j] = bi[j] + (j < np ? ws.h_bs.p[np + j] : 0.0);
                    w.rhs[j] = bi[j] + (j < np ? ws.h_bs.p[j] : 0.0);
                }
                w.ok2 = nn == 0 ? true : w.solver.factorise(Hi.data(), ws.h_S.p, w.lambda);
                if (w.ok2 && nn) w.solver.solve(w.rhs.data(), w.x.data());
                w.scale = 0;
                for (int j = 0; j < nn; ++j) w.scale += w.x[j] * (w.lambda * w.x[j] + w.bfull[j]);
                if (w.ok2 && np) memcpy(ws.h_xp.p, w.x.data(), np * sizeof(double));
            });
            lap(4);
            std::vector<int> step, step_lidar;
            for (int i : trial) if (W[i].ok2) { step.push_back(i); if (W[i].lidar) step_lidar.push_back(i); }
            if (!step.empty()) {
                stage_steps(step);
                // (the trial ImuCamPose states come back through slot.iposes_host, written by the trial kernel: a copy launch per trial before)
                pieces(step, xp_pinned ? h_xp_area : d_xp_area, [&](const BaPhase& ph, int cnt) { ba_batch_launch_trial(ph, cnt, X, st); });
                if (X.any_trial_unfused) pieces(step_lidar, nullptr, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, true, st); });  // (windows with pb.trial_fused: inside the trial launch)
                pool.parallel_for((int)step.size(), [&](int k) {  // velocity / bias part of the step, on the host
                    LviWindow& w = W[step[k]];
                    const int np = w.vp.np;
                    w.sv_trial = w.sv;
                    for (int q = 0; q < w.p->n_keyframes; ++q)
                        if (w.inertial.imu_var[q] >= 0) {
                            const double* u = &w.x[np + 9 * w.inertial.imu_var[q]];
                            for (int c = 0; c < 3; ++c) { w.sv_trial[q].v[c] += u[c]; w.sv_trial[q].bg[c] += u[3 + c]; w.sv_trial[q].ba[c] += u[6 + c]; }
                        }
                });
                sync();
                if (failed) break;
                lap(5);
                pool.parallel_for((int)step.size(), [&](int k) {
                    LviWindow& w = W[step[k]];
                    memcpy(w.hp_trial.data(), w.ws->h_iposes.p, w.p->n_keyframes * sizeof(ImuPose));
                    w.tempChi = w.inertial.cost(w.hp_trial, w.sv_trial, false) + w.ws->h_scal.p[4];
                    if (w.lidar) { w.lidar->finish_error(); w.tempChi += w.lidar->chi2(); }
                    w.scale += w.ws->h_scal.p[3];
                    w.last_chi = w.tempChi;
                });
            }
            lap(6);
            trial = lm_decisions(trial);
        }
        for (int i : active) {
            LviWindow& w = W[i];
            ++w.done;
            ++w.it;
            if (w.p->stats) w.p->stats->final_lambda = w.lambda;
            if (w.qmax == 10 || w.rho == 0) { w.ok = false; continue; }
            if ((w.iniChi - w.currentChi) * 1e3 < w.iniChi) w.n_bad++; else w.n_bad = 0;
            if (w.n_bad >= 3) w.ok = false;
        }
    }
    // ---- results ----
    std::vector<int> all;
    for (int i = 0; i < n; ++i) if (W[i].rc >= 0) all.push_back(i);
    if (!failed && !all.empty()) {
        pieces(all, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_depth(ph, cnt, X, st); });
        size_t n_tasks = 0, max_bytes = 0;
        if (C.h_tasks.ensure(3 * all.size()) != hipSuccess) failed = true;
        for (int i : all) {
            if (failed) break;
            LviWindow& w = W[i];
            const tc2li_lvi_problem& p = *w.p;
            const size_t E = p.n_edges, P = p.n_points;
            const size_t bytes = 3 * P * sizeof(double) + E * sizeof(double) + E;
            if (w.ws->h_result.ensure(bytes) != hipSuccess) { failed = true; break; }
            uint8_t* h = w.ws->h_result.p;
            auto add = [&](void* dst, const void* src, size_t nbytes) { C.h_tasks.p[n_tasks++] = CopyTask{dst, src, nbytes}; max_bytes = std::max(max_bytes, nbytes); };
            add(h, w.vp.pb.points, 3 * P * sizeof(double));
            if (p.edge_chi2) add(h + 3 * P * sizeof(double), w.ws->d_chi2.p, E * sizeof(double));
            if (p.edge_depth_positive) add(h + 3 * P * sizeof(double) + E * sizeof(double), w.ws->d_depth.p, E);
        }
        if (!failed) launch_copy_tasks(C.h_tasks.p, (int)n_tasks, max_bytes, st);
        sync();
        if (!failed)
            pool.parallel_for((int)all.size(), [&](int k) {
                LviWindow& w = W[all[k]];
                const tc2li_lvi_problem& p = *w.p;
                const size_t E = p.n_edges, P = p.n_points;
                const uint8_t* h = w.ws->h_result.p;
                memcpy(p.points3, h, 3 * P * sizeof(double));
                if (p.edge_chi2) memcpy(p.edge_chi2, h + 3 * P * sizeof(double), E * sizeof(double));
                if (p.edge_depth_positive) memcpy(p.edge_depth_positive, h + 3 * P * sizeof(double) + E * sizeof(double), E);
                for (int q = 0; q < p.n_keyframes; ++q) {
                    tc2li_inertial_keyframe& kf = p.keyframes[q];
                    memcpy(kf.Rcw, w.hp[q].Rcw, 72); memcpy(kf.tcw, w.hp[q].tcw, 24); memcpy(kf.Rwb, w.hp[q].Rwb, 72); memcpy(kf.twb, w.hp[q].twb, 24);
                    memcpy(kf.velocity, w.sv[q].v, 24); memcpy(kf.gyro_bias, w.sv[q].bg, 24); memcpy(kf.acc_bias, w.sv[q].ba, 24);
                }
            });
    }
    lap(7);
    if (kTiming) fprintf(stderr, "LVI lock-step timing ms (%d windows): setup %.3f inertial edges (host, kernels queued) %.3f + wait %.3f host-lin %.3f schur %.3f solve %.3f trial %.3f trial-cost %.3f results+rest %.3f total %.3f\n",
                         n, tm[0], tm[8], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], tm[7], now() - t_begin);
    for (int i = 0; i < n; ++i) {
        LviWindow& w = W[i];
        if (w.rc < 0) { results[i] = w.rc; continue; }
        if (failed) { set_error("tc2li_local_lvi_bundle_adjustment_batch: HIP error in the lock-step loop: %s", hipGetErrorString(hipGetLastError())); results[i] = TC2LI_ERR_HIP; continue; }
        const tc2li_lvi_problem& p = *w.p;
        if (p.stats) { p.stats->iterations = w.done; p.stats->trials = w.trials_total; p.stats->n_free_poses = w.vp.n_free; p.stats->final_chi2 = w.last_chi; }
        if (w.lidar && p.lidar_stats) {
            p.lidar_stats->n_planes = w.lidar->n_planes; p.lidar_stats->hessian_evaluations = w.lidar->hessian_evaluations;
            p.lidar_stats->residual = w.lidar->error; p.lidar_stats->chi2 = w.lidar->chi2();
        }
        results[i] = w.done;
    }
    return true;
}

}  // namespace

extern "C" {

int tc2li_local_lvi_bundle_adjustment_batch(const tc2li_lvi_problem* problems, int n_problems, const tc2li_imu_calib* calib, const tc2li_camera* cam,
                                            int max_concurrency, int32_t* results) {
    if (n_problems < 0 || (n_problems > 0 && (!problems || !results)) || !calib || !cam) { set_error("tc2li_local_lvi_bundle_adjustment_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n_problems == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    const BaOptions opt = BaOptions::read();
    const bool kNoLockstep = !opt.lockstep;
    const int kGroups = opt.groups;
    // as in tc2li_local_bundle_adjustment_batch: only the windows of a group that declined go through the one-window path
    std::vector<uint8_t> todo(n_problems, 1);
    if (max_concurrency > 1 && n_problems > 1 && !kNoLockstep) {
        const int groups = std::max(1, std::min(kGroups, n_problems / 2));
        auto run_group = [&](int g) {
            const int b = (int)((long)n_problems * g / groups), e = (int)((long)n_problems * (g + 1) / groups);
            if (lvi_batch_lockstep(problems + b, e - b, calib, cam, named_pool(kPoolLviGroup0 + g), results + b, g))
                for (int i = b; i < e; ++i) todo[i] = results[i] == kLockstepDeclined;
        };
        if (groups == 1) run_group(0);
        else named_pool(kPoolLviTop).parallel_for(groups, run_group);
    }
    // one window after the other (a LiDAR window outside the batched kernels' range, or a batch of one)
    for (int i = 0; i < n_problems; ++i) {
        if (!todo[i]) continue;
        const tc2li_lvi_problem& p = problems[i];
        results[i] = tc2li_local_lvi_bundle_adjustment(p.keyframes, p.fixed, p.has_imu, p.n_keyframes, calib, p.points3, p.n_points, p.edges, p.n_edges, p.links,
                                                       p.n_links, cam, p.iterations, p.lambda_init, p.stop_flag, p.edge_chi2, p.edge_depth_positive, p.stats,
                                                       p.lidar, p.Tbl, p.lidar_stats, private_stream());
    }
    int ok = 0;
    for (int i = 0; i < n_problems; ++i) ok += results[i] >= 0;
    return ok;
}

int tc2li_host_reduced_solve(const double* Hi, const double* S, int n, int np, double lambda, const double* rhs, double* x) {
    if (!Hi || n <= 0 || np < 0 || np > n || (np > 0 && !S) || !rhs || !x) { set_error("tc2li_host_reduced_solve: invalid argument"); return TC2LI_ERR_INVALID; }
    ReducedSolver solver;
    solver.set_pattern(Hi, n, np);
    if (!solver.factorise(Hi, S, lambda)) return 0;
    solver.solve(rhs, x);
    return 1;
}

int tc2li_device_reduced_solve(const double* Hi, const double* S, int n, int np, double lambda, const double* rhs, double* x, void* stream_) {
    if (!Hi || n <= 0 || np <= 0 || np >= n || !S || !rhs || !x) { set_error("tc2li_device_reduced_solve: invalid argument"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    ReducedSolver solver;
    solver.set_pattern(Hi, n, np, false);
    if (np > kLviMaxPoseRows || solver.band() > kLviBand || !lvi_device_solve_available()) {
        set_error("tc2li_device_reduced_solve: %d pose unknowns / band %d: outside the kernel's range (%d / %d)", np, solver.band(), kLviMaxPoseRows, kLviBand);
        return TC2LI_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream_;
    BaWorkspace& ws = ba_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    TC2LI_HIP_CHECK(ws.lvi.ensure(np, n - np)); TC2LI_HIP_CHECK(ws.d_S.ensure((size_t)np * np)); TC2LI_HIP_CHECK(ws.d_bs.ensure(2 * (size_t)np));
    TC2LI_HIP_CHECK(ws.d_xp.ensure(n)); TC2LI_HIP_CHECK(ws.h_xp.ensure(n)); TC2LI_HIP_CHECK(ws.h_ok.ensure(1));
    const size_t bytes = ws.lvi.pack(solver, Hi, rhs);   // (the whole right-hand side as the inertial part's; the visual part b_s is zero)
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.lvi.d_blob.p, ws.lvi.h_blob.p, bytes, hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(ws.d_S.p, S, (size_t)np * np * sizeof(double), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemsetAsync(ws.d_bs.p, 0, 2 * (size_t)np * sizeof(double), st));
    lvi_launch_solve(ws.lvi.dev, ws.d_S.p, ws.d_bs.p, lambda, ws.d_xp.p, ws.h_xp.p, ws.h_ok.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(x, ws.h_xp.p, (size_t)n * sizeof(double));
    return ws.h_ok.p[0] != 0 ? 1 : 0;
}

// The same as ONE lock-step group on the context `group` (as tc2li_local_bundle_adjustment_batch_group): for the mapping workers of a multi-sequence
// camera-LiDAR-inertial system
int tc2li_local_lvi_bundle_adjustment_batch_group(const tc2li_lvi_problem* problems, int n_problems, const tc2li_imu_calib* calib, const tc2li_camera* cam,
                                                  int group, int32_t* results) {
    if (n_problems < 0 || (n_problems > 0 && (!problems || !results)) || !calib || !cam || group < 0 || group >= kMaxLockstepGroups) {
        set_error("tc2li_local_lvi_bundle_adjustment_batch_group: invalid argument (group 0 .. %d)", kMaxLockstepGroups - 1);
        return TC2LI_ERR_INVALID;
    }
    if (n_problems == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    const BaOptions opt = BaOptions::read();
    const bool kNoLockstep = !opt.lockstep;
    bool done = false;
    if (n_problems > 1 && !kNoLockstep) done = lvi_batch_lockstep(problems, n_problems, calib, cam, named_pool(kPoolLviGroup0 + group), results, group);
    {   // a window outside the batched kernels' range or a batch of one: all of them; windows the group handed back: those
        for (int i = 0; i < n_problems; ++i) {
            if (done && results[i] != kLockstepDeclined) continue;
            const tc2li_lvi_problem& p = problems[i];
            results[i] = tc2li_local_lvi_bundle_adjustment(p.keyframes, p.fixed, p.has_imu, p.n_keyframes, calib, p.points3, p.n_points, p.edges, p.n_edges, p.links,
                                                           p.n_links, cam, p.iterations, p.lambda_init, p.stop_flag, p.edge_chi2, p.edge_depth_positive, p.stats,
                                                           p.lidar, p.Tbl, p.lidar_stats, private_stream());
        }
    }
    int ok = 0;
    for (int i = 0; i < n_problems; ++i) ok += results[i] >= 0;
    return ok;
}

}  // extern "C"
