// Host side of tc2li_pose_inertial_optimization_batch (include/tc2li_hip.h): Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame
// (SF/src/Optimizer.cc:2469-2852, 2854-3270) for a batch of frames.  The optimisation itself -- 4 rounds x 10 Gauss-Newton iterations per
// frame -- is one kernel launch for the whole batch (pose_inertial_kernel.hip).  The host prepares what the reference computes once per
// call (EdgeInertial's information with its eigenvalue clamp, the random-walk informations) and, afterwards, the Hessian of the frame's
// new prior mpcpi: the GetHessian* blocks at the final estimate, Optimizer::Marginalize (:2087-2166) over the previous frame and
// ConstraintPoseImu's eigenvalue clamp (G2oTypes.h:721-732) -- a 30 x 30 dense problem per frame.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "inertial_host.hpp"
#include "pose_inertial_device.hpp"

using namespace tc2li;

namespace {

struct PiWorkspace {
    DevBuf<PiProblem> d_probs;
    DevBuf<PiResult> d_results;
    DevBuf<double> d_Xw, d_chi2;
    DevBuf<BaEdge> d_edges;
    DevBuf<uint8_t> d_close, d_outlier;
    PinnedBuf<PiResult> h_results;
    PinnedBuf<uint8_t> h_stage;  // [Xw | edges | close flags] of all frames on their way up, then the outlier flags on their way back
    std::mutex mu;
};
PiWorkspace& pi_ws() { static thread_local PiWorkspace w; return w; }

void state_from(const tc2li_inertial_keyframe& k, PiState& s) {
    memcpy(s.P.Rcw, k.Rcw, 72); memcpy(s.P.tcw, k.tcw, 24); memcpy(s.P.Rwb, k.Rwb, 72); memcpy(s.P.twb, k.twb, 24);
    s.P.its = 0; s.P.pad_ = 0;  // ImuCamPose(Frame*): its = 0
    memcpy(s.v, k.velocity, 24); memcpy(s.bg, k.gyro_bias, 24); memcpy(s.ba, k.acc_bias, 24);
}
void state_to(const PiState& s, tc2li_inertial_keyframe& k) {
    memcpy(k.Rcw, s.P.Rcw, 72); memcpy(k.tcw, s.P.tcw, 24); memcpy(k.Rwb, s.P.Rwb, 72); memcpy(k.twb, s.P.twb, 24);
    memcpy(k.velocity, s.v, 24); memcpy(k.gyro_bias, s.bg, 24); memcpy(k.acc_bias, s.ba, 24);
}

// out[cols x cols] = J^T Omega J (J: rows x cols, Omega: rows x rows, row-major)
void jtoj(const double* J, int rows, int cols, const double* Om, std::vector<double>& out) {
    std::vector<double> T((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) { double s = 0; for (int k = 0; k < rows; ++k) s += Om[rows * r + k] * J[cols * k + c]; T[(size_t)r * cols + c] = s; }
    out.assign((size_t)cols * cols, 0.0);
    for (int i = 0; i < cols; ++i)
        for (int j = 0; j < cols; ++j) { double h = 0; for (int r = 0; r < rows; ++r) h += J[cols * r + i] * T[(size_t)r * cols + j]; out[(size_t)i * cols + j] = h; }
}

// The Hessian of the new prior (Optimizer.cc:2778-2821 / :3200-3267) from the final states and the inlier edges' pose block
void prior_hessian(const PiProblem& pb, const PiResult& R, double Hn[225]) {
    using namespace inertial_detail;
    double e9[9], J[9 * 24];
    pi_inertial_edge(pb.pre, R.other, R.cur, e9, J);
    double Hv[36];
    {
        int h = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { Hv[6 * r + c] = R.Hv[h]; Hv[6 * c + r] = R.Hv[h]; ++h; }
    }
    memset(Hn, 0, 225 * sizeof(double));
    if (!pb.last_frame) {
        double J2[81];
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) J2[9 * r + c] = J[24 * r + 15 + c];
        std::vector<double> H2;
        jtoj(J2, 9, 9, pb.pre.info, H2);
        for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) Hn[15 * r + c] += H2[9 * r + c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Hn[15 * (9 + r) + 9 + c] += pb.pre.infoG[3 * r + c]; Hn[15 * (12 + r) + 12 + c] += pb.pre.infoA[3 * r + c]; }
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hn[15 * r + c] += Hv[6 * r + c];
    } else {
        std::vector<double> H30(900, 0.0), H24, Hp;
        jtoj(J, 9, 24, pb.pre.info, H24);
        for (int r = 0; r < 24; ++r) for (int c = 0; c < 24; ++c) H30[30 * r + c] += H24[24 * r + c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const double g = pb.pre.infoG[3 * r + c], a = pb.pre.infoA[3 * r + c];
                H30[30 * (9 + r) + 9 + c] += g; H30[30 * (9 + r) + 24 + c] -= g; H30[30 * (24 + r) + 9 + c] -= g; H30[30 * (24 + r) + 24 + c] += g;
                H30[30 * (12 + r) + 12 + c] += a; H30[30 * (12 + r) + 27 + c] -= a; H30[30 * (27 + r) + 12 + c] -= a; H30[30 * (27 + r) + 27 + c] += a;
            }
        double e15[15], Jr[9], Jt[9], Jp[225];
        pi_prior_edge(pb.prior, R.other, e15, Jr, Jt);
        memset(Jp, 0, sizeof(Jp));
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Jp[15 * r + c] = Jr[3 * r + c]; Jp[15 * (3 + r) + 3 + c] = Jt[3 * r + c]; }
        for (int k = 6; k < 15; ++k) Jp[15 * k + k] = 1.0;
        jtoj(Jp, 15, 15, pb.prior.H, Hp);
        for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) H30[30 * r + c] += Hp[15 * r + c];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H30[30 * (15 + r) + 15 + c] += Hv[6 * r + c];
        // Marginalize(H, 0, 14): pseudo-inverse of the previous frame's block through its (symmetric) eigen decomposition, |w| > 1e-6
        std::vector<double> Bm(225), w, V, pinv(225, 0.0);
        for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) Bm[15 * r + c] = H30[30 * r + c];
        sym_eigen(Bm, 15, w, V);
        for (int k = 0; k < 15; ++k) {
            if (!(std::fabs(w[k]) > 1e-6)) continue;
            const double iw = 1.0 / w[k];
            for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) pinv[15 * r + c] += V[15 * r + k] * iw * V[15 * c + k];
        }
        for (int r = 0; r < 15; ++r)
            for (int c = 0; c < 15; ++c) {
                double s = 0;
                for (int i = 0; i < 15; ++i) { double t = 0; for (int j = 0; j < 15; ++j) t += pinv[15 * i + j] * H30[30 * j + 15 + c]; s += H30[30 * (15 + r) + i] * t; }
                Hn[15 * r + c] = H30[30 * (15 + r) + 15 + c] - s;
            }
    }
    // ConstraintPoseImu: eigenvalues below 1e-12 cleared
    std::vector<double> A(Hn, Hn + 225), w, V;
    sym_eigen(A, 15, w, V);
    for (double& x : w) if (x < 1e-12) x = 0;
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += V[15 * r + k] * w[k] * V[15 * c + k]; Hn[15 * r + c] = s; }
}

}  // namespace

extern "C" int tc2li_pose_inertial_optimization_batch(tc2li_pose_inertial_problem* problems, int n_frames, const tc2li_imu_calib* calib,
                                                      const tc2li_camera* cam, int32_t* results, void* stream_) {
    if (n_frames < 0 || (n_frames > 0 && (!problems || !calib || !cam))) { set_error("tc2li_pose_inertial_optimization_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n_frames == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    std::vector<PiProblem> probs(n_frames);
    size_t total = 0;
    for (int f = 0; f < n_frames; ++f) {
        const tc2li_pose_inertial_problem& p = problems[f];
        if (p.n_edges < 0 || (p.n_edges > 0 && (!p.Xw || !p.edges || !p.close_point || !p.outlier)) || !p.preintegrated || !p.preintegrated_rw ||
            (p.last_frame && !p.prior)) {
            set_error("tc2li_pose_inertial_optimization_batch: frame %d: invalid argument", f);
            return TC2LI_ERR_INVALID;
        }
        probs[f].edge_off = (int32_t)total;
        total += (size_t)p.n_edges;
    }
    PiWorkspace& w = pi_ws();
    std::lock_guard<std::mutex> lk(w.mu);
    const size_t T1 = std::max<size_t>(total, 1);
    TC2LI_HIP_CHECK(w.d_probs.ensure(n_frames)); TC2LI_HIP_CHECK(w.d_results.ensure(n_frames)); TC2LI_HIP_CHECK(w.h_results.ensure(n_frames));
    TC2LI_HIP_CHECK(w.d_Xw.ensure(3 * T1)); TC2LI_HIP_CHECK(w.d_chi2.ensure(T1));
    TC2LI_HIP_CHECK(w.d_edges.ensure(T1)); TC2LI_HIP_CHECK(w.d_close.ensure(T1));
    TC2LI_HIP_CHECK(w.d_outlier.ensure(T1));
    static_assert(sizeof(BaEdge) == sizeof(tc2li_ba_edge), "ABI layout");
    // the frames' arrays go up as three copies from one pinned block (a copy per frame and array costs more than the kernel at hundreds of
    // frames); the per-frame host work -- information matrices of the inertial edge, later the marginalisation -- runs on the pool threads
    const size_t o_edges = 3 * T1 * sizeof(double), o_close = o_edges + T1 * sizeof(BaEdge), stage_bytes = o_close + T1;
    TC2LI_HIP_CHECK(w.h_stage.ensure(stage_bytes));
    double* const h_Xw = reinterpret_cast<double*>(w.h_stage.p);
    BaEdge* const h_edges = reinterpret_cast<BaEdge*>(w.h_stage.p + o_edges);
    uint8_t* const h_close = w.h_stage.p + o_close;
    std::atomic<int> bad_frame{-1};
    tracking_pool().parallel_for(n_frames, [&](int f) {
        const tc2li_pose_inertial_problem& p = problems[f];
        PiProblem& d = probs[f];
        const int32_t edge_off = d.edge_off;
        memset(&d, 0, sizeof(d));
        state_from(p.frame, d.cur); state_from(p.other, d.other);
        const tc2li_preintegrated& q = *p.preintegrated;
        d.pre.dT = q.dT;
        memcpy(d.pre.dR, q.dR, 36); memcpy(d.pre.dV, q.dV, 12); memcpy(d.pre.dP, q.dP, 12); memcpy(d.pre.JRg, q.JRg, 36); memcpy(d.pre.JVg, q.JVg, 36);
        memcpy(d.pre.JVa, q.JVa, 36); memcpy(d.pre.JPg, q.JPg, 36); memcpy(d.pre.JPa, q.JPa, 36);
        d.pre.bias[0] = q.bias.bax; d.pre.bias[1] = q.bias.bay; d.pre.bias[2] = q.bias.baz; d.pre.bias[3] = q.bias.bwx; d.pre.bias[4] = q.bias.bwy; d.pre.bias[5] = q.bias.bwz;
        InertialLinkHost li, lw;  // the informations: EdgeInertial's from `preintegrated`, the random-walk ones from `preintegrated_rw` (:2645, :3049)
        li.pre = p.preintegrated; lw.pre = p.preintegrated_rw;
        if (!li.prepare(1.0) || !lw.prepare(1.0)) { bad_frame.store(f); return; }
        memcpy(d.pre.info, li.info, sizeof(li.info)); memcpy(d.pre.infoG, lw.infoG, sizeof(lw.infoG)); memcpy(d.pre.infoA, lw.infoA, sizeof(lw.infoA));
        if (p.last_frame) {
            memcpy(d.prior.Rwb, p.prior->Rwb, 72); memcpy(d.prior.twb, p.prior->twb, 24); memcpy(d.prior.vwb, p.prior->vwb, 24);
            memcpy(d.prior.bg, p.prior->bg, 24); memcpy(d.prior.ba, p.prior->ba, 24); memcpy(d.prior.H, p.prior->H, sizeof(d.prior.H));
        }
        d.edge_off = edge_off; d.n_edges = p.n_edges; d.last_frame = p.last_frame ? 1 : 0; d.rec_init = p.rec_init ? 1 : 0;
        if (p.n_edges) {
            memcpy(h_Xw + 3 * (size_t)edge_off, p.Xw, 3 * (size_t)p.n_edges * sizeof(double));
            memcpy(h_edges + edge_off, p.edges, (size_t)p.n_edges * sizeof(BaEdge));
            memcpy(h_close + edge_off, p.close_point, (size_t)p.n_edges);
        }
    });
    if (bad_frame.load() >= 0) { set_error("frame %d: the pre-integration covariance is not positive definite", bad_frame.load()); return TC2LI_ERR_INVALID; }
    if (total) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_Xw.p, h_Xw, 3 * total * sizeof(double), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_edges.p, h_edges, total * sizeof(BaEdge), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_close.p, h_close, total, hipMemcpyHostToDevice, st));
    }
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_probs.p, probs.data(), n_frames * sizeof(PiProblem), hipMemcpyHostToDevice, st));
    ImuCalib cal;
    CameraD c;
    static_assert(sizeof(ImuCalib) == sizeof(tc2li_imu_calib), "ABI layout");
    memcpy(&cal, calib, sizeof(cal));
    memcpy(&c, cam, sizeof(c));
    launch_pose_inertial(w.d_probs.p, n_frames, w.d_Xw.p, w.d_edges.p, w.d_close.p, cal, c, w.d_outlier.p, w.d_chi2.p, w.d_results.p, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.h_results.p, w.d_results.p, n_frames * sizeof(PiResult), hipMemcpyDeviceToHost, st));
    uint8_t* const h_outlier = w.h_stage.p;  // the uploads above have completed when this copy runs (same stream)
    if (total) TC2LI_HIP_CHECK(hipMemcpyAsync(h_outlier, w.d_outlier.p, total, hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    tracking_pool().parallel_for(n_frames, [&](int f) {
        tc2li_pose_inertial_problem& p = problems[f];
        const PiResult& R = w.h_results.p[f];
        state_to(R.cur, p.frame);
        if (p.last_frame) state_to(R.other, p.other);
        p.n_initial = p.n_edges; p.n_bad = R.n_bad; p.n_inliers = R.n_inliers; p.solver_failed = R.solver_failed;
        if (p.n_edges) memcpy(p.outlier, h_outlier + probs[f].edge_off, (size_t)p.n_edges);
        if (p.prior_out) {
            tc2li_pose_imu_prior& o = *p.prior_out;
            memcpy(o.Rwb, R.cur.P.Rwb, 72); memcpy(o.twb, R.cur.P.twb, 24); memcpy(o.vwb, R.cur.v, 24); memcpy(o.bg, R.cur.bg, 24); memcpy(o.ba, R.cur.ba, 24);
            prior_hessian(probs[f], R, o.H);
        }
        if (results) results[f] = p.n_edges - R.n_bad;
    });
    return n_frames;
}
