// Internal to the bundle-adjustment host code (ba_host.cpp, ba_lockstep.cpp, ba_engine.cpp, lvi_host.cpp): the options, the per-thread and
// per-window work spaces, the visual problem's structure, the inertial term, and what the lock-step drivers share.  Not part of the C ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <pthread.h>
#include "common.hpp"
#include "ba_device.hpp"
#include "balm_host.hpp"
#include "inertial_host.hpp"
#include "reduced_solve.hpp"
#include "pose_opt_device.hpp"

namespace tc2li {
namespace ba_detail {


// The environment switches of the bundle-adjustment entry points, parsed in ONE place at the start of every call (the tests flip some of
// them between two calls of one process) and reported by tc2li_ba_options (bench.py logs it in its detail file).  Round 6 retired the
// switches whose A/B measurements are in DESIGN.md and whose losing form was only kept as a fall-back: TC2LI_BA_SCHUR_LEAN / _MFMA / _GROUP
// (the lean Schur product is the only block-by-block form), TC2LI_BA_FUSE, TC2LI_BA_XP_PINNED, TC2LI_BA_PRE_SCHUR, TC2LI_BA_PHASE_SERIAL,
// TC2LI_BA_DENSE_SLICES.
struct BaOptions {
    bool device_lm = true;        // TC2LI_BA_DEVICE_LM=0: the LM decisions on the host between the phases (rounds 2-5)
    bool device_solve = false;    // TC2LI_BA_DEVICE_SOLVE=1: host-driven loop with the reduced systems solved by k_ba_solve_b
    bool fuse_linearize = false;  // TC2LI_BA_FUSE_LIN=1: the linearisation's closing sums by the window's last workgroup
    bool fuse_trial = false;      // TC2LI_BA_FUSE_TRIAL=1: a trial as one launch over the landmark groups (pb.trial_fused)
    bool lvi_device_solve = true; // TC2LI_LVI_DEVICE_SOLVE=0: the inertial windows' reduced systems on the host's envelope solver
    bool lockstep = true;         // TC2LI_BA_NO_LOCKSTEP: every window through the one-window path
    bool timing = false;          // TC2LI_BA_TIMING: per-call laps on stderr
    int groups = 3;               // TC2LI_BA_LOCKSTEP_GROUPS: lock-step groups of the batch entry points
    std::string shard_fail;       // TC2LI_TEST_SHARD_FAIL: "<rank>:setup" / "<rank>:trial" (tests of the sharded window's failure protocol)
    static BaOptions read() {
        auto flag = [](const char* name, bool dflt) { const char* e = getenv(name); return e ? atoi(e) != 0 : dflt; };
        BaOptions o;
        o.device_lm = flag("TC2LI_BA_DEVICE_LM", true); o.device_solve = flag("TC2LI_BA_DEVICE_SOLVE", false);
        o.fuse_linearize = flag("TC2LI_BA_FUSE_LIN", false); o.fuse_trial = flag("TC2LI_BA_FUSE_TRIAL", false);
        o.lvi_device_solve = flag("TC2LI_LVI_DEVICE_SOLVE", true);
        o.lockstep = getenv("TC2LI_BA_NO_LOCKSTEP") == nullptr; o.timing = getenv("TC2LI_BA_TIMING") != nullptr;
        if (const char* e = getenv("TC2LI_BA_LOCKSTEP_GROUPS")) o.groups = atoi(e);
        o.groups = std::max(1, std::min(kMaxLockstepGroups, o.groups));
        if (const char* e = getenv("TC2LI_TEST_SHARD_FAIL")) o.shard_fail = e;
        return o;
    }
};

// result of a window a lock-step group hands back to the one-window path (never seen by a caller: the batch entry points run that path at once)
constexpr int kLockstepDeclined = -1000000;

struct PoseOptWorkspace {
    DevBuf<PoseProblem> d_probs;
    DevBuf<double> d_Xw, d_poses, d_chi2;
    DevBuf<BaEdge> d_edges;
    DevBuf<uint8_t> d_outlier;
    DevBuf<int> d_inliers;
    std::mutex mu;
};
inline PoseOptWorkspace& po_ws() { static thread_local PoseOptWorkspace w; return w; }

// The inertial reduced system on its way to the device solve (k_lvi_solve*, ba_kernels.hip): one blob per window and linearisation --
// [first n | rowoff n + 1] ints, then [bi n | the envelope's entries] doubles -- and the kernel's scratch.
struct LviSolveBuffers {
    DevBuf<uint8_t> d_blob;
    PinnedBuf<uint8_t> h_blob;
    DevBuf<double> d_LB, d_Lband;
    size_t ints_bytes = 0;
    LviSolveDev dev{};
    hipError_t ensure(int np, int ni) {
        const size_t n = (size_t)np + ni;
        ints_bytes = ((n + 2 * (size_t)np + 1) * sizeof(int32_t) + 15) / 16 * 16;
        const size_t max_pose = (size_t)np * ni + (size_t)np * (np + 1) / 2;  // pose rows against the band (worst case: every column), pose block
        const size_t bytes = ints_bytes + (n + (size_t)ni * 32 + max_pose) * sizeof(double);
        hipError_t e;
        if ((e = d_blob.ensure(bytes)) != hipSuccess || (e = h_blob.ensure(bytes)) != hipSuccess || (e = d_LB.ensure(std::max<size_t>((size_t)ni * np, 1))) != hipSuccess ||
            (e = d_Lband.ensure(std::max<size_t>((size_t)ni * 32, 1))) != hipSuccess) return e;
        dev.n = (int32_t)n; dev.np = np; dev.ni = ni; dev.pad_ = 0;
        dev.first = reinterpret_cast<const int32_t*>(d_blob.p);
        dev.span_end = dev.first + n;
        dev.rowoff = dev.span_end + np;
        dev.bi = reinterpret_cast<const double*>(d_blob.p + ints_bytes);
        dev.hband = dev.bi + n;
        dev.hpose = dev.hband + (size_t)ni * 32;
        dev.LB = d_LB.p; dev.Lband = d_Lband.p;
        return hipSuccess;
    }
    // after ReducedSolver::set_pattern: the blob of this linearisation; returns the bytes to copy (h_blob -> d_blob)
    size_t pack(const ReducedSolver& rs, const double* Hi, const double* bi) {
        const size_t n = (size_t)rs.n;
        int32_t* ints = reinterpret_cast<int32_t*>(h_blob.p);
        double* dbl = reinterpret_cast<double*>(h_blob.p + ints_bytes);
        memcpy(ints, rs.first.data(), n * sizeof(int32_t));
        memcpy(dbl, bi, n * sizeof(double));
        std::vector<int32_t> span_first(std::max(rs.np, 1));  // (= first[ni + r]: already in the blob)
        const size_t entries = rs.pack_for_device(Hi, span_first.data(), ints + n, ints + n + rs.np, dbl + n, dbl + n + (size_t)rs.ni * 32);
        return ints_bytes + (n + (size_t)rs.ni * 32 + entries) * sizeof(double);
    }
};
struct BaWorkspace {
    LviSolveBuffers lvi;
    DevBuf<Se3> d_poses, d_poses_trial;  // d_poses: tc2li_lidar_window_evaluate only; a window's poses live in d_in
    DevBuf<double> d_points_trial, d_chi2, d_rho0, d_cp, d_W, d_Hll, d_bl, d_diag_l, d_Hpp, d_diag_p,
        d_coef_e, d_coef, d_Y, d_Spart, d_scale_part, d_chi_part, d_red;
    // the window as the caller hands it over -- poses, points, edges and the index arrays -- goes up in ONE copy: a stream operation
    // costs about as much as one of the loop's kernels, and a batch has one such set per window
    DevBuf<uint8_t> d_in;
    PinnedBuf<uint8_t> h_in;
    DevBuf<uint8_t> d_depth;
    PinnedBuf<double> h_S, h_bs, h_xp, h_scal, h_Hpp, h_stat;
    DevBuf<ImuPose> d_iposes, d_iposes_trial;
    PinnedBuf<ImuPose> h_iposes, h_iposes_up;  // trial states on their way back; the initial states on their way up (lock-step batch)
    PinnedBuf<uint8_t> h_result;  // lock-step batch: poses, points, per-edge chi2 and depth flags on their way to the caller
    // lock-step batch with the reduced system solved on the device: S, [b_s | b_p], the step; the LiDAR term's Hessian | gradient on both sides
    DevBuf<double> d_S, d_bs, d_xp, d_Hl;
    // device-side LM (round 6): the kernels' scalar sums, the LiDAR term's output record and its camera-se3 Jacobian / Hessian stay in device memory
    DevBuf<double> d_scal, d_balm_out, d_lidar_JH;
    PinnedBuf<double> h_Hl;
    PinnedBuf<int32_t> h_ok;
    BalmTerm lidar;
    std::mutex mu;
};
// one workspace per host thread: windows optimised from different threads (tc2li_local_bundle_adjustment_batch) do not
// share device buffers
inline BaWorkspace& ba_ws() { static thread_local BaWorkspace w; return w; }

// Structure and device state of the projection-edge part of a local BA (shared by the visual / LiDAR and the inertial
// entry points): free-pose numbering, CSR of the edges by landmark and by free pose, workspace sizing, uploads, and the
// BaProblemDev handed to the kernels.  poses7 == NULL: the caller uploads ImuPose states itself (inertial mode).
struct VisualProblem {
    BaProblemDev pb{};
    std::vector<int> pose_var;
    std::vector<Se3> poses;
    int n_free = 0, np = 0, n_slices = 1, k_per_slice = 4;
    int max_group_landmarks = 0;
    // pb.trial_fused (ba_device.hpp): whether the window's trials run as the one fused launch -- a property of the window (it fixes the order
    // of two sums), decided here and again by a caller that switches the vertices to ImuCamPose records.  OFF unless TC2LI_BA_FUSE_TRIAL=1:
    // built for VERDICT r4 item 2 ("a trial <= 3 launches"), parity-green in both forms (tests/test_ba_gpu.py, test_balm_gpu.py,
    // test_inertial_ba_gpu.py run whichever the environment selects) and measured SLOWER in the whole loop -- 27.1 / 27.3 ms per step against
    // 26.1 / 25.8 in two A/B pairs of one call, the camera threads 26.6-27.3 against 25.4-26.1; local BA alone 13.0-13.7 against 12.1-13.9 ms
    // per 128 windows.  A workgroup of the fused launch runs six dependent trips to memory (step, poses, slots, W blocks, edge list, edges)
    // while it holds 28 KB of LDS; the three launches it replaces are thin kernels of two or three trips each that start and finish quickly
    // beside the other stages' wavefronts.  In a loop bound by the kernels' combined occupancy, fewer launches is not the lever; shorter
    // residency is.
    void decide_trial_fused() {
        const size_t pose_bytes = (size_t)pb.n_poses * (pb.inertial ? sizeof(ImuPose) : sizeof(Se3));
        pb.trial_fused = BaOptions::read().fuse_trial && np <= kBacksubMaxNp && pose_bytes <= (size_t)kTrialPoseBytes && max_group_landmarks <= 256 ? 1 : 0;
    }

    int setup(BaWorkspace& ws, const double* poses7, const uint8_t* fixed, int n_poses, const double* points3, int n_points,
              const tc2li_ba_edge* edges, int n_edges, const tc2li_camera* cam, const uint8_t* extra_used, hipStream_t st) {
    // ---- structure: free-pose numbering, CSR by landmark and by free pose ----
    pose_var.assign(n_poses, -1);
    n_free = 0;
    std::vector<uint8_t> used(n_poses, 0);
    for (int e = 0; e < n_edges; ++e) {
        if (edges[e].pose < 0 || edges[e].pose >= n_poses || edges[e].point < 0 || edges[e].point >= n_points) {
            set_error("edge %d references pose %d / point %d out of range", e, edges[e].pose, edges[e].point);
            return TC2LI_ERR_INVALID;
        }
        used[edges[e].pose] = 1;
    }
    for (int k = 0; k < n_poses; ++k) if (extra_used && extra_used[k]) used[k] = 1;
    for (int k = 0; k < n_poses; ++k) if (!fixed[k] && used[k]) pose_var[k] = n_free++;
    std::vector<int> pt_off(n_points + 1, 0), pt_edges(n_edges), pv_off(n_free + 1, 0);
    for (int e = 0; e < n_edges; ++e) { pt_off[edges[e].point + 1]++; if (pose_var[edges[e].pose] >= 0) pv_off[pose_var[edges[e].pose] + 1]++; }
    for (int l = 0; l < n_points; ++l) {
        if (pt_off[l + 1] == 0) { set_error("point %d has no edge", l); return TC2LI_ERR_INVALID; }
        pt_off[l + 1] += pt_off[l];
    }
    for (int i = 0; i < n_free; ++i) pv_off[i + 1] += pv_off[i];
    int n_free_edges = pv_off[n_free];  // edges with a free pose; after the slots are made: the SLOTS (duplicates of a (point, pose) pair have none)
    std::vector<int> pv_edges(std::max(n_free_edges, 1));
    {
        std::vector<int> fl(pt_off.begin(), pt_off.end() - 1), fp(pv_off.begin(), pv_off.end() - 1);
        for (int e = 0; e < n_edges; ++e) {
            pt_edges[fl[edges[e].point]++] = e;
            const int i = pose_var[edges[e].pose];
            if (i >= 0) pv_edges[fp[i]++] = e;
        }
    }
    // the edges with a free pose in landmark-major order: where the W blocks live (the Schur product and the back substitution walk
    // them by landmark)
    // fl_off: per landmark [begin, end) of its slots, the landmarks in index order.  (Tried: slots in the order of the poses a landmark
    // is seen from, so that a chunk of the Schur kernel spans a narrow band of poses and the product's empty tiles can be skipped -- the
    // windows' covisibility is not banded enough for that, and the linearisation lost its locality: 64 -> 98 us.)
    // Every window of at most kSchurLeanMaxFree (24) free keyframes runs the lean block-by-block Schur product (ba_device.hpp) -- up to
    // kSchurBlocksMaxFree (21) with one workgroup per part, above with two (schur_ranges_wide); wider windows the block-sparse MFMA kernels.
    const bool lean_wide = n_free > kSchurBlocksMaxFree && n_free <= kSchurLeanMaxFree;
    const bool schur_lean = (6 * n_free + 1 + 15) / 16 <= 8 || lean_wide;
    struct DupEdge { int pose, edge, slot; };
    std::vector<DupEdge> dups;
    std::vector<int> fl_off(2 * (size_t)n_points, 0), fl_pose(std::max(n_free_edges, 1)), fl_lm(std::max(n_free_edges, 1)), fl_place(std::max(n_free_edges, 1)),
        fl_edge(std::max(n_free_edges, 1)), w_slot(n_edges, -1), slice_off(1, 0);
    {
        // slices of the sparse Schur kernel: whole landmarks, at most 256 edges (one per thread) of at most 64 landmarks; a function of
        // the window alone, so that a window gives the same bits alone and in a batch
        // (the lean form of the block-by-block product stages half as many slots at a time: kSchurLeanSlots)
        // Dense windows (more than 21 free keyframes -- the temporal window of LocalInertialBA's bLarge case; round 5, d_ba_schur_units): the
        // slots follow the landmarks sorted by the first and the last free pose that sees them, and a slice is a CHUNK of 16 landmarks -- a
        // landmark of a temporal window is seen from a run of consecutive keyframes, so a chunk touches a band of the reduced system and the
        // product skips the rest.  (The covisibility windows of the sparse path are not banded: see above.)
        const bool dense_window = (6 * n_free + 1 + 15) / 16 > 8 && !lean_wide;
        const int kSliceEdges = dense_window ? std::numeric_limits<int>::max() : kSchurLeanSlots;
        const int kSliceLandmarks = dense_window ? kUnitChunkHost : 64;
        std::vector<int> order(n_points);
        for (int l = 0; l < n_points; ++l) order[l] = l;
        if (dense_window) {
            std::vector<int> first(n_points, std::numeric_limits<int>::max()), last(n_points, -1);
            for (int e = 0; e < n_edges; ++e) {
                const int i = pose_var[edges[e].pose], l = edges[e].point;
                if (i >= 0) { first[l] = std::min(first[l], i); last[l] = std::max(last[l], i); }
            }
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] != first[b] ? first[a] < first[b] : last[a] < last[b]; });
        }
        std::vector<int> seen(std::max(n_free, 1), -1), seen_slot(std::max(n_free, 1), -1);
        int at = 0, slice_lms = 0;
        for (int lo = 0; lo < n_points; ++lo) {
            const int l = order[lo];
            const int begin = at;
            for (int k = pt_off[l]; k < pt_off[l + 1]; ++k) {
                const int e = pt_edges[k], i = pose_var[edges[e].pose];
                if (i < 0) continue;
                // A second edge between the same point and the same free pose: g2o adds the two edges' blocks (BaseBinaryEdge::
                // constructQuadraticForm on the same Hpl / Hpp blocks, base_binary_edge.hpp:55-137).  The slot arrays hold one W block per
                // (landmark, pose): the later edge gets no slot -- k_ba_dups adds its W block to the first edge's slot and its pose block to
                // the pose's sums after the linearisation (round 5; rounds 2-4 refused such a window).  The reference's gather cannot produce
                // one for a pinhole rig, a two-camera shim can.
                if (seen[i] == l) { dups.push_back(DupEdge{i, e, seen_slot[i]}); continue; }
                seen[i] = l; seen_slot[i] = at;
                w_slot[e] = at; fl_pose[at] = i; fl_lm[at] = l; fl_edge[at] = e; ++at;
            }
            fl_off[2 * (size_t)l] = begin; fl_off[2 * (size_t)l + 1] = at;
            if (at == begin) continue;
            if (slice_lms == kSliceLandmarks || at - slice_off.back() > kSliceEdges) { slice_off.push_back(begin); slice_lms = 0; }
            for (int k = begin; k < at; ++k) fl_place[k] = slice_lms;
            ++slice_lms;
        }
        if (at > slice_off.back()) slice_off.push_back(at);
        n_free_edges = at;
    }
    // duplicates (k_ba_dups): by pose, in edge order; the per-pose edge lists of the dense windows' coefficient sums hold the slots' edges only
    std::vector<int> dup_off(n_free + 1, 0), dup_edge(std::max(dups.size(), (size_t)1)), dup_slot(std::max(dups.size(), (size_t)1));
    if (!dups.empty()) {
        std::stable_sort(dups.begin(), dups.end(), [](const DupEdge& a, const DupEdge& b) { return a.pose != b.pose ? a.pose < b.pose : a.edge < b.edge; });
        for (size_t k = 0; k < dups.size(); ++k) { dup_off[dups[k].pose + 1]++; dup_edge[k] = dups[k].edge; dup_slot[k] = dups[k].slot; }
        for (int i = 0; i < n_free; ++i) dup_off[i + 1] += dup_off[i];
        std::fill(pv_off.begin(), pv_off.end(), 0);
        for (int e = 0; e < n_edges; ++e) if (w_slot[e] >= 0) pv_off[pose_var[edges[e].pose] + 1]++;
        for (int i = 0; i < n_free; ++i) pv_off[i + 1] += pv_off[i];
        std::vector<int> fp(pv_off.begin(), pv_off.end() - 1);
        for (int e = 0; e < n_edges; ++e) if (w_slot[e] >= 0) pv_edges[fp[pose_var[edges[e].pose]]++] = e;
    }
    // blocks of 256 free-pose edges (the pose role of the linearisation): the block's rows sorted by pose, for the per-pose sums
    const int n_blocks = (n_free_edges + 255) / 256;
    std::vector<int> blk_off((size_t)std::max(n_blocks, 1) * (n_free + 1), 0);
    std::vector<uint8_t> blk_rows((size_t)std::max(n_blocks, 1) * 256, 0);
    for (int b = 0; b < n_blocks; ++b) {
        int* off = blk_off.data() + (size_t)b * (n_free + 1);
        const int s0 = 256 * b, s1 = std::min(n_free_edges, s0 + 256);
        for (int s = s0; s < s1; ++s) off[fl_pose[s] + 1]++;
        for (int i = 0; i < n_free; ++i) off[i + 1] += off[i];
        std::vector<int> fill(off, off + n_free);
        for (int s = s0; s < s1; ++s) blk_rows[(size_t)b * 256 + fill[fl_pose[s]]++] = (uint8_t)(s - s0);
    }
    // groups of the linearisation: whole landmarks, at most 256 edges (one per thread)
    std::vector<int> grp_k0(1, 0), grp_l0(1, 0);
    for (int l = 0; l < n_points; ++l) {
        if (pt_off[l + 1] - pt_off[l] > 256) { set_error("point %d has more than 256 edges", l); return TC2LI_ERR_INVALID; }
        if (pt_off[l + 1] - grp_k0.back() > 256) { grp_k0.push_back(pt_off[l]); grp_l0.push_back(l); }
    }
    grp_k0.push_back(n_edges); grp_l0.push_back(n_points);
    const int n_groups = (int)grp_k0.size() - 1;
    max_group_landmarks = 0;
    for (int g = 0; g < n_groups; ++g) max_group_landmarks = std::max(max_group_landmarks, grp_l0[g + 1] - grp_l0[g]);
    if (max_group_landmarks > 256) { set_error("more than 256 landmarks without edges in a row"); return TC2LI_ERR_INVALID; }  // (a landmark-role workgroup has a thread per landmark)
    np = 6 * n_free;
    // sparse path: one spare row for W D^-1 b_l (row np of the product); dense path: the operands' width
    const bool sparse = schur_lean;
    const int np_pad = sparse ? (np + 1 + 15) / 16 * 16 : std::max(16, (np + 15) / 16 * 16);
    const int n_schur_slices = (int)slice_off.size() - 1;
    int schur_group = 1;
    if (sparse) {
        schur_group = kSchurGroupLean;  // slices per part
        n_slices = ba_schur_parts(n_schur_slices, schur_group);  // partial sums in S_part
        k_per_slice = 0;
    } else {
        // dense windows (round 5: d_ba_schur_units): the chunks (slices of slice_off: 16 landmarks each) in at most 8 ranges = partial sums
        const int want_slices = 8;  // (full-width form, 32 windows per launch beside two other groups: 2 / 4 / 8 slices 0.263 / 0.154 / 0.099 ms)
        k_per_slice = std::min(64, std::max(1, (n_schur_slices + want_slices - 1) / want_slices));   // chunks per partial sum (at most kUnitMaxChunks: ba_kernels.hip)
        n_slices = std::max(1, (n_schur_slices + k_per_slice - 1) / k_per_slice);
    }
    // which 16-column tiles of the reduced system a chunk of landmarks touches (bit t: a pose with columns in tile t sees one of them)
    std::vector<uint32_t> chunk_mask;
    if (!sparse) {
        if (np_pad / 16 > 32) { set_error("more than 85 free keyframes"); return TC2LI_ERR_INVALID; }
        chunk_mask.assign((size_t)std::max(n_schur_slices, 1), 0u);
        for (int c = 0; c < n_schur_slices; ++c)
            for (int sl = slice_off[c]; sl < slice_off[c + 1]; ++sl) {
                const int c0 = 6 * fl_pose[sl];
                chunk_mask[c] |= (1u << (c0 / 16)) | (1u << ((c0 + 5) / 16));
            }
    }

    // ---- device memory: a per-thread workspace that only grows (hipMalloc per call would dominate the run time) ----
    auto& d_poses_trial = ws.d_poses_trial;
    auto &d_points_trial = ws.d_points_trial, &d_chi2 = ws.d_chi2, &d_rho0 = ws.d_rho0,
         &d_cp = ws.d_cp, &d_W = ws.d_W, &d_Hll = ws.d_Hll, &d_bl = ws.d_bl, &d_diag_l = ws.d_diag_l, &d_Hpp = ws.d_Hpp,
         &d_diag_p = ws.d_diag_p, &d_coef_e = ws.d_coef_e, &d_coef = ws.d_coef,
         &d_Spart = ws.d_Spart, &d_scale_part = ws.d_scale_part, &d_chi_part = ws.d_chi_part;
    auto& d_depth = ws.d_depth;
    auto &h_S = ws.h_S, &h_bs = ws.h_bs, &h_xp = ws.h_xp, &h_scal = ws.h_scal;
    const size_t E = n_edges, P = n_points;
    TC2LI_HIP_CHECK(d_poses_trial.ensure(n_poses));
    TC2LI_HIP_CHECK(d_points_trial.ensure(3 * P));
    TC2LI_HIP_CHECK(d_chi2.ensure(E)); TC2LI_HIP_CHECK(d_rho0.ensure(E)); TC2LI_HIP_CHECK(d_cp.ensure(kContribP * (size_t)std::max(n_blocks * n_free, 1)));
    TC2LI_HIP_CHECK(d_W.ensure(18 * (size_t)std::max(n_free_edges, 1))); TC2LI_HIP_CHECK(d_Hll.ensure(6 * P)); TC2LI_HIP_CHECK(d_bl.ensure(3 * P)); TC2LI_HIP_CHECK(d_diag_l.ensure(P));
    TC2LI_HIP_CHECK(d_Hpp.ensure(27 * (size_t)std::max(n_free, 1))); TC2LI_HIP_CHECK(d_diag_p.ensure(std::max(n_free, 1)));
    if (!sparse) { TC2LI_HIP_CHECK(d_coef_e.ensure(6 * E)); TC2LI_HIP_CHECK(ws.d_Y.ensure(18 * (size_t)std::max(n_free_edges, 1))); }
    TC2LI_HIP_CHECK(d_coef.ensure(6 * (size_t)std::max(n_free, 1)));
    TC2LI_HIP_CHECK(d_Spart.ensure((size_t)std::max(n_slices, 1) * np_pad * np_pad)); TC2LI_HIP_CHECK(d_scale_part.ensure(P / 256 + 1)); TC2LI_HIP_CHECK(d_chi_part.ensure(std::max(E / 256 + 1, (size_t)n_groups)));
    TC2LI_HIP_CHECK(d_depth.ensure(E));
    TC2LI_HIP_CHECK(h_S.ensure((size_t)std::max(np * np, 1))); TC2LI_HIP_CHECK(h_bs.ensure(2 * (size_t)std::max(np, 1)));
    TC2LI_HIP_CHECK(h_xp.ensure(std::max(np, 1))); TC2LI_HIP_CHECK(h_scal.ensure(8));
    memset(h_S.p, 0, (size_t)std::max(np * np, 1) * sizeof(double));  // the finish kernel writes the lower triangle only; the rest stays defined
    // ---- the input block: [poses | points | edges | pose_var | pt_off | pt_edges | pv_off | pv_edges | fl_off | fl_pose | chunk_mask | fl_lm | fl_place | slice_off | fl_edge | grp_k0 | grp_l0 | blk_off | blk_rows | ticket words], every
    // part 16-byte aligned ----
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_poses = 0, o_points = align16(o_poses + n_poses * sizeof(Se3)), o_edges = align16(o_points + 3 * P * sizeof(double)),
                 o_pose_var = align16(o_edges + E * sizeof(BaEdge)), o_pt_off = align16(o_pose_var + n_poses * sizeof(int)),
                 o_pt_edges = align16(o_pt_off + (P + 1) * sizeof(int)), o_pv_off = align16(o_pt_edges + E * sizeof(int)),
                 o_pv_edges = align16(o_pv_off + (n_free + 1) * sizeof(int)), o_fl_off = align16(o_pv_edges + (sparse ? 0 : pv_edges.size()) * sizeof(int)),
                 o_fl_pose = align16(o_fl_off + 2 * P * sizeof(int)), o_w_slot = align16(o_fl_pose + fl_pose.size() * sizeof(int)),
                 o_fl_lm = align16(o_w_slot + chunk_mask.size() * sizeof(uint32_t)), o_fl_place = align16(o_fl_lm + fl_lm.size() * sizeof(int)),
                 o_slice_off = align16(o_fl_place + fl_place.size() * sizeof(int)), o_fl_edge = align16(o_slice_off + slice_off.size() * sizeof(int)),
                 o_grp_k0 = align16(o_fl_edge + fl_edge.size() * sizeof(int)), o_grp_l0 = align16(o_grp_k0 + grp_k0.size() * sizeof(int)),
                 o_blk_off = align16(o_grp_l0 + grp_l0.size() * sizeof(int)), o_blk_rows = align16(o_blk_off + blk_off.size() * sizeof(int)),
                 o_ticket = align16(o_blk_rows + blk_rows.size()), o_dup_off = align16(o_ticket + 4 * sizeof(int32_t)),
                 o_dup_edge = align16(o_dup_off + (dups.empty() ? 0 : dup_off.size()) * sizeof(int)),
                 o_dup_slot = align16(o_dup_edge + (dups.empty() ? 0 : dups.size()) * sizeof(int)),
                 in_bytes = align16(o_dup_slot + (dups.empty() ? 0 : dups.size()) * sizeof(int));
    TC2LI_HIP_CHECK(ws.d_in.ensure(in_bytes)); TC2LI_HIP_CHECK(ws.h_in.ensure(in_bytes));
    uint8_t* const h = ws.h_in.p;
    if (poses7) {
        poses.resize(n_poses);
        for (int k = 0; k < n_poses; ++k) { memcpy(poses[k].q, poses7 + 7 * k, 4 * sizeof(double)); memcpy(poses[k].t, poses7 + 7 * k + 4, 3 * sizeof(double)); }
        memcpy(h + o_poses, poses.data(), n_poses * sizeof(Se3));
    }
    memcpy(h + o_points, points3, 3 * P * sizeof(double));
    memcpy(h + o_edges, edges, E * sizeof(BaEdge));
    memcpy(h + o_pose_var, pose_var.data(), n_poses * sizeof(int));
    memcpy(h + o_pt_off, pt_off.data(), (P + 1) * sizeof(int));
    memcpy(h + o_pt_edges, pt_edges.data(), E * sizeof(int));
    memcpy(h + o_pv_off, pv_off.data(), (n_free + 1) * sizeof(int));
    if (!sparse) memcpy(h + o_pv_edges, pv_edges.data(), pv_edges.size() * sizeof(int));  // pv_edges, w_slot: the dense Schur path's
    memcpy(h + o_fl_off, fl_off.data(), 2 * P * sizeof(int));
    memcpy(h + o_fl_pose, fl_pose.data(), fl_pose.size() * sizeof(int));
    if (!sparse) memcpy(h + o_w_slot, chunk_mask.data(), chunk_mask.size() * sizeof(uint32_t));  // (the region held w_slot for the dense form's prepare kernel)
    memcpy(h + o_fl_lm, fl_lm.data(), fl_lm.size() * sizeof(int));
    memcpy(h + o_fl_place, fl_place.data(), fl_place.size() * sizeof(int));
    memcpy(h + o_slice_off, slice_off.data(), slice_off.size() * sizeof(int));
    memcpy(h + o_fl_edge, fl_edge.data(), fl_edge.size() * sizeof(int));
    memcpy(h + o_grp_k0, grp_k0.data(), grp_k0.size() * sizeof(int));
    memcpy(h + o_grp_l0, grp_l0.data(), grp_l0.size() * sizeof(int));
    memcpy(h + o_blk_off, blk_off.data(), blk_off.size() * sizeof(int));
    memcpy(h + o_blk_rows, blk_rows.data(), blk_rows.size());
    memset(h + o_ticket, 0, 4 * sizeof(int32_t));  // (the kernels that use them leave them at zero again)
    if (!dups.empty()) {
        memcpy(h + o_dup_off, dup_off.data(), dup_off.size() * sizeof(int));
        memcpy(h + o_dup_edge, dup_edge.data(), dups.size() * sizeof(int));
        memcpy(h + o_dup_slot, dup_slot.data(), dups.size() * sizeof(int));
    }
    // inertial mode (poses7 == NULL) uploads ImuPose states itself and does not read the Se3 block
    const size_t first = poses7 ? 0 : o_points;
    TC2LI_HIP_CHECK(upload_or_defer(ws.d_in.p + first, h + first, in_bytes - first, st));  // h is pinned
    uint8_t* const d = ws.d_in.p;

    pb = BaProblemDev{};
    pb.n_edges = n_edges; pb.n_points = n_points; pb.n_poses = n_poses; pb.n_free = n_free; pb.n_free_edges = n_free_edges; pb.np_pad = np_pad;
    memcpy(&pb.cam, cam, sizeof(CameraD));
    const float dm = sqrtf(5.991f), ds = sqrtf(7.815f);  // thHuberMono / thHuberStereo are floats (OptimizerWithLidar.cc:219-220)
    pb.delta_mono = dm; pb.delta_stereo = ds;
    pb.dsqr_mono = (float)((double)dm * (double)dm); pb.dsqr_stereo = (float)((double)ds * (double)ds);
    pb.poses = (Se3*)(d + o_poses); pb.poses_trial = d_poses_trial.p; pb.points = (double*)(d + o_points); pb.points_trial = d_points_trial.p;
    pb.edges = (const BaEdge*)(d + o_edges); pb.pose_var = (const int*)(d + o_pose_var); pb.pt_off = (const int*)(d + o_pt_off);
    pb.pt_edges = (const int*)(d + o_pt_edges); pb.pv_off = (const int*)(d + o_pv_off); pb.pv_edges = (const int*)(d + o_pv_edges);
    pb.fl_off = (const int*)(d + o_fl_off); pb.fl_pose = (const int*)(d + o_fl_pose); pb.chunk_mask = (const uint32_t*)(d + o_w_slot);
    pb.fl_lm = (const int*)(d + o_fl_lm); pb.fl_place = (const int*)(d + o_fl_place); pb.slice_off = (const int*)(d + o_slice_off); pb.fl_edge = (const int*)(d + o_fl_edge);
    pb.grp_k0 = (const int*)(d + o_grp_k0); pb.grp_l0 = (const int*)(d + o_grp_l0); pb.n_groups = n_groups;
    pb.blk_off = (const int*)(d + o_blk_off); pb.blk_rows = (const uint8_t*)(d + o_blk_rows);
    pb.ticket = (int32_t*)(d + o_ticket);
    pb.n_dups = (int32_t)dups.size();
    pb.dup_off = dups.empty() ? nullptr : (const int*)(d + o_dup_off);
    pb.dup_edge = dups.empty() ? nullptr : (const int*)(d + o_dup_edge);
    pb.dup_slot = dups.empty() ? nullptr : (const int*)(d + o_dup_slot);
    pb.sparse_schur = sparse ? 1 : 0; pb.schur_blocks = sparse ? 2 : 0; pb.schur_group = schur_group; pb.n_schur_slices = n_schur_slices;  // (dense windows: the chunks of d_ba_schur_units)
    pb.schur_rd = pb.schur_ro = 1;
    decide_trial_fused();
    if (sparse) {
        if (lean_wide) schur_ranges_wide(n_free, pb.schur_rd, pb.schur_ro); else schur_ranges(n_free, pb.schur_rd, pb.schur_ro);
    }
    pb.chi2 = d_chi2.p; pb.rho0 = d_rho0.p; pb.cp_part = d_cp.p; pb.W = d_W.p; pb.Hll = d_Hll.p; pb.bl = d_bl.p;
    pb.diag_l = d_diag_l.p; pb.Hpp = d_Hpp.p; pb.diag_p = d_diag_p.p; pb.coef_e = d_coef_e.p; pb.coef = d_coef.p; pb.Y = sparse ? nullptr : ws.d_Y.p;
    pb.S_part = d_Spart.p; pb.scale_part = d_scale_part.p; pb.chi_part = d_chi_part.p;

        return TC2LI_OK;
    }
};


// The inertial edges of a window (EdgeInertial + EdgeGyroRW + EdgeAccRW per link, SF/src/OptimizerWithLidar.cc:729-800) on the host:
// their robust cost at a state and, when asked, their dense normal equations in the numbering [6 per free pose | 9 per free
// keyframe with IMU state].  Shared by the one-window entry point and the lock-step batch.
struct InertialTerm {
    std::vector<InertialLinkHost> L;
    std::vector<int> imu_var;
    const std::vector<int>* pose_var = nullptr;
    int np = 0, n = 0, n_imu = 0;
    std::vector<double> Hi, bi;
    // The row segments of Hi a linearisation writes (the blocks of the inertial edges; the caller adds those of the LiDAR term): from the second
    // linearisation on only they are cleared -- Hi is n x n (1.1 MB at 375 unknowns) and almost empty, and clearing it whole was half of what
    // the inertial edges of a window cost the host.
    std::vector<uint32_t> seg_at;
    std::vector<uint8_t> seg_len;
    bool segs_ready = false;
    void note_segment(size_t at, int len) {  // (a length is a byte: longer runs are recorded in pieces)
        if (segs_ready) return;
        for (; len > 0; at += 255, len -= 255) { seg_at.push_back((uint32_t)at); seg_len.push_back((uint8_t)std::min(len, 255)); }
    }
    void note_block(int row0, int col0, int rows, int cols) { for (int r = 0; r < rows; ++r) note_segment((size_t)(row0 + r) * n + col0, cols); }
    double d_imu = 0;
    float dsqr_imu = 0;

    // links -> L; extra_used / imu_used [n_kfs]: keyframes an inertial edge touches
    int prepare(const tc2li_inertial_link* links, int n_links, const uint8_t* has_imu, int n_kfs, std::vector<uint8_t>& extra_used) {
        L.resize(n_links);
        extra_used.assign(n_kfs, 0);
        for (int l = 0; l < n_links; ++l) {
            const tc2li_inertial_link& in = links[l];
            if (in.kf1 < 0 || in.kf1 >= n_kfs || in.kf2 < 0 || in.kf2 >= n_kfs || !in.preintegrated) { set_error("inertial link %d: invalid keyframe index or null pre-integration", l); return TC2LI_ERR_INVALID; }
            if (!has_imu[in.kf1] || !has_imu[in.kf2]) { set_error("inertial link %d joins a keyframe without IMU state", l); return TC2LI_ERR_INVALID; }
            L[l].kf1 = in.kf1; L[l].kf2 = in.kf2; L[l].robust = in.robust != 0; L[l].pre = in.preintegrated;
            if (!L[l].prepare(in.info_scale)) { set_error("inertial link %d: the pre-integration covariance is not positive definite", l); return TC2LI_ERR_INVALID; }
            extra_used[in.kf1] = extra_used[in.kf2] = 1;
        }
        const float d_imu_f = sqrtf(16.92f);
        d_imu = d_imu_f;
        dsqr_imu = (float)((double)d_imu_f * (double)d_imu_f);
        return TC2LI_OK;
    }
    // imu_used: keyframes whose velocity / bias vertices an inertial edge touches (extra_used before the LiDAR window was added)
    void number(const uint8_t* fixed, const uint8_t* has_imu, const std::vector<uint8_t>& imu_used, int n_kfs, const std::vector<int>& pose_var_, int np_) {
        pose_var = &pose_var_; np = np_;
        imu_var.assign(n_kfs, -1);
        n_imu = 0;
        for (int k = 0; k < n_kfs; ++k) if (!fixed[k] && has_imu[k] && imu_used[k]) imu_var[k] = n_imu++;
        n = np + 9 * n_imu;
        Hi.assign((size_t)n * n, 0.0); bi.assign(n, 0.0);
        seg_at.clear(); seg_len.clear(); segs_ready = false;
    }
    // whether k_lvi_solve* takes this window's reduced system: velocity / bias unknowns present, the pose block and the rings fit a CU's LDS
    // (25 free keyframes: the reference's largest window), every inertial edge joins keyframes at most two places apart in the numbering
    // (band <= kLviBand).  The decision depends on the window alone: the same alone and in a batch.  TC2LI_LVI_DEVICE_SOLVE=0: the host's
    // envelope LDL^T (reduced_solve.hpp) for every window.
    bool device_solve_ok() const {
        if (!BaOptions::read().lvi_device_solve || n_imu <= 0 || np <= 0 || np > kLviMaxPoseRows || !lvi_device_solve_available()) return false;
        for (const InertialLinkHost& lk_ : L) {
            const int i1 = imu_var[lk_.kf1], i2 = imu_var[lk_.kf2];
            if (i1 >= 0 && i2 >= 0 && std::abs(i1 - i2) > 2) return false;
        }
        return true;
    }
    double cost(const std::vector<ImuPose>& Pz, const std::vector<ImuVertexState>& Sz, bool linearize) {
        const std::vector<int>& pv = *pose_var;
        double chi = 0;
        if (linearize) {
            if (segs_ready) {
                for (size_t k = 0; k < seg_at.size(); ++k) std::fill_n(Hi.data() + seg_at[k], seg_len[k], 0.0);
                // TC2LI_TEST_HI_CLEAR (tests): every writer of Hi must have registered its blocks during the first linearisation -- after the
                // segment-wise clear the matrix has to be zero everywhere, or a term added later is accumulating stale entries
                static const bool kCheck = getenv("TC2LI_TEST_HI_CLEAR") != nullptr;
                if (kCheck) for (double v : Hi) if (v != 0.0) { fprintf(stderr, "tc2li: InertialTerm: Hi holds an entry outside the registered segments\n"); abort(); }
            } else std::fill(Hi.begin(), Hi.end(), 0.0);
            std::fill(bi.begin(), bi.end(), 0.0);
        }
        for (const InertialLinkHost& lk_ : L) {
            double er[9], J[9 * 24];
            lk_.evaluate(Pz[lk_.kf1], Sz[lk_.kf1], Pz[lk_.kf2], Sz[lk_.kf2], er, linearize ? J : nullptr);
            double Oe[9], c2 = 0;
            for (int r = 0; r < 9; ++r) { double s = 0; for (int k = 0; k < 9; ++k) s += lk_.info[9 * r + k] * er[k]; Oe[r] = s; c2 += er[r] * s; }
            double rho0 = c2, rho1 = 1.0;
            if (lk_.robust) huber(c2, d_imu, dsqr_imu, rho0, rho1);
            chi += rho0;
            double eg[3], ea[3], Og[3], Oa[3];
            for (int k = 0; k < 3; ++k) { eg[k] = Sz[lk_.kf2].bg[k] - Sz[lk_.kf1].bg[k]; ea[k] = Sz[lk_.kf2].ba[k] - Sz[lk_.kf1].ba[k]; }
            for (int r = 0; r < 3; ++r) {
                Og[r] = lk_.infoG[3 * r] * eg[0] + lk_.infoG[3 * r + 1] * eg[1] + lk_.infoG[3 * r + 2] * eg[2];
                Oa[r] = lk_.infoA[3 * r] * ea[0] + lk_.infoA[3 * r + 1] * ea[1] + lk_.infoA[3 * r + 2] * ea[2];
                chi += eg[r] * Og[r] + ea[r] * Oa[r];
            }
            if (!linearize) continue;
            const int i1 = imu_var[lk_.kf1], i2 = imu_var[lk_.kf2], p1 = pv[lk_.kf1], p2 = pv[lk_.kf2];
            const int off[6] = {p1 >= 0 ? 6 * p1 : -1, i1 >= 0 ? np + 9 * i1 : -1, i1 >= 0 ? np + 9 * i1 + 3 : -1, i1 >= 0 ? np + 9 * i1 + 6 : -1,
                                p2 >= 0 ? 6 * p2 : -1, i2 >= 0 ? np + 9 * i2 : -1};
            const int col[6] = {0, 6, 9, 12, 15, 21}, sz[6] = {6, 3, 3, 3, 6, 3};
            double OJ[9 * 24];  // (rho' Omega) J
            for (int r = 0; r < 9; ++r)
                for (int c = 0; c < 24; ++c) { double t = 0; for (int k = 0; k < 9; ++k) t += rho1 * lk_.info[9 * r + k] * J[24 * k + c]; OJ[24 * r + c] = t; }
            for (int a = 0; a < 6; ++a) {
                if (off[a] < 0) continue;
                for (int r = 0; r < sz[a]; ++r) {
                    double s = 0;
                    for (int k = 0; k < 9; ++k) s += J[24 * k + col[a] + r] * (rho1 * Oe[k]);
                    bi[off[a] + r] -= s;
                    for (int b2 = 0; b2 < 6; ++b2) {
                        if (off[b2] < 0) continue;
                        note_segment((size_t)(off[a] + r) * n + off[b2], sz[b2]);
                        for (int c = 0; c < sz[b2]; ++c) {
                            double h = 0;
                            for (int k = 0; k < 9; ++k) h += J[24 * k + col[a] + r] * OJ[24 * k + col[b2] + c];
                            Hi[(size_t)(off[a] + r) * n + off[b2] + c] += h;
                        }
                    }
                }
            }
            for (int which = 0; which < 2; ++which) {  // EdgeGyroRW / EdgeAccRW: J = (-I, I)
                const double* Om = which == 0 ? lk_.infoG : lk_.infoA;
                const double* Oe3 = which == 0 ? Og : Oa;
                const int o1 = i1 >= 0 ? np + 9 * i1 + 3 + 3 * which : -1, o2 = i2 >= 0 ? np + 9 * i2 + 3 + 3 * which : -1;
                for (int r = 0; r < 3; ++r) {
                    if (o1 >= 0) { note_segment((size_t)(o1 + r) * n + o1, 3); if (o2 >= 0) note_segment((size_t)(o1 + r) * n + o2, 3); }
                    if (o2 >= 0) { note_segment((size_t)(o2 + r) * n + o2, 3); if (o1 >= 0) note_segment((size_t)(o2 + r) * n + o1, 3); }
                    if (o1 >= 0) bi[o1 + r] += Oe3[r];
                    if (o2 >= 0) bi[o2 + r] -= Oe3[r];
                    for (int c = 0; c < 3; ++c) {
                        if (o1 >= 0) Hi[(size_t)(o1 + r) * n + o1 + c] += Om[3 * r + c];
                        if (o2 >= 0) Hi[(size_t)(o2 + r) * n + o2 + c] += Om[3 * r + c];
                        if (o1 >= 0 && o2 >= 0) { Hi[(size_t)(o1 + r) * n + o2 + c] -= Om[3 * r + c]; Hi[(size_t)(o2 + r) * n + o1 + c] -= Om[3 * r + c]; }
                    }
                }
            }
        }
        if (linearize) segs_ready = true;
        return chi;
    }
};


// ---- lock-step batch -----------------------------------------------------------------------------------------------------
// All windows advance through the phases of the Levenberg-Marquardt loop together: one launch per kernel and one stream
// synchronisation per phase for the whole batch (a window alone is bound by launch and synchronisation latency: its kernels take
// 5-20 us each).  The arithmetic of a window is the one of tc2li_local_lv_bundle_adjustment (same kernel bodies, same host
// steps), so the results are identical to optimising the windows one by one.
struct LockstepWindow {
    const tc2li_ba_problem* p = nullptr;
    BaWorkspace* ws = nullptr;
    VisualProblem vp;
    BalmTerm* lidar = nullptr;
    std::vector<uint8_t> extra_used;
    std::vector<double> Swork, x;
    double *Hl = nullptr, *bl_ = nullptr;  // the LiDAR term's (6K)^2 Hessian and 6K gradient (pinned: ws->h_Hl)
    double lambda = -1, ni = 2, currentChi = 0, tempChi = 0, iniChi = 0, rho = 0, scale = 0, max_pose_diag = 0;
    int n_bad = 0, done = 0, trials_total = 0, qmax = 0, it = 0, rc = 0;
    int parity = 0;  // 1: the accepted estimate lives in the trial buffers of the slot (an odd number of accepted steps)
    bool ok = true, ok2 = true, need_diag = false, want_maxdiag = false;
    bool wants_hpp() const { return need_diag; }
    bool stopped() const { return p->stop_flag && *p->stop_flag; }
    bool wants_iteration() const { return rc >= 0 && it < p->iterations && !stopped() && ok; }
};

// The windows `list[c0 .. c1)` of a phase as kernel arguments (ba_device.hpp BaPhase): table index, parity / request bits, lambda.
template <typename Win>
BaPhase make_phase(const BaBatchSlot* d_table, const double* d_xp_area, const std::vector<Win>& W, const std::vector<int>& list, size_t c0, size_t c1, int expect = 0) {
    BaPhase ph;
    ph.table = d_table; ph.xp_area = d_xp_area; ph.first = (int32_t)c0; ph.pad_ = 0; ph.expect = expect; ph.pad2_ = 0;
    for (size_t k = c0; k < c1; ++k) {
        const Win& w = W[list[k]];
        ph.win[k - c0] = (uint16_t)list[k];
        ph.flags[k - c0] = (uint8_t)((w.parity ? kBaAcceptedInTrial : 0u) | (w.want_maxdiag ? kBaWantMaxdiag : 0u) | (w.wants_hpp() ? kBaWantHpp : 0u));
        ph.lambda[k - c0] = w.lambda;
    }
    return ph;
}
// fn(phase, windows in it) for every piece of at most kBaPhaseMax windows of `list`
template <typename Win, typename Fn>
void for_phase_pieces(const BaBatchSlot* d_table, const double* d_xp_area, const std::vector<Win>& W, const std::vector<int>& list, Fn&& fn, int expect = 0) {
    for (size_t c0 = 0; c0 < list.size(); c0 += kBaPhaseMax) {
        const size_t c1 = std::min(list.size(), c0 + (size_t)kBaPhaseMax);
        const BaPhase ph = make_phase(d_table, d_xp_area, W, list, c0, c1, expect);
        fn(ph, (int)(c1 - c0));
    }
}

struct LockstepContext {
    std::mutex mu;
    std::vector<std::unique_ptr<BaWorkspace>> ws;
    // slot table and, behind it, the two index lists of a phase: one host buffer, one device buffer, one copy per phase
    DevBuf<uint8_t> d_table;
    PinnedBuf<uint8_t> h_table;
    PinnedBuf<CopyTask> h_tasks;  // the uploads / operand fills of a batch's setup, then its result copies: one launch each (copy_kernels.hip)
    PinnedBuf<CopyTask> h_table_task;  // the steps x_p of a trial phase on their way up: one entry for k_copy_tasks
    // plane extraction of the batch's LiDAR windows on the device (balm_cut_kernels.hip): a task per window, those with points to cut
    // compacted into the list the kernels read, and the uploads the LiDAR tasks deferred (the clouds)
    PinnedBuf<BalmCutTask> h_cut, h_cut_list;
    DevBuf<BalmCutTask> d_cut_list;
    PinnedBuf<CopyTask> h_cut_copies;
    // device-side LM: the windows' states (device), their initial values and the mirror the decide kernel writes (pinned), the stop words
    DevBuf<BaLmState> d_lm;
    PinnedBuf<BaLmState> h_lm_init, h_lm;
    PinnedBuf<int32_t> h_stop;
    hipEvent_t round_done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    ~LockstepContext() {
        for (hipEvent_t e : round_done) if (e) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    }
};
struct LockstepContexts { LockstepContext c[kMaxLockstepGroups]; };
inline LockstepContext& lockstep_ctx(int group) { return shutdown_owned<LockstepContexts, 0>().c[group]; }


// The LiDAR half of a lock-step batch's setup, after its tasks (the odd entries of `deferred`) have staged the clouds and described the
// extractions in C.h_cut[0..n): the uploads and the extraction kernels of all windows are queued in one go, so that they run while the
// host builds the visual structure.  plane_extraction_finish (after that) waits for them and gives every window its planes.
inline bool plane_extraction_begin(LockstepContext& C, std::vector<std::vector<CopyTask>>& deferred, int n, hipStream_t st) {
    int m = 0, max_points = 0, max_table = 0;
    if (C.h_cut_list.ensure(std::max(n, 1)) != hipSuccess || C.d_cut_list.ensure(std::max(n, 1)) != hipSuccess) return false;
    for (int i = 0; i < n; ++i) {
        const BalmCutTask& t = C.h_cut.p[i];
        if (t.n_points <= 0) continue;
        C.h_cut_list.p[m++] = t;
        max_points = std::max(max_points, t.n_points);
        max_table = std::max(max_table, 1 << t.table_bits);
    }
    if (!m) return true;
    size_t n_copies = 1, max_bytes = (size_t)m * sizeof(BalmCutTask);
    for (int i = 0; i < n; ++i) n_copies += deferred[2 * (size_t)i + 1].size();
    if (C.h_cut_copies.ensure(n_copies) != hipSuccess) return false;
    size_t at = 0;
    C.h_cut_copies.p[at++] = CopyTask{C.d_cut_list.p, C.h_cut_list.p, (size_t)m * sizeof(BalmCutTask)};
    for (int i = 0; i < n; ++i) {
        for (const CopyTask& t : deferred[2 * (size_t)i + 1]) { C.h_cut_copies.p[at++] = t; max_bytes = std::max(max_bytes, t.bytes); }
        deferred[2 * (size_t)i + 1].clear();
    }
    launch_copy_tasks(C.h_cut_copies.p, (int)n_copies, max_bytes, st);
    launch_balm_cut(C.d_cut_list.p, m, max_points, max_table, st);
    return hipGetLastError() == hipSuccess;
}
// rc per window (0, or the error of a window whose planes could not be set up)
inline bool plane_extraction_finish(LockstepContext& C, int n, std::vector<int>& rc_lidar, hipStream_t st) {
    bool any = false;
    for (int i = 0; i < n; ++i) any |= C.h_cut.p[i].n_points > 0;
    if (!any) return true;
    if (stream_wait_blocking(st) != hipSuccess) return false;
    for (int i = 0; i < n; ++i) {
        if (C.h_cut.p[i].n_points <= 0 || rc_lidar[i] < 0) continue;
        rc_lidar[i] = C.ws[i]->lidar.finish_cut(st);
    }
    return true;
}

// What the launches of a phase have to cover: the largest sizes among the listed windows and which kernel families they need.  Taken over
// the whole batch once (the fusion switches follow from it) and, with the device-side LM loop, again over the windows still alive whenever that
// list shrinks: the rounds a few stragglers need after the bulk has finished are launched for THEIR sizes and families only (no dense-path
// kernels once the last window of more than 21 free keyframes is done, the narrow solve kernel for narrow systems).
template <typename Win>
BaBatchExtent batch_extent(const std::vector<Win>& W, const std::vector<int>& list, bool* all_block_parts_out = nullptr) {
    BaBatchExtent X{};
    bool all_block_parts = true;
    for (int i : list) {
        if (W[i].rc < 0) continue;
        const BaProblemDev& pb = W[i].vp.pb;
        X.max_edges = std::max(X.max_edges, pb.n_edges); X.max_points = std::max(X.max_points, pb.n_points); X.max_poses = std::max(X.max_poses, pb.n_poses);
        X.max_free = std::max(X.max_free, pb.n_free); X.max_free_edges = std::max(X.max_free_edges, pb.n_free_edges); X.max_groups = std::max(X.max_groups, pb.n_groups);
        if (!(pb.sparse_schur && pb.schur_blocks && pb.n_free > 0 && W[i].vp.n_slices > 0)) all_block_parts = false;
        if (pb.trial_fused) X.any_trial_fused = 1; else X.any_trial_unfused = 1;
        if (pb.n_dups) X.any_dups = 1;
        if (pb.sparse_schur && pb.schur_blocks) {
            if (pb.n_free > 0 && W[i].vp.n_slices > 0) {
                X.min_block_free = X.max_block_parts ? std::min(X.min_block_free, pb.n_free) : pb.n_free;
                X.max_block_parts = std::max(X.max_block_parts, W[i].vp.n_slices); X.max_block_free = std::max(X.max_block_free, pb.n_free);
                X.any_block_lean = 1;
                if (pb.n_free > kSchurBlocksMaxFree) X.any_block_wide = 1;
            }
        } else if (!pb.sparse_schur) {
            X.any_dense = 1; X.max_np_pad = std::max(X.max_np_pad, pb.np_pad); X.max_slices = std::max(X.max_slices, W[i].vp.n_slices);
        }
        if (W[i].lidar) {
            X.max_planes = std::max(X.max_planes, W[i].lidar->n_planes); X.max_chunks = std::max(X.max_chunks, W[i].lidar->dev.n_chunks);
            X.max_W = std::max(X.max_W, W[i].lidar->W);
        }
    }
    if (all_block_parts_out) *all_block_parts_out = all_block_parts;
    return X;
}


// ---- what the device-side Levenberg-Marquardt loop's two drivers share (ba_batch_lockstep's device-LM form, tc2li_ba_engine) ----
// The table entry and the initial optimiser state of window w in slot i of context C: everything a phase leaves for the next one stays in
// device memory, the decide kernel mirrors the window's state into C.h_lm[i].  The entry goes up with the caller's next copy launch
// (C.d_table + i), the state from C.h_lm_init[i] to C.d_lm[i].
inline void fill_device_lm_slot(BaBatchSlot& s, const LockstepWindow& w, LockstepContext& C, int i) {
    s = BaBatchSlot{};
    s.pb = w.vp.pb;
    s.lambda_init = w.p->lambda_init; s.iterations = w.p->iterations;
    s.n_slices = w.vp.n_slices; s.k_per_slice = w.vp.k_per_slice; s.has_lidar = w.lidar != nullptr;
    const size_t nn = (size_t)w.vp.np * w.vp.np;
    double* sc = w.ws->d_scal.p;
    s.chi_out = sc; s.maxdiag_out = sc + 1; s.scale_out = sc + 3; s.chi_trial_out = sc + 4;
    s.S_out = w.ws->d_S.p; s.bs_out = w.ws->d_bs.p; s.xp = s.x_dev = w.ws->d_xp.p; s.depth_out = w.ws->d_depth.p;
    s.lm = C.d_lm.p + i; s.lm_host = C.h_lm.p + i; s.stop_host = C.h_stop.p + i; s.ok_host = &s.lm->solve_ok;
    if (w.lidar) {
        s.balm = w.lidar->dev;
        s.Hl = w.ws->d_Hl.p; s.bl_lidar = w.ws->d_Hl.p + nn; s.balm.out = w.ws->d_balm_out.p; s.lidar_JH = w.ws->d_lidar_JH.p;
        s.lidar_information = w.lidar->information;
    }
    BaLmState& m = C.h_lm_init.p[i];
    m = BaLmState{};
    m.lambda = -1; m.ni = 2; m.r1 = 1000; m.r2 = 1000; m.is_calc_hess = 1; m.ok = 1; m.solve_ok = 1;
    m.status = w.wants_iteration() ? kLmIterate : kLmDone;
    C.h_lm.p[i] = m;
    C.h_stop.p[i] = 0;
}
// One round of the windows in `live` queued on st as ONE sequence without a host decision in it: linearise (windows whose status is
// kLmIterate) + the plane term + begin, then Schur + solve + trial estimate + errors + decide (windows in kLmTrial).  live_lidar: those of
// `live` with a LiDAR edge; lidar_first: those of them whose plane residual at the current estimate is not in place yet (their first
// linearisation: later the accepted estimate is the last trial, whose residual and decompositions are).
inline void queue_lm_round(const BaBatchSlot* d_table, const std::vector<LockstepWindow>& W, const std::vector<int>& live, const std::vector<int>& live_lidar,
                           const std::vector<int>& lidar_first, const BaBatchExtent& XL, bool want_maxdiag, hipStream_t st) {
    auto pieces_for = [&](const std::vector<int>& list, int expect, auto&& fn) { for_phase_pieces(d_table, (const double*)nullptr, W, list, fn, expect); };
    pieces_for(live, kLmIterate, [&](const BaPhase& ph, int cnt) { ba_batch_launch_linearize(ph, cnt, XL, want_maxdiag, st); });
    pieces_for(lidar_first, kLmIterate, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, false, st); });
    pieces_for(live_lidar, kLmIterate, [&](const BaPhase& ph, int cnt) { balm_batch_launch_hessian(ph, cnt, XL, st); });
    pieces_for(live, kLmIterate, [&](const BaPhase& ph, int cnt) { ba_batch_launch_lm_begin(ph, cnt, st); });
    pieces_for(live, kLmTrial, [&](const BaPhase& ph, int cnt) {
        ba_batch_launch_schur(ph, cnt, XL, st);
        ba_batch_launch_solve(ph, cnt, XL, st);
        ba_batch_launch_trial(ph, cnt, XL, st);
    });
    if (XL.any_trial_unfused) pieces_for(live_lidar, kLmTrial, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, true, st); });  // (windows with pb.trial_fused: inside the trial kernel)
    pieces_for(live, kLmTrial, [&](const BaPhase& ph, int cnt) { ba_batch_launch_lm_decide(ph, cnt, st); });
}

// the lock-step drivers (ba_lockstep.cpp, lvi_host.cpp): false = the batch goes through the one-window path
bool ba_batch_lockstep(const tc2li_ba_problem* problems, int n, const tc2li_camera* cam, WorkerPool& pool, int32_t* results, int group = 0);

}  // namespace ba_detail
}  // namespace tc2li
