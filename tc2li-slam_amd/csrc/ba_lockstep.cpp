// The lock-step batch of LocalBundleAdjustment / LocalLVBundleAdjustment windows: ba_batch_lockstep and the batch entry points (include/tc2li_hip.h).
#include "ba_internal.hpp"

using namespace tc2li;
using namespace tc2li::ba_detail;

namespace tc2li {
namespace ba_detail {

// returns false when the batch has to go through the one-thread-per-window path (a LiDAR window outside the batched kernels' range)
bool ba_batch_lockstep(const tc2li_ba_problem* problems, int n, const tc2li_camera* cam, WorkerPool& pool, int32_t* results, int group) {
    LockstepContext& C = lockstep_ctx(group);
    std::lock_guard<std::mutex> lk(C.mu);
    const BaOptions opt = BaOptions::read();
    for (int i = 0; i < n; ++i)
        if (problems[i].lidar && (problems[i].lidar->n_keyframes > 7)) return false;
    if (!C.st) {
        // the loop is a chain of ~140 short dependent launches: on a GPU shared with the front-end kernels they go first.  (Round 6 also tried
        // compute units of their own -- hipExtStreamCreateWithCUMask: the lock-step groups on 32 / 64 / 96 of the 256, every other stream of the
        // loop on the rest.  A chain of tiny kernels beside GEMMs gains 25x from that; this one does not: its large kernels want the whole chip --
        // 512 sequences 9.6 / 14.7 / 16.8 k frames/s against 20.2 k unpartitioned, 64 sequences 8.3 / 12.0 k against 15.1 k.  Removed.)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&C.st, hipStreamNonBlocking, hi) != hipSuccess &&
            hipStreamCreateWithFlags(&C.st, hipStreamNonBlocking) != hipSuccess) { C.st = nullptr; return false; }
    }
    hipStream_t st = C.st;
    while ((int)C.ws.size() < n) C.ws.emplace_back(new BaWorkspace());
    // the slot table: one slot per window, filled once after the setup and uploaded with the windows' input blocks; what a phase changes
    // (which windows take part, lambda, which of a slot's two buffers holds the accepted estimate) travels in the kernels' arguments
    // (BaPhase).  Behind the table: the steps x_p of the windows of a trial phase (kBaXpStride doubles each), so that the trial kernels
    // read them from device memory (a window with more free keyframes than that keeps reading the solver's pinned buffer).
    if (n > 65535) return false;  // BaPhase names a window by 16 bits
    constexpr size_t kXpStride = kBaXpStride;
    const size_t table_bytes = (size_t)n * sizeof(BaBatchSlot), xp_bytes = (size_t)n * kXpStride * sizeof(double);
    if (C.d_table.ensure(table_bytes + xp_bytes) != hipSuccess || C.h_table.ensure(table_bytes + xp_bytes) != hipSuccess) return false;
    BaBatchSlot* const h_slots = (BaBatchSlot*)C.h_table.p;
    double* const h_xp_area = (double*)(C.h_table.p + table_bytes);
    const BaBatchSlot* const d_table = (const BaBatchSlot*)C.d_table.p;
    double* const d_xp_area = (double*)(C.d_table.p + table_bytes);
    std::vector<LockstepWindow> W(n);
    const bool kTiming = opt.timing;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = now();
    const double t_begin = t0;
    // ---- setup: argument checks, plane extraction (device, queued first: it runs while the host builds the visual structure), uploads ----
    std::vector<int> rc_lidar(n, 0);
    std::vector<std::vector<CopyTask>> deferred(2 * (size_t)n);  // what the tasks would have queued as copies / fills of their own
    if (C.h_cut.ensure(std::max(n, 1)) != hipSuccess) return false;
    for (int i = 0; i < n; ++i) C.h_cut.p[i].n_points = 0;
    auto setup_task = [&](int task) {  // two tasks per window: the visual structure + uploads (even), the LiDAR window (odd)
        CopySink sink(&deferred[task]);
        const int i = task >> 1;
        LockstepWindow& w = W[i];
        const tc2li_ba_problem& p = problems[i];
        const bool args_ok = p.poses7 && p.fixed && p.points3 && p.edges && p.n_poses > 0 && p.n_points > 0 && p.n_edges > 0 && p.iterations >= 0;
        bool lidar_ok = true;
        if (args_ok && p.lidar) {
            if (p.lidar->n_keyframes < 1 || !p.lidar->pose_index) lidar_ok = false;
            else for (int k = 0; k < p.lidar->n_keyframes; ++k) if (p.lidar->pose_index[k] < 0 || p.lidar->pose_index[k] >= p.n_poses) lidar_ok = false;
        }
        if (task & 1) {
            if (!args_ok || !lidar_ok || !p.lidar) return;
            const double tb = now();
            rc_lidar[i] = C.ws[i]->lidar.build(p.poses7, p.n_poses, p.lidar, st, &C.h_cut.p[i]);
            if (kTiming && i == 0) fprintf(stderr, "  window 0: lidar build %.3f ms\n", now() - tb);
            return;
        }
        w.p = &p; w.ws = C.ws[i].get();
        if (!args_ok) { set_error("tc2li_local_bundle_adjustment: invalid argument"); w.rc = TC2LI_ERR_INVALID; return; }
        if (!lidar_ok) { set_error("lidar window: invalid argument or pose_index out of range"); w.rc = TC2LI_ERR_INVALID; return; }
        if (p.stats) memset(p.stats, 0, sizeof(*p.stats));
        if (p.lidar_stats) memset(p.lidar_stats, 0, sizeof(*p.lidar_stats));
        if (p.lidar) {
            w.extra_used.assign(p.n_poses, 0);
            for (int k = 0; k < p.lidar->n_keyframes; ++k) w.extra_used[p.lidar->pose_index[k]] = 1;
        }
        const double ts = now();
        w.rc = w.vp.setup(*w.ws, p.poses7, p.fixed, p.n_poses, p.points3, p.n_points, p.edges, p.n_edges, cam,
                          w.extra_used.empty() ? nullptr : w.extra_used.data(), st);
        if (kTiming && i == 0) fprintf(stderr, "  window 0: visual setup %.3f ms\n", now() - ts);
        if (w.rc < 0) return;
        const int np = w.vp.np;
        w.Swork.assign((size_t)std::max(np * np, 1), 0.0);
        w.x.assign(std::max(np, 1), 0.0);
        BaWorkspace& ws = *w.ws;
        const size_t nn = (size_t)std::max(np * np, 1), n1 = (size_t)std::max(np, 1);
        if (ws.d_S.ensure(nn) != hipSuccess || ws.d_bs.ensure(2 * n1) != hipSuccess || ws.d_xp.ensure(n1) != hipSuccess || ws.h_ok.ensure(1) != hipSuccess ||
            ws.d_scal.ensure(8) != hipSuccess) {
            w.rc = TC2LI_ERR_HIP; return;
        }
        if (p.lidar) {
            const size_t nl = 6 * (size_t)p.lidar->n_keyframes;
            if (ws.d_balm_out.ensure((size_t)balm_out_size(p.lidar->n_keyframes)) != hipSuccess || ws.d_lidar_JH.ensure(nl + nl * nl) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
            // computeLambdaInit with a LiDAR term reads the diagonal of Hpp on the host (first iteration, no lambda given): where the reduction writes it
            if (ws.h_Hpp.ensure(27 * (size_t)std::max(w.vp.n_free, 1)) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
            if (ws.d_Hl.ensure(nn + n1) != hipSuccess || ws.h_Hl.ensure(nn + n1) != hipSuccess) { w.rc = TC2LI_ERR_HIP; return; }
            w.Hl = ws.h_Hl.p; w.bl_ = ws.h_Hl.p + nn;
            std::fill(w.Hl, w.Hl + nn + n1, 0.0);
        }
    };
    pool.parallel_for(n, [&](int i) { setup_task(2 * i + 1); });
    if (!plane_extraction_begin(C, deferred, n, st)) { (void)hipStreamSynchronize(st); return false; }
    if (kTiming) fprintf(stderr, "  lidar tasks + queueing the extraction: %.3f ms\n", now() - t0);
    pool.parallel_for(n, [&](int i) { setup_task(2 * i); });
    if (kTiming) fprintf(stderr, "  + visual tasks: %.3f ms\n", now() - t0);
    if (!plane_extraction_finish(C, n, rc_lidar, st)) return false;
    if (kTiming) fprintf(stderr, "  + extraction back: %.3f ms\n", now() - t0);
    for (int i = 0; i < n; ++i) {
        if (W[i].rc < 0 || !problems[i].lidar) continue;
        if (rc_lidar[i] < 0) W[i].rc = rc_lidar[i]; else W[i].lidar = &C.ws[i]->lidar;
    }
    for (int i = 0; i < n; ++i)
        if (W[i].rc >= 0 && W[i].lidar && W[i].lidar->n_planes > 2048) {  // outside the batched LiDAR kernels: per-window path for this batch
            (void)hipStreamSynchronize(st);
            return false;
        }
    tm[0] = now() - t0;
    std::vector<int> all_windows(n);
    for (int i = 0; i < n; ++i) all_windows[i] = i;
    bool all_block_parts = true;
    BaBatchExtent X = batch_extent(W, all_windows, &all_block_parts);
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
    // TC2LI_BA_DEVICE_SOLVE=1 (read per call): the reduced systems of the batch are solved on the device (k_ba_solve_b; every window on the
    // sparse Schur path, i.e. at most 21 free keyframes) -- Schur product, solve and trial estimate are then one queue of launches with one
    // host round trip per LM trial instead of two, and the step is the host's bit for bit.  Built for VERDICT 5 and measured: the
    // workgroup-per-window LDL^T (its substitutions are serial chains through LDS) takes longer on the stream than the host's solves on
    // the pool threads plus the extra synchronisation -- 8.8 against 10.2 k frames/s at 64 sequences, no difference at 512 -- so the host
    // solve stays the default.
    const bool dev_solve = opt.device_solve && X.max_free <= kSolveMaxFree && X.max_free > 0;
    // TC2LI_BA_DEVICE_LM=0: the Levenberg-Marquardt decisions of rounds 2-5, on the host between the phases.  Default (round 6): on
    // the device (ba_device.hpp: BaLmState) for every batch whose reduced systems the solve kernel takes: at most kSolveMaxFree free keyframes.
    const bool device_lm = opt.device_lm && X.max_free <= kSolveMaxFree;
    if (device_lm && (C.d_lm.ensure(n) != hipSuccess || C.h_lm_init.ensure(n) != hipSuccess || C.h_lm.ensure(n) != hipSuccess || C.h_stop.ensure(n) != hipSuccess)) return false;
    auto fill_slot = [&](int i) {
        LockstepWindow& w = W[i];
        BaBatchSlot& s = h_slots[i];
        s.pb = w.vp.pb;
        s.lm = nullptr; s.lm_host = nullptr; s.stop_host = nullptr; s.lidar_JH = nullptr; s.lambda_init = w.p->lambda_init; s.lidar_information = 0;
        s.iterations = w.p->iterations; s.lm_pad_ = 0;
        s.n_slices = w.vp.n_slices; s.k_per_slice = w.vp.k_per_slice; s.has_lidar = w.lidar != nullptr; s.pad_ = 0;
        double* sc = w.ws->h_scal.p;
        s.chi_out = sc; s.maxdiag_out = sc + 1; s.scale_out = sc + 3; s.chi_trial_out = sc + 4;
        s.S_out = w.ws->h_S.p; s.bs_out = w.ws->h_bs.p; s.xp = w.ws->h_xp.p; s.depth_out = w.ws->d_depth.p;
        s.hpp_out = w.lidar ? w.ws->h_Hpp.p : nullptr;  // written when a phase asks for it (kBaWantHpp)
        s.iposes_host = nullptr;
        s.bp_host = nullptr; s.Hl = s.bl_lidar = nullptr; s.x_dev = s.x_host = nullptr; s.ok_host = nullptr;
        if (dev_solve) {
            const size_t nn = (size_t)w.vp.np * w.vp.np;
            s.S_out = w.ws->d_S.p; s.bs_out = w.ws->d_bs.p; s.bp_host = w.ws->h_bs.p + w.vp.np;
            s.xp = s.x_dev = w.ws->d_xp.p; s.x_host = w.ws->h_xp.p; s.ok_host = w.ws->h_ok.p;
            if (w.lidar) { s.Hl = w.ws->d_Hl.p; s.bl_lidar = w.ws->d_Hl.p + nn; }
        }
        if (w.lidar) s.balm = w.lidar->dev; else s.balm = BalmDev{};
        if (device_lm) fill_device_lm_slot(s, w, C, i);   // (replaces the host-loop entries above: everything stays in device memory)
    };
    // the per-window host steps between two phases are tens of microseconds each: few windows run on the calling thread
    // (a pool dispatch costs more than it saves, and far more on a busy host)
    auto phase_for = [&](int cnt, const std::function<void(int)>& fn) { pool.parallel_for(cnt, fn); };
    bool failed = false;
    // the table and everything the setup deferred (uploads, operand fills): one launch; the windows' megabyte input blocks go through the copy engines (launch_copy_tasks)
    {
        for (int i = 0; i < n; ++i)
            if (W[i].rc >= 0) fill_slot(i);
            else { h_slots[i] = BaBatchSlot{}; if (device_lm) { C.h_lm_init.p[i] = BaLmState{}; C.h_lm.p[i] = BaLmState{}; C.h_stop.p[i] = 0; } }
        size_t n_tasks = device_lm ? 2 : 1, max_bytes = std::max(table_bytes, device_lm ? (size_t)n * sizeof(BaLmState) : (size_t)0);
        for (const auto& d : deferred) n_tasks += d.size();
        if (C.h_tasks.ensure(n_tasks) != hipSuccess) return false;
        size_t at = 0;
        C.h_tasks.p[at++] = CopyTask{C.d_table.p, C.h_table.p, table_bytes};
        if (device_lm) C.h_tasks.p[at++] = CopyTask{C.d_lm.p, C.h_lm_init.p, (size_t)n * sizeof(BaLmState)};
        for (const auto& d : deferred) for (const CopyTask& t : d) { C.h_tasks.p[at++] = t; max_bytes = std::max(max_bytes, t.bytes); }
        launch_copy_tasks(C.h_tasks.p, (int)n_tasks, max_bytes, st);
    }
    // the steps of a trial phase: window k of `step` at h_xp_area + k * kXpStride, up through a one-entry k_copy_tasks launch on the group's
    // own stream (not hipMemcpyAsync: the runtime's copy path is where the other groups' 1.4 MB window blocks are queued)
    // No upload launch for the steps: the trial kernels read a window's step (<= 1.5 KB, once per workgroup) from the pinned staging area
    // over the bus -- ten launches fewer per call at the same speed (BA stage alone 15.3 / 15.4 ms per 128 windows, the loop 28.5-28.7 /
    // 28.6-29.1 ms per step against the one-entry k_copy_tasks launch of rounds 3-4)
    constexpr bool xp_pinned = true;
    auto stage_steps = [&](const std::vector<int>& step) {
        for (size_t k = 0; k < step.size(); ++k) {
            const LockstepWindow& w = W[step[k]];
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
    auto sync = [&] { if (hipGetLastError() != hipSuccess || (device_lm ? stream_wait_blocking(st) : hipStreamSynchronize(st)) != hipSuccess) failed = true; };

    // ---- device-side LM: rounds queued ahead of the device, one status read per window and round ----
    // A round = [linearisation set: the windows in kLmIterate | trial set: the windows in kLmTrial]; a window that accepted its step takes both
    // halves of the next round, one that rejected it only the second, each at its own pace.  The host's lists are what it last SAW alive -- a
    // superset: the kernels themselves skip a window whose status is not the launch's -- so round r + 1 is queued before round r has been
    // waited for and the device never idles on the host; the wait is a sleeping one (no spinning thread per group: 2.8 of the 13 CPUs the
    // loop kept busy in round 5), and the reduced solves, the LiDAR term's change of variables and the LM bookkeeping (5.3 more) are gone
    // from the pool threads.  The caller's stop flag is polled at every round and handed to the decide kernel through a pinned word.
    if (device_lm) {
        for (hipEvent_t& e : C.round_done)
            if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; failed = true; }
        std::vector<int> live, live_lidar;
        bool first_maxdiag = false;
        BaBatchExtent XL = X;   // the extent of the windows still alive (the batch's fusion switches stay: a window's sums keep their order)
        auto refresh = [&] {
            const size_t before = live.size();
            live.clear(); live_lidar.clear();
            for (int i = 0; i < n; ++i)
                if (W[i].rc >= 0 && C.h_lm.p[i].status != kLmDone) { live.push_back(i); if (W[i].lidar) live_lidar.push_back(i); }
            if (live.size() != before && !live.empty()) {
                XL = batch_extent(W, live);
                XL.fuse_trial = X.fuse_trial; XL.fuse_linearize = X.fuse_linearize; XL.inertial = X.inertial;
            }
        };
        refresh();
        for (int i : live) first_maxdiag |= !(W[i].p->lambda_init > 0);
        int queued = 0, seen = 0;
        auto queue_round = [&] {
            const double tq = now();
            const bool first = queued == 0;
            queue_lm_round(d_table, W, live, live_lidar, first ? live_lidar : std::vector<int>(), XL, first && first_maxdiag, st);
            if (hipGetLastError() != hipSuccess || hipEventRecord(C.round_done[queued & 1], st) != hipSuccess) failed = true;
            ++queued;
            tm[6] += now() - tq;  // the host's time to queue the rounds
        };
        t0 = now();
        while (!live.empty() && !failed) {
            if (queued == seen) queue_round();
            // one round ahead while some window cannot be finished by what is queued (it has iterations left even if every queued trial is accepted)
            if (!failed && queued - seen < 2) {
                bool more = false;
                for (int i : live) more |= C.h_lm.p[i].it + (queued - seen) < W[i].p->iterations;
                if (more) queue_round();
            }
            if (failed || event_wait_sleeping(C.round_done[seen & 1]) != hipSuccess) { failed = true; break; }
            ++seen;
            for (int i : live) if (W[i].stopped()) C.h_stop.p[i] = 1;
            refresh();
        }
        if (!failed && queued > seen && event_wait_sleeping(C.round_done[(queued - 1) & 1]) != hipSuccess) failed = true;  // (a round queued ahead that found nothing to do)
        tm[1] += now() - t0;
        for (int i = 0; i < n && !failed; ++i) {
            LockstepWindow& w = W[i];
            if (w.rc < 0) continue;
            const BaLmState& m = C.h_lm.p[i];
            w.lambda = m.lambda; w.currentChi = m.currentChi; w.done = m.done; w.it = m.it; w.trials_total = m.trials_total; w.parity = m.parity;
            if (m.parity) { std::swap(w.vp.pb.poses, w.vp.pb.poses_trial); std::swap(w.vp.pb.points, w.vp.pb.points_trial); }
            if (w.p->stats && m.done > 0) { w.p->stats->initial_chi2 = m.initial_chi2; w.p->stats->final_chi2 = m.currentChi; w.p->stats->final_lambda = m.lambda; }
            if (w.lidar) { w.lidar->error = m.lidar_error; w.lidar->hessian_evaluations = m.hessian_evaluations; }
        }
    }
    for (; !device_lm;) {
        std::vector<int> active, with_lidar;
        for (int i = 0; i < n; ++i) if (W[i].wants_iteration()) active.push_back(i);
        if (active.empty() || failed) break;
        // ---- phase A: linearisation at the accepted estimate ----
        t0 = now();
        bool any_maxdiag = false;
        for (int i : active) {
            LockstepWindow& w = W[i];
            w.want_maxdiag = w.it == 0 && !(w.p->lambda_init > 0);
            w.need_diag = w.lidar && w.want_maxdiag && w.vp.n_free > 0;
            any_maxdiag |= w.want_maxdiag;
            if (w.lidar) with_lidar.push_back(i);
        }
        pieces(active, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_linearize(ph, cnt, X, any_maxdiag, st); });
        // (running the LiDAR kernels on a second stream of the group beside the visual ones -- fork / join by events around the plane
        // Hessian and around the planes' residual of a trial -- was measured twice: round 3 with the BA stage alone, no gain; round 4 in
        // the whole loop, three A/B pairs in one call: 30.2-30.5 ms per step against 28.8-29.5 without: the events' cross-stream waits cost
        // more than the overlap of two short kernels brings)
        // the residual pass at the accepted estimate: only before the first iteration -- later the accepted estimate is the last
        // trial, whose residual and plane decompositions are still in place (same bits)
        bool first_pass = false;
        for (int i : with_lidar) first_pass |= W[i].it == 0;
        pieces(with_lidar, nullptr, [&](const BaPhase& ph, int cnt) {
            if (first_pass) balm_batch_launch_residual(ph, cnt, false, st);
            balm_batch_launch_hessian(ph, cnt, X, st);
        });
        // Round 5: the first trial's Schur product does not wait for the host -- its operands are the linearisation's, its damping the window's
        // current lambda (known unless this is the first iteration of a window whose lambda comes from computeLambdaInit) -- so it is queued
        // behind the linearisation and the phase's one synchronisation covers both: a host round trip fewer per iteration, and the host's part
        // of the linearisation (the LiDAR term's change of variables) runs beside the product.  TC2LI_BA_PRE_SCHUR=0: queued after the host's part.
        constexpr bool kPreSchur = true;
        bool pre_schur = kPreSchur && !dev_solve && !any_maxdiag;
        if (pre_schur) {
            for (int i : active) if (W[i].it == 0) W[i].lambda = W[i].p->lambda_init;  // (what the host's part sets below)
            pieces(active, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_schur(ph, cnt, X, st); });
        }
        tm[6] += now() - t0;  // of the phase: the time to queue it
        sync();
        if (failed) break;
        tm[1] += now() - t0; t0 = now();
        phase_for((int)active.size(), [&](int k) {
            LockstepWindow& w = W[active[k]];
            const int np = w.vp.np;
            const double* sc = w.ws->h_scal.p;
            w.currentChi = sc[0];
            w.max_pose_diag = sc[2];
            if (w.lidar) {
                w.lidar->finish_error();
                w.currentChi = w.lidar->chi2() + w.currentChi;
                w.lidar->finish_linearization();
                std::fill(w.Hl, w.Hl + (size_t)np * np, 0.0);
                std::fill(w.bl_, w.bl_ + np, 0.0);
                w.lidar->add_quadratic_form(w.vp.pose_var.data(), np, w.Hl, w.bl_);
                if (w.need_diag) {
                    static const int dpos[6] = {0, 6, 11, 15, 18, 20};
                    w.max_pose_diag = 0;
                    for (int j = 0; j < np; ++j)
                        w.max_pose_diag = std::max(w.max_pose_diag, std::fabs(w.ws->h_Hpp.p[27 * (size_t)(j / 6) + dpos[j % 6]] + w.Hl[(size_t)j * np + j]));
                }
            }
            w.tempChi = w.currentChi;
            w.iniChi = w.currentChi;
            if (w.it == 0) {
                if (w.p->stats) w.p->stats->initial_chi2 = w.currentChi;
                w.lambda = w.p->lambda_init > 0 ? w.p->lambda_init : 1e-5 * std::max(sc[1], w.max_pose_diag);
                w.ni = 2;
                w.n_bad = 0;
            }
            w.rho = 0;
            w.qmax = 0;
        });
        if (dev_solve && !with_lidar.empty()) {  // the LiDAR term of this linearisation goes where the solve kernel adds it (one launch)
            if (C.h_tasks.ensure(with_lidar.size()) != hipSuccess) { failed = true; break; }
            size_t max_bytes = 0;
            for (size_t k = 0; k < with_lidar.size(); ++k) {
                LockstepWindow& w = W[with_lidar[k]];
                const size_t bytes = ((size_t)w.vp.np * w.vp.np + w.vp.np) * sizeof(double);
                C.h_tasks.p[k] = CopyTask{w.ws->d_Hl.p, w.ws->h_Hl.p, bytes};
                max_bytes = std::max(max_bytes, bytes);
            }
            launch_copy_tasks(C.h_tasks.p, (int)with_lidar.size(), max_bytes, st);
        }
        // ---- trials ----
        tm[2] += now() - t0;
        std::vector<int> trial = active;
        while (!trial.empty() && !failed) {
            // phase B: reduced camera system at the window's lambda
            t0 = now();
            if (dev_solve) {
                // phases B + C in one queue: Schur product, solve, trial estimate and its cost; the host sees the step, whether the
                // factorisation went through, and the sums at the one synchronisation
                std::vector<int> trial_lidar;
                for (int i : trial) if (W[i].lidar) trial_lidar.push_back(i);
                pieces(trial, nullptr, [&](const BaPhase& ph, int cnt) {
                    ba_batch_launch_schur(ph, cnt, X, st);
                    ba_batch_launch_solve(ph, cnt, X, st);
                    ba_batch_launch_trial(ph, cnt, X, st);
                });
                if (X.any_trial_unfused) pieces(trial_lidar, nullptr, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, true, st); });  // (windows with pb.trial_fused: inside the trial launch)
                sync();
                if (failed) break;
                tm[3] += now() - t0; t0 = now();
                for (int i : trial) {
                    LockstepWindow& w = W[i];
                    const int np = w.vp.np;
                    BaWorkspace& ws = *w.ws;
                    w.ok2 = np == 0 || ws.h_ok.p[0] != 0;
                    w.scale = 0;
                    // pose part of computeScale(): b_p (+ the LiDAR gradient) as the host path has it in h_bs[np .. 2 np)
                    for (int j = 0; j < np; ++j) {
                        const double bpj = w.lidar ? ws.h_bs.p[np + j] + w.bl_[j] : ws.h_bs.p[np + j];
                        w.scale += ws.h_xp.p[j] * (w.lambda * ws.h_xp.p[j] + bpj);
                    }
                }
                tm[4] += now() - t0; t0 = now();
            } else {
            if (pre_schur) pre_schur = false;  // (the product of this trial came with the linearisation)
            else {
                pieces(trial, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_schur(ph, cnt, X, st); });
                sync();
                if (failed) break;
            }
            tm[3] += now() - t0; t0 = now();
            phase_for((int)trial.size(), [&](int k) {
                LockstepWindow& w = W[trial[k]];
                const int np = w.vp.np;
                BaWorkspace& ws = *w.ws;
                w.ok2 = true;
                if (np > 0) {
                    memcpy(w.Swork.data(), ws.h_S.p, (size_t)np * np * sizeof(double));
                    if (w.lidar) {
                        for (size_t q = 0; q < (size_t)np * np; ++q) w.Swork[q] += w.Hl[q];
                        for (int j = 0; j < np; ++j) { ws.h_bs.p[j] += w.bl_[j]; ws.h_bs.p[np + j] += w.bl_[j]; }
                    }
                    w.ok2 = ldlt_solve_small(w.Swork.data(), np, ws.h_bs.p, w.x.data(), false);
                    memcpy(ws.h_xp.p, w.x.data(), np * sizeof(double));
                }
                w.scale = 0;
                for (int j = 0; j < np; ++j) w.scale += w.x[j] * (w.lambda * w.x[j] + ws.h_bs.p[np + j]);
            });
            // phase C: the trial estimate and its cost
            tm[4] += now() - t0; t0 = now();
            std::vector<int> step, step_lidar;
            for (int i : trial) if (W[i].ok2) { step.push_back(i); if (W[i].lidar) step_lidar.push_back(i); }
            if (!step.empty()) {
                stage_steps(step);
                pieces(step, xp_pinned ? h_xp_area : d_xp_area, [&](const BaPhase& ph, int cnt) { ba_batch_launch_trial(ph, cnt, X, st); });
                if (X.any_trial_unfused) pieces(step_lidar, nullptr, [&](const BaPhase& ph, int cnt) { balm_batch_launch_residual(ph, cnt, true, st); });  // (windows with pb.trial_fused: inside the trial launch)
                sync();
                if (failed) break;
            }
            }
            tm[5] += now() - t0; t0 = now();
            std::vector<int> again;
            for (int i : trial) {
                LockstepWindow& w = W[i];
                if (w.ok2) {
                    const double* sc = w.ws->h_scal.p;
                    w.tempChi = sc[4];
                    w.scale += sc[3];
                    if (w.lidar) { w.lidar->finish_error(); w.tempChi = w.lidar->chi2() + w.tempChi; }
                } else {
                    w.tempChi = std::numeric_limits<double>::max();
                }
                w.rho = w.currentChi - w.tempChi;
                w.scale += 1e-3;
                w.rho /= w.scale;
                if (w.rho > 0 && std::isfinite(w.tempChi)) {
                    w.lambda = lm_lambda_accepted(w.lambda, w.rho);
                    w.ni = 2;
                    w.currentChi = w.tempChi;
                    std::swap(w.vp.pb.poses, w.vp.pb.poses_trial);  // the host's record (the results are read through it); the device's view: parity
                    std::swap(w.vp.pb.points, w.vp.pb.points_trial);
                    w.parity ^= 1;
                } else {
                    w.lambda *= w.ni;
                    w.ni *= 2;
                }
                w.qmax++;
                w.trials_total++;
                if (w.rho < 0 && w.qmax < 10 && !w.stopped()) again.push_back(i);
            }
            trial.swap(again);
        }
        for (int i : active) {
            LockstepWindow& w = W[i];
            ++w.done;
            ++w.it;
            if (w.p->stats) { w.p->stats->final_chi2 = w.currentChi; w.p->stats->final_lambda = w.lambda; }
            if (w.qmax == 10 || w.rho == 0) { w.ok = false; continue; }
            if ((w.iniChi - w.currentChi) * 1e3 < w.iniChi) w.n_bad++; else w.n_bad = 0;
            if (w.n_bad >= 3) w.ok = false;
        }
    }
    // ---- results ----
    t0 = now();
    std::vector<int> all;
    for (int i = 0; i < n; ++i) if (W[i].rc >= 0) all.push_back(i);
    if (!failed && !all.empty()) {
        pieces(all, nullptr, [&](const BaPhase& ph, int cnt) { ba_batch_launch_depth(ph, cnt, X, st); });
        // device -> pinned staging: one launch writes every window's results (the setup's copy list is done with: the stream has been
        // synchronised many times since), then the copies into the caller's arrays run in parallel
        size_t n_tasks = 0, max_bytes = 0;
        if (C.h_tasks.ensure(4 * all.size()) != hipSuccess) failed = true;
        for (int i : all) {
            if (failed) break;
            LockstepWindow& w = W[i];
            const tc2li_ba_problem& p = *w.p;
            const BaProblemDev& pb = w.vp.pb;
            const size_t E = p.n_edges, P = p.n_points;
            const size_t bytes = p.n_poses * sizeof(Se3) + 3 * P * sizeof(double) + E * sizeof(double) + E;
            if (w.ws->h_result.ensure(bytes) != hipSuccess) { failed = true; break; }
            uint8_t* h = w.ws->h_result.p;
            uint8_t* hc = h + p.n_poses * sizeof(Se3) + 3 * P * sizeof(double);
            auto add = [&](void* dst, const void* src, size_t nbytes) { C.h_tasks.p[n_tasks++] = CopyTask{dst, src, nbytes}; max_bytes = std::max(max_bytes, nbytes); };
            add(h, pb.poses, p.n_poses * sizeof(Se3));
            add(h + p.n_poses * sizeof(Se3), pb.points, 3 * P * sizeof(double));
            if (p.edge_chi2) add(hc, w.ws->d_chi2.p, E * sizeof(double));
            if (p.edge_depth_positive) add(hc + E * sizeof(double), w.ws->d_depth.p, E);
        }
        if (!failed) launch_copy_tasks(C.h_tasks.p, (int)n_tasks, max_bytes, st);
        sync();
        if (!failed)
            pool.parallel_for((int)all.size(), [&](int k) {
                LockstepWindow& w = W[all[k]];
                const tc2li_ba_problem& p = *w.p;
                const size_t E = p.n_edges, P = p.n_points;
                const uint8_t* h = w.ws->h_result.p;
                memcpy(w.vp.poses.data(), h, p.n_poses * sizeof(Se3));
                memcpy(p.points3, h + p.n_poses * sizeof(Se3), 3 * P * sizeof(double));
                const uint8_t* hc = h + p.n_poses * sizeof(Se3) + 3 * P * sizeof(double);
                if (p.edge_chi2) memcpy(p.edge_chi2, hc, E * sizeof(double));
                if (p.edge_depth_positive) memcpy(p.edge_depth_positive, hc + E * sizeof(double), E);
            });
    }
    for (int i = 0; i < n; ++i) {
        LockstepWindow& w = W[i];
        if (w.rc < 0) { results[i] = w.rc; continue; }
        if (failed) { set_error("tc2li_local_bundle_adjustment_batch: HIP error in the lock-step loop: %s", hipGetErrorString(hipGetLastError())); results[i] = TC2LI_ERR_HIP; continue; }
        const tc2li_ba_problem& p = *w.p;
        for (int k = 0; k < p.n_poses; ++k) { memcpy(p.poses7 + 7 * k, w.vp.poses[k].q, 4 * sizeof(double)); memcpy(p.poses7 + 7 * k + 4, w.vp.poses[k].t, 3 * sizeof(double)); }
        if (p.stats) { p.stats->iterations = w.done; p.stats->trials = w.trials_total; p.stats->n_free_poses = w.vp.n_free; }
        if (w.lidar && p.lidar_stats) {
            p.lidar_stats->n_planes = w.lidar->n_planes; p.lidar_stats->hessian_evaluations = w.lidar->hessian_evaluations;
            p.lidar_stats->residual = w.lidar->error; p.lidar_stats->chi2 = w.lidar->chi2();
        }
        results[i] = w.done;
    }
    if (kTiming) fprintf(stderr, "BA lock-step timing ms (%d windows): setup %.3f linearize %.3f (queueing %.3f) host-lin %.3f schur %.3f solve %.3f trial %.3f results %.3f total %.3f\n",
                         n, tm[0], tm[1], tm[6], tm[2], tm[3], tm[4], tm[5], now() - t0, now() - t_begin);
    return true;
}



}  // namespace ba_detail
}  // namespace tc2li

extern "C" {

int tc2li_local_bundle_adjustment_batch(const tc2li_ba_problem* problems, int n_problems, const tc2li_camera* cam, int max_concurrency,
                                        int32_t* results) {
    if (n_problems < 0 || (n_problems > 0 && (!problems || !results)) || !cam) { set_error("tc2li_local_bundle_adjustment_batch: invalid argument"); return TC2LI_ERR_INVALID; }
    if (n_problems == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    const int workers = std::max(1, std::min(std::min(max_concurrency, n_problems), 16));
    WorkerPool* pool = &named_pool(kPoolBaGroup0);  // persistent: its threads keep their streams and workspaces
    const BaOptions opt = BaOptions::read();
    const bool kNoLockstep = !opt.lockstep;
    // The lock-step loop is a chain of dependent launches with a host step after every phase: while the host works the stream is
    // empty.  Several groups of windows, each a lock-step batch of its own on its own stream and host thread, fill each other's gaps.
    const int kGroups = opt.groups;
    // windows the lock-step groups could not take (a LiDAR window outside the batched kernels' range: a group that declines has written
    // nothing but zeroed stats) go through the one-window path below -- those windows only, every other window keeps its lock-step result
    std::vector<uint8_t> todo(n_problems, 1);
    if (max_concurrency > 1 && n_problems > 1 && !kNoLockstep) {
        const int groups = std::max(1, std::min(kGroups, n_problems / 2));
        // the setup of a group (per window: graph structure + staging, plane extraction of the LiDAR window) and the per-window host steps
        // between the phases (LiDAR quadratic form, 6K LDL^T) are host work on the group's own pool (common.cpp pool_threads: 16 threads per
        // group on a one-GPU box -- 8 -> 16 took the step from 41.8-42.9 to 40.7-40.8 ms in round 2, 32: 41.0 -- fewer under a smaller budget)
        auto run_group = [&](int g) {
            const int b = (int)((long)n_problems * g / groups), e = (int)((long)n_problems * (g + 1) / groups);
            if (ba_batch_lockstep(problems + b, e - b, cam, named_pool(kPoolBaGroup0 + g), results + b, g))
                std::fill(todo.begin() + b, todo.begin() + e, (uint8_t)0);
        };
        if (groups == 1) run_group(0);
        else named_pool(kPoolBaTop).parallel_for(groups, run_group);
    }
    std::vector<int> rest;
    for (int i = 0; i < n_problems; ++i) if (todo[i]) rest.push_back(i);
    if (rest.empty()) {
        int ok_ = 0;
        for (int i = 0; i < n_problems; ++i) ok_ += results[i] >= 0;
        return ok_;
    }
    struct ThreadStream {
        hipStream_t s = nullptr;
        ~ThreadStream() { if (s) (void)hipStreamDestroy(s); }
    };
    std::atomic<int> next{0};
    pool->parallel_for(std::min(workers, (int)rest.size()), [&](int) {
        static thread_local ThreadStream ts;
        if (!ts.s && hipStreamCreateWithFlags(&ts.s, hipStreamNonBlocking) != hipSuccess) ts.s = nullptr;
        for (int k; (k = next.fetch_add(1)) < (int)rest.size();) {
            const int i = rest[k];
            const tc2li_ba_problem& p = problems[i];
            results[i] = tc2li_local_lv_bundle_adjustment(p.poses7, p.fixed, p.n_poses, p.points3, p.n_points, p.edges, p.n_edges, cam,
                                                          p.iterations, p.lambda_init, p.stop_flag, p.edge_chi2, p.edge_depth_positive,
                                                          p.stats, p.lidar, p.lidar_stats, ts.s);
        }
    });
    int ok = 0;
    for (int i = 0; i < n_problems; ++i) ok += results[i] >= 0;
    return ok;
}

int tc2li_local_bundle_adjustment_batch_group(const tc2li_ba_problem* problems, int n_problems, const tc2li_camera* cam, int group, int32_t* results) {
    if (n_problems < 0 || (n_problems > 0 && (!problems || !results)) || !cam || group < 0 || group >= kMaxLockstepGroups) {
        set_error("tc2li_local_bundle_adjustment_batch_group: invalid argument (group 0 .. %d)", kMaxLockstepGroups - 1);
        return TC2LI_ERR_INVALID;
    }
    if (n_problems == 0) return 0;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    const BaOptions opt = BaOptions::read();
    const bool kNoLockstep = !opt.lockstep;
    // ONE lock-step group on the caller's thread: the context `group` (stream, work spaces, host pool) is the caller's choice, so that the
    // mapping workers of a multi-sequence system run their windows side by side without meeting at the end of a common call
    bool done = false;
    if (n_problems > 1 && !kNoLockstep) done = ba_batch_lockstep(problems, n_problems, cam, named_pool(kPoolBaGroup0 + group), results, group);
    if (!done) {  // a window outside the batched kernels' range (the group has written nothing but zeroed stats), or a batch of one
        for (int i = 0; i < n_problems; ++i) {
            const tc2li_ba_problem& p = problems[i];
            results[i] = tc2li_local_lv_bundle_adjustment(p.poses7, p.fixed, p.n_poses, p.points3, p.n_points, p.edges, p.n_edges, cam, p.iterations,
                                                          p.lambda_init, p.stop_flag, p.edge_chi2, p.edge_depth_positive, p.stats, p.lidar,
                                                          p.lidar_stats, private_stream());
        }
    }
    int ok = 0;
    for (int i = 0; i < n_problems; ++i) ok += results[i] >= 0;
    return ok;
}

}  // extern "C"
