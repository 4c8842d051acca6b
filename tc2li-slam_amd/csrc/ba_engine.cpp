// tc2li_ba_engine: running lock-step queues that windows join and leave one at a time (include/tc2li_hip.h).
#include "ba_internal.hpp"

using namespace tc2li;
using namespace tc2li::ba_detail;

namespace {

// ---- The bundle-adjustment ENGINE: continuous admission (round 6) -------------------------------------------------------------------------
// The batch entry points (ba_lockstep.cpp) are calls: a call's windows are set up together, optimised together and handed back together -- a group
// returns when its SLOWEST window is done (a window whose steps keep being rejected needs up to 26 rounds where the others need 10), and
// the setup of the next call starts only then.  With the Levenberg-Marquardt decisions on the device (BaLmState) a round no longer cares
// which call a window came with or how far it has got: every batched kernel runs for the windows whose status asks for it.  The engine keeps
// ONE stream of rounds going and lets windows in and out at the round boundaries:
//   submit    hands over windows (a ticket); they wait in a queue
//   tick      (engine thread) windows whose result copies have landed are finished and their slots freed; windows whose plane extraction
//             has run get their table entry and state and are LIVE from this round on; new windows from the queue go to free slots (best
//             fit: the work space that has held the smallest window at least as large) and are set up on the SETUP THREADS beside the rounds;
//             windows whose setup is done are staged a few at a time (clouds up, plane extraction queued); windows the last mirror shows
//             as done are RETIRED (depth flags, result copies queued); then the next round is queued for everything alive, and the tick
//             waits for the round BEFORE it -- one round is always in flight while the host works
//   wait      blocks until a ticket's windows have all been finished (poll: without blocking)
// Same kernels, same per-window arithmetic: a window's bits are those of the batch calls (tests/test_balm_gpu.py).  Work spaces grow
// through a per-engine BufferCache (common.hpp): hipFree would wait for the whole device every time a slot meets a larger window.
// Measured against the alternatives in DESIGN.md section 4, round 6 item 5: the engines are the best form of local mapping from 512 sequences
// per GPU on; below that four mapping workers that each take a whole step's windows as one group call are ahead.
enum { kSlotFree = 0, kSlotStaged = 1, kSlotLive = 2, kSlotRetiring = 3, kSlotSetup = 4 };
struct EngineTicket {
    double t_submit = 0;
    const tc2li_ba_problem* problems = nullptr;
    int32_t* results = nullptr;
    int n = 0, next = 0, remaining = 0, n_ok = 0;
    int64_t id = 0;
};
struct EngineSlot {
    int state = kSlotFree, index = 0, rc_lidar = 0;
    std::atomic<int> setup_left{0};   // the window's setup tasks (structure + staging; LiDAR window) still running on the setup threads
    double t_submit = 0, t_admit = 0, t_ready = 0, t_staged = 0, t_live = 0, t_retire = 0;   // TC2LI_BA_TIMING: where a window's time goes
    long ready_tick = 0;              // > 0: the tick its setup was first seen finished (it waits to be staged with others)
    long side_tick = 0;               // > 0: the tick whose plane-extraction launch (on the engine's side stream) the window waits for
    long seq = 0;               // the event (tick) whose completion means the work queued for this state has run
    int cap[6] = {0, 0, 0, 0, 0, 0};  // the largest window its work space has held: poses, points, edges, LiDAR keyframes, cloud points, free poses
    EngineTicket* ticket = nullptr;
    std::vector<CopyTask> deferred_vis, deferred_lidar;
};
struct EngineStaging {  // pinned staging of one tick's launches (two sets, by tick parity: a set is rewritten when its tick's round has been waited for)
    PinnedBuf<BalmCutTask> h_cut_list;
    DevBuf<BalmCutTask> d_cut_list;
    PinnedBuf<CopyTask> h_copies_a, h_copies_b, h_copies_r;
};

}  // namespace

struct tc2li_ba_engine {
    tc2li_camera cam{};
    int capacity = 0, device = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<EngineTicket*> queue;                          // tickets with windows still to admit
    std::map<int64_t, std::unique_ptr<EngineTicket>> open;    // every ticket not yet collected by tc2li_ba_engine_wait
    int64_t next_id = 1;
    bool quit = false;
    int failed = 0;                                           // a HIP error in the engine thread: every window ends with TC2LI_ERR_HIP
    std::string error;
    std::thread th;
    BufferCache cache;                                        // declared before everything that owns buffers: destroyed after them
    // ---- engine thread only ----
    LockstepContext C;
    EngineStaging S[2];
    std::vector<EngineSlot> slots;
    std::vector<LockstepWindow> W;
    std::unique_ptr<WorkerPool> pool;
    long tick = 0;
    int busy = 0;                                             // slots not free
    // The plane extraction of newly admitted windows (their clouds up, six kernels of 0.3-0.5 ms together) sits between two rounds of
    // everything alive when it is queued in the main stream.  TC2LI_BA_ENGINE_SIDE=1 queues it on a SIDE stream of the engine instead (one event
    // per tick parity; a parity's staging buffers and event are reused only when its last launch has completed).  Measured, three engines,
    // frames/s main -> side: 512 sequences 20.3-20.6 k -> 19.7-20.3 k, 256: 16.8-17.2 k -> 17.8-18.5 k, 128: 14.8-15.2 k -> 14.4-14.6 k -- with
    // the side streams the LiDAR thread's step grows from 16-18 to 25 ms at 512 (the process's streams share four hardware queues, and the
    // long single-workgroup sort of the extraction then sits in front of another stage's kernels): off by default.
    hipStream_t side = nullptr;
    bool use_side = false;
    // TC2LI_BA_ENGINE_STAGE="min,wait": windows staged together / ticks one waits at most.  Measured, three engines, frames/s at "1,0" / "4,2" /
    // "8,3": 512 sequences 20.4-20.6 k / 20.5-20.7 k / 20.7-20.9 k; 256: 17.4-17.7 / 17.7-17.8 / 17.7-18.0; 128: 14.7-15.3 / 15.3-15.7 / 15.2-15.5
    int stage_min = 8, stage_wait = 3;
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    long side_last[2] = {0, 0};                               // the tick of the last launch recorded on side_ev[parity] (0: none)
    bool side_done(int par) { return side_last[par] == 0 || hipEventQuery(side_ev[par]) == hipSuccess; }
    ~tc2li_ba_engine() {
        for (hipEvent_t e : side_ev) if (e) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
    }
    // ---- the setup threads: a window's host-side setup runs beside the rounds of the others ----
    std::mutex smu;
    std::condition_variable scv;
    std::deque<int> setup_queue;                              // 2 * slot + (0: structure and staging, 1: LiDAR window)
    bool setup_quit = false;
    std::vector<std::thread> setup_threads;
    std::atomic<long> setup_us[2] = {{0}, {0}}, setup_n[2] = {{0}, {0}};   // (TC2LI_BA_TIMING) time inside the two kinds of setup task
    void run();
    void finish_window(int s, int rc);
    void setup_task(int task);
    void setup_loop();
};


void tc2li_ba_engine::finish_window(int s, int rc) {
    EngineSlot& sl = slots[s];
    EngineTicket* t = sl.ticket;
    {
        std::lock_guard<std::mutex> lk(mu);
        t->results[sl.index] = rc;
        if (rc >= 0) ++t->n_ok;
        if (--t->remaining == 0) cv_done.notify_all();
    }
    sl.state = kSlotFree; sl.ticket = nullptr; sl.deferred_vis.clear(); sl.deferred_lidar.clear();
    --busy;
}

void tc2li_ba_engine::setup_task(int task) {
    hipStream_t st = C.st;
    const int s = task >> 1;
    EngineSlot& sl = slots[s];
    LockstepWindow& w = W[s];
    const tc2li_ba_problem& p = sl.ticket->problems[sl.index];
    const bool args_ok = p.poses7 && p.fixed && p.points3 && p.edges && p.n_poses > 0 && p.n_points > 0 && p.n_edges > 0 && p.iterations >= 0;
    bool lidar_ok = true;
    if (args_ok && p.lidar) {
        if (p.lidar->n_keyframes < 1 || p.lidar->n_keyframes > 7 || !p.lidar->pose_index) lidar_ok = false;
        else for (int k = 0; k < p.lidar->n_keyframes; ++k) if (p.lidar->pose_index[k] < 0 || p.lidar->pose_index[k] >= p.n_poses) lidar_ok = false;
    }
    if (task & 1) {
        if (!args_ok || !lidar_ok || !p.lidar) return;
        CopySink sink(&sl.deferred_lidar);
        sl.rc_lidar = C.ws[s]->lidar.build(p.poses7, p.n_poses, p.lidar, st, &C.h_cut.p[s]);
        return;
    }
    CopySink sink(&sl.deferred_vis);
    w.p = &p; w.ws = C.ws[s].get();
    if (!args_ok) { set_error("tc2li_ba_engine: invalid window"); w.rc = TC2LI_ERR_INVALID; return; }
    if (!lidar_ok) { set_error("tc2li_ba_engine: a LiDAR window of 1 .. 7 keyframes with pose_index in range"); w.rc = TC2LI_ERR_INVALID; return; }
    if (p.stats) memset(p.stats, 0, sizeof(*p.stats));
    if (p.lidar_stats) memset(p.lidar_stats, 0, sizeof(*p.lidar_stats));
    if (p.lidar) {
        w.extra_used.assign(p.n_poses, 0);
        for (int k = 0; k < p.lidar->n_keyframes; ++k) w.extra_used[p.lidar->pose_index[k]] = 1;
    }
    w.rc = w.vp.setup(*w.ws, p.poses7, p.fixed, p.n_poses, p.points3, p.n_points, p.edges, p.n_edges, &cam, w.extra_used.empty() ? nullptr : w.extra_used.data(), st);
    if (w.rc < 0) return;
    BaWorkspace& ws = *w.ws;
    const size_t nn = (size_t)std::max(w.vp.np * w.vp.np, 1), n1 = (size_t)std::max(w.vp.np, 1), E = p.n_edges, P = p.n_points;
    bool ok = ws.d_S.ensure(nn) == hipSuccess && ws.d_bs.ensure(2 * n1) == hipSuccess && ws.d_xp.ensure(n1) == hipSuccess && ws.d_scal.ensure(8) == hipSuccess &&
              ws.h_result.ensure(p.n_poses * sizeof(Se3) + 3 * P * sizeof(double) + E * sizeof(double) + E) == hipSuccess;
    if (ok && p.lidar) {
        const size_t nl = 6 * (size_t)p.lidar->n_keyframes;
        ok = ws.d_Hl.ensure(nn + n1) == hipSuccess && ws.d_balm_out.ensure((size_t)balm_out_size(p.lidar->n_keyframes)) == hipSuccess && ws.d_lidar_JH.ensure(nl + nl * nl) == hipSuccess;
    }
    if (!ok) w.rc = TC2LI_ERR_HIP;
}
void tc2li_ba_engine::setup_loop() {
    (void)pthread_setname_np(pthread_self(), "tc2li-ba-setup");
    (void)hipSetDevice(device);
    BufferCacheScope cached(&cache);
    for (;;) {
        int task;
        {
            std::unique_lock<std::mutex> lk(smu);
            scv.wait(lk, [&] { return setup_quit || !setup_queue.empty(); });
            if (setup_queue.empty()) return;
            task = setup_queue.front(); setup_queue.pop_front();
        }
        const auto t0 = std::chrono::steady_clock::now();
        setup_task(task);
        setup_us[task & 1].fetch_add((long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
        setup_n[task & 1].fetch_add(1, std::memory_order_relaxed);
        slots[task >> 1].setup_left.fetch_sub(1, std::memory_order_release);
    }
}

void tc2li_ba_engine::run() {
    (void)pthread_setname_np(pthread_self(), "tc2li-ba-engine");
    (void)hipSetDevice(device);
    BufferCacheScope cached(&cache);
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&C.st, hipStreamNonBlocking, hi) != hipSuccess && hipStreamCreateWithFlags(&C.st, hipStreamNonBlocking) != hipSuccess) { C.st = nullptr; failed = 1; }
        for (hipEvent_t& e : C.round_done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; failed = 1; }
        use_side = getenv("TC2LI_BA_ENGINE_SIDE") && atoi(getenv("TC2LI_BA_ENGINE_SIDE")) != 0;
        if (const char* e = getenv("TC2LI_BA_ENGINE_STAGE")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 0) { stage_min = a; stage_wait = b; } }
        if (use_side && hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi) != hipSuccess && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; failed = 1; }
        for (hipEvent_t& e : side_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; failed = 1; }
    }
    hipStream_t st = C.st;
    const int cap = capacity;
    const size_t table_bytes = (size_t)cap * sizeof(BaBatchSlot);
    if (C.d_table.ensure(table_bytes) != hipSuccess || C.h_table.ensure(table_bytes) != hipSuccess || C.d_lm.ensure(cap) != hipSuccess || C.h_lm_init.ensure(cap) != hipSuccess ||
        C.h_lm.ensure(cap) != hipSuccess || C.h_stop.ensure(cap) != hipSuccess || C.h_cut.ensure(cap) != hipSuccess) failed = 1;
    BaBatchSlot* const h_slots = (BaBatchSlot*)C.h_table.p;
    const BaBatchSlot* const d_table = (const BaBatchSlot*)C.d_table.p;
    if (!failed) {
        memset(C.h_table.p, 0, table_bytes);
        for (int s = 0; s < cap; ++s) { C.h_lm.p[s] = BaLmState{}; C.h_stop.p[s] = 0; C.h_cut.p[s].n_points = 0; }
    }
    while ((int)C.ws.size() < cap) C.ws.emplace_back(new BaWorkspace());
    const bool kTiming = BaOptions::read().timing;
    double lat[6] = {0}; long n_lat = 0;
    double tm[8] = {0}; long n_live_sum = 0, n_windows = 0, n_ticks = 0; const long allocs0 = g_buffer_allocs.load();
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_lap = 0;
    auto lap = [&](int k) { if (kTiming) { const double t = now(); tm[k] += t - t_lap; t_lap = t; } };
    long recorded = 0, waited = 0;   // one event per tick (tick t records event t); `waited`: every event up to it has completed
    auto pieces_for = [&](const std::vector<int>& list, int expect, auto&& fn) { for_phase_pieces(d_table, (const double*)nullptr, W, list, fn, expect); };
    for (;;) {
        {   // ---- sleep while there is nothing to do ----
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return quit || !queue.empty() || busy > 0; });
            if (quit && queue.empty() && busy == 0) break;
        }
        ++tick;
        if (kTiming) { t_lap = now(); ++n_ticks; }
        EngineStaging& G = S[tick & 1];
        bool queued_any = false;
        int n_in_setup = 0, n_waiting_side = 0;
        // ---- 1. windows whose result copies have landed (the event of the tick that queued them has been waited for) ----
        {
            std::vector<int> done;
            for (int s = 0; s < cap; ++s) if (slots[s].state == kSlotRetiring && slots[s].seq <= waited) done.push_back(s);
            if (!done.empty()) {
                pool->parallel_for((int)done.size(), [&](int k) {
                    LockstepWindow& w = W[done[k]];
                    const tc2li_ba_problem& p = *w.p;
                    const size_t E = p.n_edges, P = p.n_points;
                    const uint8_t* h = w.ws->h_result.p;
                    memcpy(w.vp.poses.data(), h, p.n_poses * sizeof(Se3));
                    for (int q = 0; q < p.n_poses; ++q) { memcpy(p.poses7 + 7 * q, w.vp.poses[q].q, 4 * sizeof(double)); memcpy(p.poses7 + 7 * q + 4, w.vp.poses[q].t, 3 * sizeof(double)); }
                    memcpy(p.points3, h + p.n_poses * sizeof(Se3), 3 * P * sizeof(double));
                    const uint8_t* hc = h + p.n_poses * sizeof(Se3) + 3 * P * sizeof(double);
                    if (p.edge_chi2) memcpy(p.edge_chi2, hc, E * sizeof(double));
                    if (p.edge_depth_positive) memcpy(p.edge_depth_positive, hc + E * sizeof(double), E);
                });
                for (int s : done) {
                    LockstepWindow& w = W[s];
                    const tc2li_ba_problem& p = *w.p;
                    const BaLmState& m = C.h_lm.p[s];
                    if (p.stats) {
                        p.stats->iterations = m.done; p.stats->trials = m.trials_total; p.stats->n_free_poses = w.vp.n_free;
                        if (m.done > 0) { p.stats->initial_chi2 = m.initial_chi2; p.stats->final_chi2 = m.currentChi; p.stats->final_lambda = m.lambda; }
                    }
                    if (w.lidar && p.lidar_stats) {
                        p.lidar_stats->n_planes = w.lidar->n_planes; p.lidar_stats->hessian_evaluations = m.hessian_evaluations;
                        p.lidar_stats->residual = m.lidar_error; p.lidar_stats->chi2 = m.lidar_error * w.lidar->information * m.lidar_error;
                    }
                    if (kTiming) {
                        const EngineSlot& sl = slots[s]; const double t = now();
                        lat[0] += sl.t_admit - sl.t_submit; lat[1] += sl.t_ready - sl.t_admit; lat[2] += sl.t_staged - sl.t_ready; lat[3] += sl.t_live - sl.t_staged;
                        lat[4] += sl.t_retire - sl.t_live; lat[5] += t - sl.t_retire; ++n_lat;
                    }
                    finish_window(s, failed ? (int)TC2LI_ERR_HIP : m.done);
                }
            }
        }
        lap(0);
        if (failed) {   // nothing more is queued: every window that is still somewhere ends with the error
            for (int s = 0; s < cap; ++s) {
                if (slots[s].state == kSlotFree) continue;
                while (slots[s].state == kSlotSetup && slots[s].setup_left.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
                finish_window(s, (int)TC2LI_ERR_HIP);
            }
            std::unique_lock<std::mutex> lk(mu);
            while (!queue.empty()) {
                EngineTicket* t = queue.front(); queue.pop_front();
                for (; t->next < t->n; ++t->next) { t->results[t->next] = (int)TC2LI_ERR_HIP; --t->remaining; }
                cv_done.notify_all();
            }
            continue;
        }
        // ---- 2. staged windows whose plane extraction has run: table entry, state, uploads; alive from this tick's round on ----
        {
            std::vector<CopyTask> copies;
            size_t max_bytes = 0;
            for (int s = 0; s < cap; ++s) {
                EngineSlot& sl = slots[s];
                if (sl.state != kSlotStaged) continue;
                // (a window without planes to extract has nothing in flight: alive at the tick after its setup)
                if (sl.side_tick && side_last[sl.side_tick & 1] == sl.side_tick && hipEventQuery(side_ev[sl.side_tick & 1]) != hipSuccess) { ++n_waiting_side; continue; }
                LockstepWindow& w = W[s];
                if (w.rc >= 0 && w.p->lidar) {
                    if (sl.rc_lidar >= 0 && C.h_cut.p[s].n_points > 0) sl.rc_lidar = C.ws[s]->lidar.finish_cut(st);
                    if (sl.rc_lidar < 0) w.rc = sl.rc_lidar; else w.lidar = &C.ws[s]->lidar;
                }
                C.h_cut.p[s].n_points = 0;
                // outside the batched kernels: the one-window path, here (rare: more than kSolveMaxFree free keyframes, more than 2048 planes)
                if (w.rc >= 0 && (w.vp.n_free > kSolveMaxFree || (w.lidar && w.lidar->n_planes > 2048))) {
                    if (hipStreamSynchronize(st) != hipSuccess) { failed = 1; break; }
                    const tc2li_ba_problem& p = *w.p;
                    const int rc = tc2li_local_lv_bundle_adjustment(p.poses7, p.fixed, p.n_poses, p.points3, p.n_points, p.edges, p.n_edges, &cam, p.iterations, p.lambda_init,
                                                                    p.stop_flag, p.edge_chi2, p.edge_depth_positive, p.stats, p.lidar, p.lidar_stats, st);
                    finish_window(s, rc);
                    continue;
                }
                if (w.rc < 0) { finish_window(s, w.rc); continue; }
                // the slot: everything a phase leaves for the next one stays in device memory (ba_batch_lockstep's device-LM form)
                BaBatchSlot& b = h_slots[s];
                fill_device_lm_slot(b, w, C, s);
                BaLmState& m = C.h_lm_init.p[s];
                copies.push_back(CopyTask{C.d_table.p + (size_t)s * sizeof(BaBatchSlot), &b, sizeof(BaBatchSlot)});
                copies.push_back(CopyTask{C.d_lm.p + s, &m, sizeof(BaLmState)});
                for (const CopyTask& t : sl.deferred_vis) copies.push_back(t);
                sl.deferred_vis.clear();
                sl.state = kSlotLive; sl.seq = tick; if (kTiming) sl.t_live = now();
            }
            if (!failed && !copies.empty()) {
                if (G.h_copies_b.ensure(copies.size()) != hipSuccess) failed = 1;
                else {
                    for (size_t k = 0; k < copies.size(); ++k) { G.h_copies_b.p[k] = copies[k]; max_bytes = std::max(max_bytes, copies[k].bytes); }
                    launch_copy_tasks(G.h_copies_b.p, (int)copies.size(), max_bytes, st);
                    queued_any = true;
                }
            }
        }
        lap(1);
        // ---- 3. new windows from the queue into free slots: structure + staging on the pool, the plane extraction queued ----
        {
            std::vector<int> fresh, admitted;
            {
                std::lock_guard<std::mutex> lk(mu);
                // A slot's work space keeps its device and pinned buffers from window to window, and growing one is a device-wide synchronisation
                // (hipFree): a window goes to the free slot whose work space has held the SMALLEST window at least as large in every measure
                // (best fit); if none has, to the one that has held the largest (it grows, and there is one more large work space).
                int n_free_slots = 0;
                for (int s = 0; s < cap; ++s) n_free_slots += slots[s].state == kSlotFree;
                while (n_free_slots > 0 && !queue.empty()) {
                    EngineTicket* t = queue.front();
                    const tc2li_ba_problem& p = t->problems[t->next];
                    int dims[6] = {p.n_poses, p.n_points, p.n_edges, 0, 0, 0};
                    if (p.fixed) for (int q = 0; q < p.n_poses; ++q) dims[5] += p.fixed[q] == 0;
                    if (p.lidar && p.lidar->n_keyframes >= 1 && p.lidar->n_keyframes <= 20 && p.lidar->cloud_offsets) { dims[3] = p.lidar->n_keyframes; dims[4] = p.lidar->cloud_offsets[p.lidar->n_keyframes]; }
                    int best = -1, largest = -1;
                    for (int s = 0; s < cap; ++s) {
                        if (slots[s].state != kSlotFree) continue;
                        const int* c = slots[s].cap;
                        if (c[0] >= dims[0] && c[1] >= dims[1] && c[2] >= dims[2] && c[3] >= dims[3] && c[4] >= dims[4] && c[5] >= dims[5] && (best < 0 || c[2] < slots[best].cap[2])) best = s;
                        if (largest < 0 || c[2] > slots[largest].cap[2]) largest = s;
                    }
                    const int s = best >= 0 ? best : largest;
                    for (int k = 0; k < 6; ++k) slots[s].cap[k] = std::max(slots[s].cap[k], dims[k]);
                    slots[s].ticket = t; slots[s].index = t->next++; slots[s].state = kSlotSetup; slots[s].rc_lidar = 0; slots[s].ready_tick = 0;
                    if (kTiming) { slots[s].t_submit = t->t_submit; slots[s].t_admit = now(); slots[s].t_ready = 0; }
                    if (t->next == t->n) queue.pop_front();
                    admitted.push_back(s);
                    ++busy; --n_free_slots;
                }
            }
            if (!admitted.empty()) {   // their setup: on the setup threads, while this thread goes on with the rounds
                for (int s : admitted) { W[s] = LockstepWindow{}; C.h_cut.p[s].n_points = 0; slots[s].setup_left.store(2, std::memory_order_relaxed); }
                { std::lock_guard<std::mutex> lk(smu); for (int s : admitted) { setup_queue.push_back(2 * s); setup_queue.push_back(2 * s + 1); } }
                scv.notify_all();
            }
            // windows whose setup has finished: staged from this tick on -- unless this parity's staging buffers still serve a plane extraction
            // that has not run (then at the next tick)
            // (and, with LiDAR windows among them, only when a few have gathered or one has waited: the extraction is six launches of 0.3-0.5 ms
            // together in front of the next round whether it serves one window or ten)
            const int par = (int)(tick & 1);
            if (side_done(par)) {
                int n_ready = 0, n_ready_lidar = 0;
                long oldest = tick;
                for (int s = 0; s < cap; ++s)
                    if (slots[s].state == kSlotSetup && slots[s].setup_left.load(std::memory_order_acquire) == 0) {
                        if (!slots[s].ready_tick) { slots[s].ready_tick = tick; if (kTiming) slots[s].t_ready = now(); }
                        ++n_ready; n_ready_lidar += C.h_cut.p[s].n_points > 0; oldest = std::min(oldest, slots[s].ready_tick);
                    }
                if (n_ready && (n_ready_lidar == 0 || n_ready >= stage_min || tick - oldest >= stage_wait))
                    for (int s = 0; s < cap; ++s)
                        if (slots[s].state == kSlotSetup && slots[s].ready_tick) { slots[s].state = kSlotStaged; slots[s].seq = tick; slots[s].side_tick = 0; slots[s].ready_tick = 0; if (kTiming) slots[s].t_staged = now(); fresh.push_back(s); }
            }
            n_in_setup = 0;
            for (int s = 0; s < cap; ++s) n_in_setup += slots[s].state == kSlotSetup;
            if (!fresh.empty()) {
                // the extraction of the staged windows' planes: their clouds up, the cut kernels (as plane_extraction_begin), on the side stream
                int m = 0, max_points = 0, max_table = 0;
                size_t n_copies = 1;
                for (int s : fresh) n_copies += slots[s].deferred_lidar.size();
                if (G.h_cut_list.ensure(fresh.size()) != hipSuccess || G.d_cut_list.ensure(fresh.size()) != hipSuccess || G.h_copies_a.ensure(n_copies) != hipSuccess) failed = 1;
                if (!failed) {
                    for (int s : fresh) {
                        const BalmCutTask& t = C.h_cut.p[s];
                        if (t.n_points <= 0 || slots[s].rc_lidar < 0) continue;
                        G.h_cut_list.p[m++] = t;
                        max_points = std::max(max_points, t.n_points); max_table = std::max(max_table, 1 << t.table_bits);
                    }
                    size_t at = 0, max_bytes = (size_t)std::max(m, 1) * sizeof(BalmCutTask);
                    if (m) G.h_copies_a.p[at++] = CopyTask{G.d_cut_list.p, G.h_cut_list.p, (size_t)m * sizeof(BalmCutTask)};
                    for (int s : fresh) {
                        if (!slots[s].deferred_lidar.empty() || (C.h_cut.p[s].n_points > 0 && slots[s].rc_lidar >= 0)) slots[s].side_tick = tick;
                        for (const CopyTask& t : slots[s].deferred_lidar) { G.h_copies_a.p[at++] = t; max_bytes = std::max(max_bytes, t.bytes); }
                        slots[s].deferred_lidar.clear();
                    }
                    hipStream_t cut_st = use_side ? side : st;
                    if (at) launch_copy_tasks(G.h_copies_a.p, (int)at, max_bytes, cut_st);
                    if (m) launch_balm_cut(G.d_cut_list.p, m, max_points, max_table, cut_st);
                    if (at || m) {
                        if (hipGetLastError() != hipSuccess || hipEventRecord(side_ev[par], cut_st) != hipSuccess) failed = 1;
                        side_last[par] = tick;
                    }
                }
            }
        }
        lap(2);
        // ---- 4. retire what the last mirror shows as done; the caller's stop flags; 5. the next round for everything alive ----
        std::vector<int> live, live_lidar, lidar_first, retire;
        bool want_maxdiag = false;
        for (int s = 0; s < cap && !failed; ++s) {
            if (slots[s].state != kSlotLive) continue;
            const BaLmState& m = C.h_lm.p[s];
            if (m.status == kLmDone) { retire.push_back(s); continue; }
            if (W[s].stopped()) C.h_stop.p[s] = 1;
            live.push_back(s);
            if (W[s].lidar) { live_lidar.push_back(s); if (m.it == 0) lidar_first.push_back(s); }
            want_maxdiag |= m.it == 0 && !(W[s].p->lambda_init > 0);
        }
        if (!failed && !retire.empty()) {
            const BaBatchExtent XR = batch_extent(W, retire);
            pieces_for(retire, 0, [&](const BaPhase& ph, int cnt) { ba_batch_launch_depth(ph, cnt, XR, st); });
            if (G.h_copies_r.ensure(4 * retire.size()) != hipSuccess) failed = 1;
            size_t n_tasks = 0, max_bytes = 0;
            for (int s : retire) {
                if (failed) break;
                LockstepWindow& w = W[s];
                const tc2li_ba_problem& p = *w.p;
                const BaLmState& m = C.h_lm.p[s];
                BaProblemDev pb = w.vp.pb;
                if (m.parity) { std::swap(pb.poses, pb.poses_trial); std::swap(pb.points, pb.points_trial); }
                const size_t E = p.n_edges, P = p.n_points;
                uint8_t* h = w.ws->h_result.p;
                uint8_t* hc = h + p.n_poses * sizeof(Se3) + 3 * P * sizeof(double);
                auto add = [&](void* dst, const void* src, size_t nbytes) { G.h_copies_r.p[n_tasks++] = CopyTask{dst, src, nbytes}; max_bytes = std::max(max_bytes, nbytes); };
                add(h, pb.poses, p.n_poses * sizeof(Se3));
                add(h + p.n_poses * sizeof(Se3), pb.points, 3 * P * sizeof(double));
                if (p.edge_chi2) add(hc, w.ws->d_chi2.p, E * sizeof(double));
                if (p.edge_depth_positive) add(hc + E * sizeof(double), w.ws->d_depth.p, E);
                slots[s].state = kSlotRetiring; slots[s].seq = tick; if (kTiming) slots[s].t_retire = now();
            }
            if (!failed && n_tasks) { launch_copy_tasks(G.h_copies_r.p, (int)n_tasks, max_bytes, st); queued_any = true; }
        }
        lap(3);
        n_live_sum += (long)live.size(); n_windows += (long)retire.size();
        if (!failed && !live.empty()) {
            bool all_block = true;
            BaBatchExtent XL = batch_extent(W, live, &all_block);
            XL.fuse_trial = all_block ? 1 : 0;
            queue_lm_round(d_table, W, live, live_lidar, lidar_first, XL, want_maxdiag, st);
            queued_any = true;
        }
        lap(4);
        // ---- 6. this tick's event; wait for the tick BEFORE it (one tick's work stays in flight while the host prepares the next) ----
        if (!failed) {
            if (hipGetLastError() != hipSuccess || hipEventRecord(C.round_done[tick & 1], st) != hipSuccess) failed = 1;
            else recorded = tick;
            auto wait_one = [&] { if (event_wait_sleeping(C.round_done[(waited + 1) & 1]) != hipSuccess) failed = 1; ++waited; };
            if (!failed && recorded - waited > 1) wait_one();
            // nothing alive: what is in flight (plane extractions, result copies) is all there is to wait for -- without this the loop would
            // run through empty ticks
            while (!failed && live.empty() && waited < recorded) wait_one();
            if (!failed && live.empty() && (n_in_setup > 0 || n_waiting_side > 0) && !queued_any) std::this_thread::sleep_for(std::chrono::microseconds(50));  // only setups / plane extractions are running
        }
        lap(5);
        if (failed) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk(mu); if (error.empty()) error = "HIP error in the engine thread"; }
    }
    if (kTiming && n_lat) fprintf(stderr, "BA engine window latency ms (%ld windows): queue %.3f setup %.3f wait-to-stage %.3f extraction %.3f rounds %.3f results %.3f\n",
                                  n_lat, lat[0] / n_lat, lat[1] / n_lat, lat[2] / n_lat, lat[3] / n_lat, lat[4] / n_lat, lat[5] / n_lat);
    if (kTiming && setup_n[0].load()) fprintf(stderr, "BA engine setup tasks: structure + staging %.3f ms each (%ld), LiDAR window %.3f ms each (%ld)\n",
                                               1e-3 * setup_us[0].load() / setup_n[0].load(), setup_n[0].load(), 1e-3 * setup_us[1].load() / std::max(setup_n[1].load(), 1L), setup_n[1].load());
    if (kTiming && n_ticks) fprintf(stderr, "BA engine timing: %ld windows in %ld ticks (%.1f alive per tick); ms per tick: finish %.3f go-live %.3f admit+setup %.3f retire %.3f queue round %.3f wait %.3f; %ld buffer (re)allocations in the process meanwhile\n",
                                    n_windows, n_ticks, (double)n_live_sum / n_ticks, tm[0] / n_ticks, tm[1] / n_ticks, tm[2] / n_ticks, tm[3] / n_ticks, tm[4] / n_ticks, tm[5] / n_ticks, g_buffer_allocs.load() - allocs0);
    if (C.st) (void)hipStreamSynchronize(C.st);
    if (side) (void)hipStreamSynchronize(side);
}

extern "C" {

// ---- the engine's entry points (include/tc2li_hip.h) ----
int tc2li_ba_engine_create(const tc2li_camera* cam, int max_windows, tc2li_ba_engine** out) {
    if (!cam || !out || max_windows < 1 || max_windows > 4096) { set_error("tc2li_ba_engine_create: invalid argument (1 .. 4096 windows in flight)"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    std::unique_ptr<tc2li_ba_engine> e(new tc2li_ba_engine());
    e->cam = *cam; e->capacity = max_windows;
    if (hipGetDevice(&e->device) != hipSuccess) { (void)hipGetLastError(); e->device = 0; }
    e->slots = std::vector<EngineSlot>(max_windows); e->W.resize(max_windows);
    // host threads: what the three lock-step groups of the batch calls have between them -- two thirds for the setups, a third for the results
    const int per_group = std::max(1, pool_threads(kPoolBaGroup0));
    e->pool.reset(new WorkerPool(per_group, "tc2li-ba-result"));
    tc2li_ba_engine* raw = e.get();
    for (int k = 0; k < 2 * per_group; ++k) e->setup_threads.emplace_back([raw] { raw->setup_loop(); });
    e->th = std::thread([raw] { raw->run(); });
    *out = e.release();
    return TC2LI_OK;
}

void tc2li_ba_engine_destroy(tc2li_ba_engine* e) {
    if (!e) return;
    { std::lock_guard<std::mutex> lk(e->mu); e->quit = true; }
    e->cv_work.notify_all();
    if (e->th.joinable()) e->th.join();
    { std::lock_guard<std::mutex> lk(e->smu); e->setup_quit = true; }
    e->scv.notify_all();
    for (std::thread& t : e->setup_threads) if (t.joinable()) t.join();
    delete e;
}

int64_t tc2li_ba_engine_submit(tc2li_ba_engine* e, const tc2li_ba_problem* problems, int n, int32_t* results) {
    if (!e || n < 0 || (n > 0 && (!problems || !results))) { set_error("tc2li_ba_engine_submit: invalid argument"); return TC2LI_ERR_INVALID; }
    std::unique_ptr<EngineTicket> t(new EngineTicket());
    t->problems = problems; t->results = results; t->n = n; t->remaining = n;
    t->t_submit = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    int64_t id;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->quit) { set_error("tc2li_ba_engine_submit: the engine is shutting down"); return TC2LI_ERR_INVALID; }
        id = t->id = e->next_id++;
        if (n > 0) e->queue.push_back(t.get());
        e->open[id] = std::move(t);
    }
    e->cv_work.notify_all();
    return id;
}

int tc2li_ba_engine_poll(tc2li_ba_engine* e, int64_t ticket) {
    if (!e) { set_error("tc2li_ba_engine_poll: invalid argument"); return TC2LI_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(e->mu);
    auto it = e->open.find(ticket);
    if (it == e->open.end()) { set_error("tc2li_ba_engine_poll: no such ticket (collected already?)"); return TC2LI_ERR_INVALID; }
    return it->second->remaining == 0 ? 1 : 0;
}

int tc2li_ba_engine_wait(tc2li_ba_engine* e, int64_t ticket) {
    if (!e) { set_error("tc2li_ba_engine_wait: invalid argument"); return TC2LI_ERR_INVALID; }
    std::unique_lock<std::mutex> lk(e->mu);
    auto it = e->open.find(ticket);
    if (it == e->open.end()) { set_error("tc2li_ba_engine_wait: no such ticket (collected already?)"); return TC2LI_ERR_INVALID; }
    EngineTicket* t = it->second.get();
    e->cv_done.wait(lk, [&] { return t->remaining == 0; });
    const int ok = t->n_ok;
    const bool failed = e->failed != 0;
    const std::string why = e->error;
    e->open.erase(it);
    lk.unlock();
    if (failed && ok == 0 && !why.empty()) set_error("tc2li_ba_engine: %s", why.c_str());
    return ok;
}

}  // extern "C"
