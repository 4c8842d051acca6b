// Shared between the host orchestration and the bundle-adjustment kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_math.hpp"
#include "global_ptr.hpp"
#include "balm_device.hpp"
#include "inertial_math.hpp"

namespace tc2li {

// ---- tasks of the block-by-block sparse Schur product (k_ba_schur_blocks, ba_kernels.hip) ----
// A window's slices (<= 256 slots of <= 64 landmarks) are multiplied kSchurGroup at a time by one workgroup (a PART) of 256 threads,
// thread = task: the 6x6 block (i, j) of free poses over one of R equal ranges of the slice's landmark ranks (diagonal blocks: tasks
// Rd i + q; pairs i > j: tasks Rd nf + Ro (i (i - 1) / 2 + j) + q), then one coefficient-row task per free pose.  The diagonal block of
// pose i gets a product from every slot of pose i, a pair block only from the landmarks both poses see; (Rd, Ro) is the finest cut
// whose tasks fit the workgroup.
constexpr int kSchurGroup = 4;
// The LEAN form of the same product (k_ba_schur_lean, pb.schur_blocks == 2): slices of at most kSchurLeanSlots slots, kSchurGroupLean of
// them per part -- half the LDS and two thirds of the registers per workgroup, so that beside the other stages' kernels a workgroup
// finds its place sooner (ba_kernels.hip)
constexpr int kSchurLeanSlots = 128, kSchurGroupLean = 8;
constexpr int kSchurOps = 38;  // doubles per slot in LDS: Y row-major [6][3] | W row-major [6][3] | 2 of padding (16-byte aligned blocks that spread over all banks)
constexpr int kSolveMaxFree = 24;        // widest reduced system k_ba_solve_b factorises (144 unknowns: nine 16 x 16 tile rows in a workgroup's registers)
constexpr int kSchurBlocksMaxFree = 21;  // every window of the sparse path (np_pad / 16 <= 8): 21 * 20 / 2 + 21 + 21 = 252 tasks
__host__ __device__ constexpr int schur_tasks_for(int nf, int rd, int ro) { return rd * nf + ro * (nf * (nf - 1) / 2) + nf; }
__host__ __device__ constexpr void schur_ranges(int nf, int& rd, int& ro) {
    // measured at 12 free keyframes (43 windows, us per launch): (5,2) 65, (4,2) 71, (3,2) 70, (3,3) 76, (2,1) 78, (1,1) 84
    const int pref[7][2] = {{5, 2}, {4, 2}, {3, 2}, {2, 2}, {3, 1}, {2, 1}, {1, 1}};
    for (int k = 0; k < 7; ++k) { rd = pref[k][0]; ro = pref[k][1]; if (schur_tasks_for(nf, rd, ro) <= 256) return; }
}
// The WIDE lean form (round 6): windows of 22 .. kSchurLeanMaxFree free keyframes -- one side of the old sparse / dense boundary, where the
// dense path's full-width MFMA kernel (139 KB of LDS per workgroup: it waits for whole CUs beside the other stages) cost 15 of the mixed loop's
// 108 ms of kernel time per step for an eighth of the windows -- run the same lean product with TWO workgroups per part, each half of the up
// to 512 tasks (24 free keyframes are 24 diagonal + 276 pair + 24 coefficient tasks), in the launch of the narrower windows.  The finest cut
// of the landmark ranges whose tasks fit 512 threads, whose diagonal blocks' range tasks stay in the first half and whose closing sums fit
// the operand area.
constexpr int kSchurLeanMaxFree = 24;
__host__ __device__ constexpr void schur_ranges_wide(int nf, int& rd, int& ro) {
    const int pref[4][2] = {{4, 1}, {3, 1}, {2, 1}, {1, 1}};
    for (int k = 0; k < 4; ++k) { rd = pref[k][0]; ro = pref[k][1]; if (schur_tasks_for(nf, rd, ro) <= 512) return; }
}
constexpr bool schur_lean_wide_fits() {
    for (int nf = kSchurBlocksMaxFree + 1; nf <= kSchurLeanMaxFree; ++nf) {
        int rd = 1, ro = 1;
        schur_ranges_wide(nf, rd, ro);
        if (schur_tasks_for(nf, rd, ro) > 512 || 36 * ((rd > 1 ? nf : 0) + (ro > 1 ? nf * (nf - 1) / 2 : 0)) > kSchurLeanSlots * kSchurOps || nf > 255 ||
            rd * nf > 256 || ro != 1) return false;  // (a block's range tasks meet in their workgroup's LDS: the diagonal ones in the first half, pairs uncut)
    }
    return true;
}
static_assert(schur_lean_wide_fits(), "k_ba_schur_lean_wide: tasks or closing sums of some window size do not fit the workgroup");
__host__ __device__ inline int schur_task_count(int nf) { int rd = 1, ro = 1; schur_ranges(nf, rd, ro); return schur_tasks_for(nf, rd, ro); }
// landmark ranks [64 q / R, 64 (q + 1) / R) as a bit mask
__host__ __device__ inline unsigned long long schur_range_mask(int R, int q) {
    const int lo = 64 * q / R, hi = 64 * (q + 1) / R;
    const unsigned long long upto_hi = hi >= 64 ? ~0ull : (1ull << hi) - 1, upto_lo = (1ull << lo) - 1;
    return upto_hi & ~upto_lo;
}

// The lean form adds a block's range tasks up through the operand area of its LDS (36 doubles per block of more than one range): every
// window of the block-by-block path must fit (a change of the preference table above that breaks this fails the build)
__host__ __device__ constexpr int schur_lean_closing_blocks(int nf) {
    int rd = 1, ro = 1;
    schur_ranges(nf, rd, ro);
    return (rd > 1 ? nf : 0) + (ro > 1 ? nf * (nf - 1) / 2 : 0);
}
constexpr bool schur_lean_fits_all() {
    for (int nf = 1; nf <= kSchurBlocksMaxFree; ++nf) {
        int rd = 1, ro = 1;
        schur_ranges(nf, rd, ro);
        if (schur_tasks_for(nf, rd, ro) > 256 || 36 * schur_lean_closing_blocks(nf) > kSchurLeanSlots * kSchurOps) return false;
    }
    return true;
}
static_assert(schur_lean_fits_all(), "k_ba_schur_lean: tasks or closing sums of some window size do not fit the workgroup");

// One local-BA problem resident on the device.  Free (non-fixed) poses are numbered 0..n_free-1 through pose_var;
// pt_* is a CSR of the edges of each landmark, pv_* a CSR of the edges of each free pose (n_free_edges in total).
// Per-edge records of the linearisation are padded to whole 16-byte pieces (9 -> 10 and 27 -> 28 doubles) so that they are written
// and read as double2: the kernels' many pointers may alias as far as the compiler knows, and scalar stores stay scalar.
constexpr int kContribP = 28;
constexpr int kUnitChunkHost = 16;     // landmarks per chunk of the dense windows' block-sparse MFMA product (ba_kernels.hip: kUnitChunk)
constexpr int kBacksubMaxNp = 512;     // doubles of a step x_p the trial kernels stage in LDS
constexpr int kTrialPoseBytes = 12288; // LDS of the fused trial launch for the window's trial poses (219 SE3 vertices / 61 ImuCamPose records)
struct BaProblemDev {
    int32_t n_edges, n_points, n_poses, n_free, n_free_edges, np_pad;
    int32_t n_groups;         // workgroups of the linearisation: whole landmarks, at most 256 edges each (grp_k0 into pt_edges, grp_l0)
    int32_t n_schur_slices;   // slices of the sparse Schur product (slice_off has one more entry)
    CameraD cam;
    double delta_mono, delta_stereo;
    float dsqr_mono, dsqr_stereo;
    Se3 *poses, *poses_trial;
    // visual-inertial mode (Optimizer::LocalInertialBA): keyframe poses as ImuCamPose instead of SE3Quat
    int32_t inertial, schur_group;  // schur_group: slices per part of the block-by-block product (kSchurGroup; the lean form: kSchurGroupLean unless set)
    ImuPose *iposes, *iposes_trial;
    ImuCalib calib;
    double *points, *points_trial;
    const BaEdge* edges;
    const int32_t *pose_var, *pt_off, *pt_edges, *pv_off, *pv_edges, *grp_k0, *grp_l0;
    // the edges with a free pose, landmark-major ("slots"; the landmarks in the order of the poses they are seen from, VisualProblem::setup in ba_internal.hpp):
    // fl_off[2 l], fl_off[2 l + 1] = begin / end of landmark l's slots, fl_pose the free pose of each slot
    // (-1: fixed pose; dense path only).  A landmark has at most one edge per pose.
    const int32_t *fl_off, *fl_pose, *fl_lm, *fl_place, *slice_off, *fl_edge;  // fl_edge: the edge of each
    const uint32_t* chunk_mask;  // dense path: per chunk of 16 landmarks, the 16-column tiles of the reduced system its landmarks touch
    // sparse_schur: the Schur complement is formed from the landmark-major W blocks (k_ba_schur_sparse; np_pad / 16 <= 8 tile rows),
    // one partial sum per slice; otherwise through the dense k-major operands AT / BT
    // schur_blocks: the sparse product block by block on the f64 vector unit (k_ba_schur_blocks, one partial per kSchurGroup slices)
    // instead of the zero-padded MFMA form (k_ba_schur_sparse4/9, one partial per slice; TC2LI_BA_SCHUR_MFMA=1)
    int32_t sparse_schur, schur_blocks;  // schur_blocks: 0 MFMA form of the sparse product, 1 block by block, 2 block by block, lean form
    int32_t schur_rd, schur_ro;  // landmark ranges per diagonal / off-diagonal block task (schur_ranges)
    // trial_fused: the trial estimate, its cost and the closing sums run as ONE launch over the linearisation's landmark groups (round 5:
    // k_ba_trial_fused*; a property of the window -- it decides the order of two sums -- set by the setup when the window's step, poses and
    // groups fit the kernel's LDS: kBacksubMaxNp, kTrialPoseBytes, 256 landmarks per group)
    int32_t trial_fused, n_dups;
    // duplicates: edges between a (point, free pose) pair that has an edge already.  They own no slot; k_ba_dups, after the linearisation's
    // sums, adds a duplicate's W block to the slot of the pair's first edge (dup_slot) and its pose block to the pose's Hpp / b_p -- per free
    // pose its duplicates in edge order (dup_off [n_free + 1] into dup_edge / dup_slot)
    const int32_t *dup_off, *dup_edge, *dup_slot;
    double *chi2, *rho0;
    double *cp_part, *W;                 // per (block of 256 free-pose edges, free pose): 27 (+1) doubles; per free-pose edge (at w_slot): 18
    const int32_t* blk_off;              // per block: n_free + 1 offsets into its rows sorted by pose
    const uint8_t* blk_rows;             // per block: 256 row numbers
    double *Hll, *bl, *diag_l;           // per landmark: 6, 3, 1
    double *Hpp, *diag_p;                // per free pose: 27 (21 packed upper + 6 b), 1
    double *coef_e, *coef;               // per edge 6, per free pose 6
    double *Y;                           // dense windows: W D^-1 per slot (18), written by k_ba_schur_coef for the trial's lambda
    double *S_part;                      // [n_slices][np_pad * np_pad]; sparse path: row 6 n_free holds W D^-1 b_l
    double *scale_part;                  // per 256 landmarks: partial sums of the gain-ratio scale
    int32_t* ticket;                     // [4] zero between launches: how many workgroups of a window have delivered their partial sums (lock-step batch:
                                         // the last one adds them up -- [1] trial errors)
    double *chi_part;                    // partial sums of the robust cost: per linearisation group / per 256 edges of a trial
};

// chi_out[0] = robust cost; maxdiag_out[0..1] = largest |diagonal| of the landmark / pose blocks when want_maxdiag
void ba_launch_linearize(const BaProblemDev& pb, double* chi_out, double* maxdiag_out, bool want_maxdiag, hipStream_t st);
// number of partial sums the sparse Schur product of a window with `n_slices` slices leaves in S_part (what `n_slices` means to
// ba_launch_schur / BaBatchSlot for such a window)
int ba_schur_parts(int n_slices, int group);  // group: slices per partial sum (pb.schur_group; 1: the MFMA form)
// S_out [np*np], bs_out [2*np]: b_s followed by b_p
// lambda_pose: what the finish kernel adds to the diagonal of S (lambda; 0 on the ranks > 0 of a sharded window, whose parts are summed)
void ba_launch_schur(const BaProblemDev& pb, double lambda, double lambda_pose, int n_slices, int k_per_slice, double* S_out, double* bs_out, hipStream_t st);
void ba_launch_trial(const BaProblemDev& pb, const double* xp, double lambda, double* scale_out, double* chi_out, hipStream_t st);
void ba_launch_depth(const BaProblemDev& pb, uint8_t* depth_pos, hipStream_t st);

// ---- lock-step batch (tc2li_local_bundle_adjustment_batch): one launch per phase for all windows ----
// A slot describes one window for the batched kernels.  Since round 4 the table goes up ONCE per call, with the windows' input blocks
// (rounds 2-3 rewrote it before every phase: 30 one-entry copy launches per call, in configs[3] the largest line of the kernel table);
// what changes from phase to phase travels in the kernels' arguments (BaPhase).
// The reduced system of an inertial window (LocalInertialBA / LocalLVIBA) for the solve on the device (k_lvi_solve*): reduced_solve.hpp's order --
// the 9 velocity / bias unknowns per keyframe first (a band: an inertial edge joins consecutive keyframes), the pose rows after them -- and its
// envelope.  The host leaves the inertial + LiDAR part of the matrix row by row inside the envelope once per linearisation; the kernel adds the
// visual Schur complement S (pose block) and lambda (velocity / bias diagonal) per trial.  n == 0: the window's system is solved on the host.
constexpr int kLviBand = 28;          // widest velocity / bias row (i - first[i]) the kernel's rings of 32 columns hold with four columns eliminated per step (17 for consecutive keyframes, 26 with one keyframe between)
constexpr int kLviMaxPoseRows = 150;  // 25 free keyframes (LocalInertialBA's maxOpt with bLarge): what the packed pose block + rings take of a CU's LDS
struct LviSolveDev {
    int32_t n, np, ni, pad_;          // unknowns; pose unknowns (the caller's first np); velocity / bias unknowns
    const int32_t* first;             // [n] solver order: column of row i's first entry (the envelope)
    const int32_t* span_end;          // [np] pose row r has its entries against the velocity / bias unknowns in columns first[ni + r] .. span_end[r] - 1
    const int32_t* rowoff;            // [np + 1] pose row r: the span's entries, then its r + 1 entries of the pose block, at hpose + rowoff[r]
    const double* hpose;
    const double* hband;              // [ni][32] the velocity / bias rows at a fixed width: entry (i, c) at 32 i + (c - (i - 31)), zero outside the envelope
    const double* bi;                 // [n] the right-hand side of the inertial + LiDAR part, the caller's numbering
    double *LB, *Lband;               // scratch: L of the pose rows per velocity / bias column [ni][np]; L of the band [ni][32] (entry (i, c) at [c][i - c - 1])
};
// ---- The Levenberg-Marquardt loop of a lock-step window ON THE DEVICE (round 6) ----
// g2o runs solve() as one sequential loop per window (optimization_algorithm_levenberg.cpp:61-169); rounds 2-5 ran its control flow on the host,
// one round trip per phase.  With a BaLmState per window in device memory the decisions are taken by two one-wavefront kernels:
//   k_ba_lm_begin_b   after a linearisation: the robust cost, the LiDAR edge's state machine (EdgeLidarSE3: r1 / r2 / is_calc_hess) and the change
//                     of variables of its Jacobian / Hessian (LidarCovisRes::ComputeJandHSE3) into the window's (6K)^2 block, computeLambdaInit;
//   k_ba_lm_decide_b  after a trial: computeScale's pose part, the gain ratio, accept / reject, the damping update, the stop rules of
//                     OptimizableGraph / the reference's `_nBad >= 3`, and what the window needs next (status).
// Every batched kernel reads lambda and the parity of the window's two estimate buffers from the state and runs only for a window whose status
// is the one its phase expects (BaPhase::expect), so the host can queue rounds AHEAD of the device -- a round = [linearisation set (windows in
// kLmIterate) | trial set (windows in kLmTrial)] -- and reads one status word per window and round from a pinned mirror, where it also polls
// the caller's stop flag (g2o's terminate()).  The arithmetic is the host loop's, operation for operation (lm_lambda_accepted, the sums' order):
// a window gives the same bits here, in the host-driven lock-step loop and in the one-window entry points.
constexpr int32_t kLmDone = 0, kLmIterate = 1, kLmTrial = 2;
struct BaLmState {
    double lambda, ni, currentChi, tempChi, iniChi, rho, initial_chi2;
    double lidar_error, r1, r2;        // EdgeLidarSE3: error, the last two residuals (SF/include/G2oTypesWithLidar.h:130-139)
    int32_t status, parity;            // parity 1: the accepted estimate lives in the slot's trial buffers
    int32_t qmax, trials_total, n_bad, it, done, ok;
    int32_t is_calc_hess, hessian_evaluations;
    int32_t solve_ok, rounds;          // solve_ok: k_ba_solve_b's pivots were usable; rounds: decide kernels that ran for this window
};
static_assert(sizeof(BaLmState) == 128, "BaLmState is mirrored to the host as sixteen 8-byte words");

struct BaBatchSlot {
    BaProblemDev pb;       // poses / points = the buffers the call starts from (BaPhase's parity bit swaps them with the trial buffers)
    int32_t n_slices, k_per_slice, has_lidar, pad_;
    double *chi_out, *maxdiag_out, *S_out, *bs_out, *scale_out, *chi_trial_out;
    double* hpp_out;  // where the host wants Hpp / b_p when the phase asks for them (pinned; BaPhase flag)
    ImuPose* iposes_host;  // inertial windows: the trial ImuCamPose states for the host's inertial cost (pinned; written by the trial kernel)
    // the reduced system solved on the device (k_ba_solve_b; S_out / bs_out are device buffers then): b_p for the host's gain-ratio scale,
    // the LiDAR term's (6K)^2 Hessian and 6K gradient to add (NULL: none), the step for the trial kernels (= xp) and for the host, and
    // whether the LDL^T went through
    double* bp_host;
    const double *Hl, *bl_lidar;
    double *x_dev, *x_host;
    int32_t* ok_host;
    LviSolveDev lvi;       // inertial windows solved on the device (k_lvi_solve_b): S_out / bs_out are device buffers, x_dev / x_host take all n unknowns
    const double* xp;      // the step when the phase's staging area does not hold it (a window of more than kBaXpStride / 6 free poses; device solve)
    uint8_t* depth_out;
    BalmDev balm;
    // device-side LM (NULL: lambda / parity come with the phase, the host decides)
    BaLmState* lm;            // device
    BaLmState* lm_host;       // pinned mirror, written by k_ba_lm_decide_b
    const int32_t* stop_host; // pinned: nonzero once the host has seen the caller's stop flag
    double* lidar_JH;         // [6W | (6W)^2]: the LiDAR edge's Jacobian / Hessian in the vertices' increments, kept while is_calc_hess is false
    double lambda_init, lidar_information;
    int32_t iterations, lm_pad_;
};
// The windows of one launch and their state in this phase, passed BY VALUE in the kernel arguments (HIP gives a kernel 4 KB of them):
// workgroups of window position y = blockIdx.y (z for the tiled GEMM) work on table[win[y]].
constexpr int kBaPhaseMax = 192;   // windows per launch; a longer phase list is launched in pieces
constexpr int kBaXpStride = 192;   // doubles per window in the staging area of the steps x_p: 6 x 32 free keyframes
struct BaPhase {
    const BaBatchSlot* table;      // the call's table in device memory
    const double* xp_area;         // steps of a trial phase, window position y at (first + y) * kBaXpStride; NULL: slot.xp
    int32_t first, pad_;           // position of win[0] in the phase's list
    int32_t expect, pad2_;         // device-side LM: the kernels of this launch run for windows whose BaLmState::status equals `expect` (0: for all)
    uint16_t win[kBaPhaseMax];
    uint8_t flags[kBaPhaseMax];    // kBaAcceptedInTrial: the accepted estimate lives in the trial buffers (an odd number of accepted steps);
                                   // kBaWantMaxdiag: computeLambdaInit's diagonal maxima; kBaWantHpp: Hpp / b_p to slot.hpp_out
    double lambda[kBaPhaseMax];
};
constexpr unsigned kBaAcceptedInTrial = 1, kBaWantMaxdiag = 2, kBaWantHpp = 4;
struct BaBatchExtent {
    int max_edges, max_points, max_poses, max_free, max_free_edges, max_groups, max_np_pad, max_slices, max_planes, max_chunks, max_W;
    // max_np_pad / max_slices: over the windows on the dense Schur path; the sparse ones:
    int any_dense;
    // the windows on the block-by-block sparse path (pb.schur_blocks): partial sums per window, free keyframes
    int max_block_parts, max_block_free, min_block_free;
    int any_block_lean;                 // some window runs the lean block-by-block product
    int any_block_wide;                 // some window runs the lean form with two workgroups per part (more than kSchurBlocksMaxFree free keyframes)
    int fuse_trial;   // the trial errors' last workgroup of a window does k_ba_trial_reduce_b's sums
    int fuse_linearize;  // the linearisation's last workgroup of a window does k_ba_reduce_all_b's / k_ba_maxdiag_b's sums (round 5)
    int any_dups;  // some window has duplicate (point, free pose) edges: k_ba_dups_b after the linearisation's sums
    int any_trial_fused, any_trial_unfused;  // windows with / without pb.trial_fused in the call (each kind has its launches; a kernel skips the other kind)
    int inertial;  // the windows' vertices are ImuCamPose records (LocalLVIBA batch): the linearisation kernel of that vertex type
};
// n_active <= kBaPhaseMax windows per call (the host cuts a longer list)
void ba_batch_launch_linearize(const BaPhase& ph, int n_active, const BaBatchExtent& x, bool any_maxdiag, hipStream_t st);
void ba_batch_launch_schur(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st);
// x = (S + Hl)^-1 (b_s + bl) per window by dense LDL^T, one workgroup per window (windows of at most 21 free keyframes)
void ba_batch_launch_solve(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st);
bool lvi_device_solve_available();  // the kernels below get the LDS of their largest window on this device
// the reduced systems of inertial windows (slot.lvi.n > 0) on the device; max_np / max_ni over the windows of the launch
void lvi_batch_launch_solve(const BaPhase& ph, int n_active, int max_np, int max_ni, hipStream_t st);
// one window: S / bs as k_ba_schur_finish left them in device memory, x (n unknowns, the caller's numbering) to x_dev and x_host, ok_host[0] = the pivots were usable
void lvi_launch_solve(const LviSolveDev& q, const double* S, const double* bs, double lambda, double* x_dev, double* x_host, int32_t* ok_host, hipStream_t st);
void ba_batch_launch_trial(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st);
void ba_batch_launch_depth(const BaPhase& ph, int n_active, const BaBatchExtent& x, hipStream_t st);
// device-side LM (ba_lm_kernels.hip): one wavefront per window of the phase
void ba_batch_launch_lm_begin(const BaPhase& ph, int n_active, hipStream_t st);
void ba_batch_launch_lm_decide(const BaPhase& ph, int n_active, hipStream_t st);
// the LiDAR term of the listed windows (all with W <= 7 and at most 2048 planes): residual at the accepted or the trial poses,
// Jacobian / Hessian at the accepted poses
void balm_batch_launch_residual(const BaPhase& ph, int n, bool trial, hipStream_t st);
void balm_batch_launch_hessian(const BaPhase& ph, int n, const BaBatchExtent& x, hipStream_t st);

#if defined(__HIPCC__)
// *p for an object no kernel of the launch writes, at an address that is the same for the whole wavefront: read through the constant
// address space, i.e. with scalar loads into SGPRs (the slot table is written by the host's upload only)
template <typename T>
__device__ __forceinline__ T load_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized objects");
    typedef const unsigned int __attribute__((address_space(4)))* const_dwords;
    const const_dwords c = (const_dwords)(unsigned long long)p;
    unsigned int w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = c[i];
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}
// the slot record's pointers as global pointers (global_ptr.hpp)
__device__ __forceinline__ void ba_problem_pointers_are_global(BaProblemDev& pb) {
#define TC2LI_G(f) pb.f = global_ptr(pb.f)
    TC2LI_G(poses); TC2LI_G(poses_trial); TC2LI_G(iposes); TC2LI_G(iposes_trial); TC2LI_G(points); TC2LI_G(points_trial); TC2LI_G(edges);
    TC2LI_G(pose_var); TC2LI_G(pt_off); TC2LI_G(pt_edges); TC2LI_G(pv_off); TC2LI_G(pv_edges); TC2LI_G(grp_k0); TC2LI_G(grp_l0);
    TC2LI_G(fl_off); TC2LI_G(fl_pose); TC2LI_G(chunk_mask); TC2LI_G(fl_lm); TC2LI_G(fl_place); TC2LI_G(slice_off); TC2LI_G(fl_edge);
    TC2LI_G(chi2); TC2LI_G(rho0); TC2LI_G(cp_part); TC2LI_G(W); TC2LI_G(blk_off); TC2LI_G(blk_rows);
    TC2LI_G(Hll); TC2LI_G(bl); TC2LI_G(diag_l); TC2LI_G(Hpp); TC2LI_G(diag_p); TC2LI_G(coef_e); TC2LI_G(coef); TC2LI_G(Y); TC2LI_G(dup_off); TC2LI_G(dup_edge); TC2LI_G(dup_slot);
    TC2LI_G(S_part); TC2LI_G(scale_part); TC2LI_G(chi_part); TC2LI_G(ticket);
#undef TC2LI_G
}
// the phase's window number / flags at position `pos` as scalars: sub-dword loads from the argument block at a dynamic index are vector
// loads, and everything addressed through their result would be fetched per lane (readfirstlane: the value is the same in every lane)
__device__ __forceinline__ int ba_phase_window(const BaPhase& ph, int pos) { return __builtin_amdgcn_readfirstlane((int)ph.win[pos]); }
__device__ __forceinline__ unsigned ba_phase_flags(const BaPhase& ph, int pos) { return (unsigned)__builtin_amdgcn_readfirstlane((int)ph.flags[pos]); }
// What a batched kernel needs of the window's LM state (device-side LM): whether the launch is meant for the window now, the parity / request
// bits in BaPhase::flags' encoding, lambda.  Scalar loads: the state was written by an earlier launch of the stream (k_ba_lm_begin_b /
// k_ba_lm_decide_b / the call's upload), never by this one.
struct BaLmView { bool active; unsigned flags; double lambda; };
__device__ __forceinline__ BaLmView ba_lm_view(const BaPhase& ph, const BaBatchSlot* sl, unsigned phase_flags, double phase_lambda) {
    const BaLmState* lm = load_uniform(&sl->lm);
    if (!lm) return BaLmView{true, phase_flags, phase_lambda};
    const int32_t status = load_uniform(&lm->status), parity = load_uniform(&lm->parity), it = load_uniform(&lm->it);
    const double lambda_init = load_uniform(&sl->lambda_init);
    BaLmView v;
    v.active = ph.expect == 0 || status == ph.expect;
    v.flags = (parity ? kBaAcceptedInTrial : 0u) | (it == 0 && !(lambda_init > 0) ? kBaWantMaxdiag : 0u);
    v.lambda = load_uniform(&lm->lambda);
    return v;
}
#endif

#if defined(__HIPCC__)
// The last workgroup of a window to deliver (a ticket counter per window, zero between launches).  The partial sums are stored and read at
// device scope (block_sum_256<true> / load_partial: `global_store ... sc1` writes through the XCD's L2, `global_load ... sc1` does not take
// what that L2 holds), so no cache-wide write-back / invalidate is needed -- but the stores must have been PERFORMED before the ticket is
// taken: every wavefront waits for its own outstanding memory operations (`s_waitcnt vmcnt(0)`: a device-scope store is acknowledged once
// it is visible at that scope), the workgroup meets, and only then one thread takes the ticket with a device-scope atomic.  (Round 4 had a
// workgroup-scope release fence in place of the wait; hipcc emits no `s_waitcnt vmcnt` for that on gfx950, so the store and the ticket --
// different addresses, different L2 channels -- were unordered and the last workgroup could add a stale partial: ADVICE r4, high.)  The
// workgroup that took the last ticket leaves the counter at zero for the next launch and does the closing sums -- in the fixed order of
// the separate kernel, whoever comes last.
__device__ __forceinline__ bool ba_last_of(int32_t* ticket, int n_workgroups) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == n_workgroups - 1 ? 1 : 0;
        if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}
#endif

}  // namespace tc2li
