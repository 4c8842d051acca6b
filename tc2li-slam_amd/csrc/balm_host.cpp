// See balm_host.hpp.  Host stages of the LiDAR plane term: plane extraction (once per local BA), the change of variables
// of the 6W x 6W system (once per linearisation) and the edge state machine.
#include "balm_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace tc2li {

namespace {

constexpr int kLayerLimit = 2, kMinPoints = 15;                       // SF/include/bavoxel.h:35-41
constexpr float kPlaneRatio[3] = {1.0f / 36, 1.0f / 25, 1.0f / 25};  // eigen_value_array
constexpr double kVoxelSize = 1.0;

struct VoxelKey {
    int64_t x, y, z;
    bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {
    size_t operator()(const VoxelKey& k) const {
        uint64_t h = (uint64_t)k.x * 0x9E3779B97F4A7C15ull;
        h ^= (uint64_t)k.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
        h ^= (uint64_t)k.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
        return (size_t)h;
    }
};

struct WindowPoint { double local[3], world[3]; int32_t slot; };

struct Moments {  // running sums in insertion order: PointCluster::push
    double P[6] = {0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
    int n = 0;
    void push(const double* x) {
        ++n;
        P[0] += x[0] * x[0]; P[1] += x[0] * x[1]; P[2] += x[0] * x[2]; P[3] += x[1] * x[1]; P[4] += x[1] * x[2]; P[5] += x[2] * x[2];
        v[0] += x[0]; v[1] += x[1]; v[2] += x[2];
    }
};

struct Cell { int begin, end, layer; float center[3], quarter; };

}  // namespace

// Work space of balm_build_planes, one per host thread: the function runs once per window on the pool threads of a lock-step group and was
// three quarters of a group's setup time (0.6-1.1 ms of a window's 0.9-1.3): a node-based hash map with an allocation per root voxel, six
// vectors sized per call, and the per-keyframe moments of BOTH frames for every cell visited.  Same arithmetic in the same order below.
struct PlaneScratch {
    std::vector<WindowPoint> pts;
    std::vector<int> root_id, root_count, start, order, scratch, table;  // table: open addressing, root id + 1 per slot, 0 = empty
    std::vector<VoxelKey> root_key;
    std::vector<Moments> local, world;
    std::vector<Cell> stack;
};

// keyframe i's LiDAR frame -> keyframe 0's: x0 = R x + p (cut_voxel works in the frame of the window's first keyframe)
static void window_relative(const LidarPose* twl, int i, double R[9], double p[3]) {
    double R0t[9], d[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R0t[3 * r + c] = twl[0].R[3 * c + r];
    for (int k = 0; k < 3; ++k) d[k] = twl[i].p[k] - twl[0].p[k];
    m3_vec(R0t, d, p);
    m3_mul(R0t, twl[i].R, R);
}

void balm_build_planes(const LidarPose* twl, int W, const float* cloud, const int32_t* off, std::vector<PlaneCluster>& clusters,
                       std::vector<double>& coe) {
    static thread_local PlaneScratch ws;
    clusters.clear();
    coe.clear();
    const int total = off[W];
    std::vector<WindowPoint>& pts = ws.pts;
    pts.resize(total);
    // ---- points into the frame of the first keyframe's LiDAR, root voxel of every point ----
    size_t cap = 64;
    int cap_bits = 6;
    while (cap < (size_t)total) { cap <<= 1; ++cap_bits; }  // every point its own voxel still leaves the table half empty
    cap <<= 1; ++cap_bits;
    std::vector<int>& table = ws.table;
    table.assign(cap, 0);
    std::vector<VoxelKey>& root_key = ws.root_key;
    std::vector<int>&root_id = ws.root_id, &root_count = ws.root_count;
    root_key.clear(); root_count.clear();
    root_id.resize(total);
    const VoxelKeyHash hasher;
    for (int i = 0; i < W; ++i) {
        double p[3], R[9];
        window_relative(twl, i, R, p);
        VoxelKey last_key{0, 0, 0};
        int last_id = -1;  // consecutive returns of a scan mostly fall into the same voxel
        for (int j = off[i]; j < off[i + 1]; ++j) {
            WindowPoint& q = pts[j];
            q.slot = i;
            for (int k = 0; k < 3; ++k) q.local[k] = (double)cloud[3 * (size_t)j + k];
            double Rx[3];
            m3_vec(R, q.local, Rx);
            VoxelKey key;
            int64_t* kk = &key.x;
            for (int k = 0; k < 3; ++k) {
                q.world[k] = Rx[k] + p[k];
                float loc = (float)(q.world[k] / kVoxelSize);
                if (loc < 0) loc -= 1.0f;
                kk[k] = (int64_t)loc;
            }
            int id;
            if (last_id >= 0 && key == last_key) {
                id = last_id;
            } else {
                // the slot from the HIGH bits of a multiplied hash: the low bits of the key mix follow the low bits of the coordinates
                size_t h = (size_t)(((uint64_t)hasher(key) * 0x9E3779B97F4A7C15ull) >> (64 - cap_bits));
                for (;;) {
                    const int t = table[h];
                    if (t == 0) {  // first appearance: ids in that order, as the insertion order of the reference's map walk
                        id = (int)root_key.size();
                        table[h] = id + 1;
                        root_key.push_back(key);
                        root_count.push_back(0);
                        break;
                    }
                    if (root_key[t - 1] == key) { id = t - 1; break; }
                    h = (h + 1) & (cap - 1);
                }
                last_key = key; last_id = id;
            }
            root_id[j] = id;
            root_count[id]++;
        }
    }
    // ---- counting sort by root voxel; inside a voxel the points stay in (keyframe, scan) order ----
    const int n_roots = (int)root_key.size();
    std::vector<int>&start = ws.start, &order = ws.order, &scratch = ws.scratch;
    start.assign(n_roots + 1, 0);
    for (int r = 0; r < n_roots; ++r) start[r + 1] = start[r] + root_count[r];
    order.resize(total); scratch.resize(total);
    {
        std::vector<int>& cur = ws.root_count;  // the counts have served: running places
        for (int r = 0; r < n_roots; ++r) cur[r] = start[r];
        for (int j = 0; j < total; ++j) order[cur[root_id[j]]++] = j;
    }
    // ---- every root voxel: plane test, split into octants while not planar ----
    std::vector<Moments>&local = ws.local, &world = ws.world;
    local.resize(W); world.resize(W);
    std::vector<Cell>& stack = ws.stack;
    for (int r = 0; r < n_roots; ++r) {
        if (start[r + 1] - start[r] <= kMinPoints) continue;  // (the walk below would pop the root and drop it)
        Cell root;
        root.begin = start[r]; root.end = start[r + 1]; root.layer = 0;
        const int64_t* kk = &root_key[r].x;
        for (int k = 0; k < 3; ++k) root.center[k] = (float)((0.5 + (double)kk[k]) * kVoxelSize);
        root.quarter = (float)(kVoxelSize / 4.0);
        stack.clear();
        stack.push_back(root);
        while (!stack.empty()) {
            const Cell cell = stack.back();
            stack.pop_back();
            const int count = cell.end - cell.begin;
            if (count <= kMinPoints) continue;
            // the moments of the common frame decide; those of the keyframes' own frames are only formed for a cell that is a plane
            for (int i = 0; i < W; ++i) world[i] = Moments();
            for (int a = cell.begin; a < cell.end; ++a) {
                const WindowPoint& q = pts[order[a]];
                world[q.slot].push(q.world);
            }
            // covariance of all points of the cell in the common frame
            double P[6] = {0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
            int n = 0;
            for (int i = 0; i < W; ++i) {
                for (int k = 0; k < 6; ++k) P[k] += world[i].P[k];
                for (int k = 0; k < 3; ++k) v[k] += world[i].v[k];
                n += world[i].n;
            }
            const double inv = 1.0 / n;
            const double c[3] = {inv * v[0], inv * v[1], inv * v[2]};
            double Ps[9], C[9], lambda[3], U[9];
            sym_unpack(P, Ps);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = inv * Ps[3 * a + b] - c[a] * c[b];
            eig_sym3(C, lambda, U);
            if (lambda[0] / lambda[1] < (double)kPlaneRatio[cell.layer]) {
                int seen = 0;
                for (int i = 0; i < W; ++i) seen += world[i].n != 0;
                if (seen < 2) continue;  // VOX_HESS::push_voxel: a plane must be seen from two keyframes
                for (int i = 0; i < W; ++i) local[i] = Moments();
                for (int a = cell.begin; a < cell.end; ++a) {
                    const WindowPoint& q = pts[order[a]];
                    local[q.slot].push(q.local);
                }
                double weight = 0;
                for (int i = 0; i < W; ++i) {
                    PlaneCluster pc;
                    memcpy(pc.P, local[i].P, sizeof(pc.P));
                    memcpy(pc.v, local[i].v, sizeof(pc.v));
                    pc.n = (double)local[i].n;
                    weight += (double)local[i].n;
                    clusters.push_back(pc);
                }
                coe.push_back(weight);
                continue;
            }
            if (cell.layer == kLayerLimit) continue;
            // stable split into the 8 octants around the cell centre
            int cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto octant = [&](const WindowPoint& q) {
                return 4 * (q.world[0] > cell.center[0]) + 2 * (q.world[1] > cell.center[1]) + (q.world[2] > cell.center[2]);
            };
            for (int a = cell.begin; a < cell.end; ++a) cnt[octant(pts[order[a]]) + 1]++;
            for (int o = 0; o < 8; ++o) cnt[o + 1] += cnt[o];
            int cur[8];
            for (int o = 0; o < 8; ++o) cur[o] = cell.begin + cnt[o];
            for (int a = cell.begin; a < cell.end; ++a) { const int j = order[a]; scratch[cur[octant(pts[j])]++] = j; }
            std::copy(scratch.begin() + cell.begin, scratch.begin() + cell.end, order.begin() + cell.begin);
            for (int o = 7; o >= 0; --o) {  // pushed in reverse so that octant 0 is visited first
                if (cnt[o + 1] == cnt[o]) continue;
                Cell child;
                child.begin = cell.begin + cnt[o]; child.end = cell.begin + cnt[o + 1]; child.layer = cell.layer + 1;
                const int bit[3] = {(o >> 2) & 1, (o >> 1) & 1, o & 1};
                for (int k = 0; k < 3; ++k) child.center[k] = cell.center[k] + (2 * bit[k] - 1) * cell.quarter;
                child.quarter = cell.quarter / 2;
                stack.push_back(child);
            }
        }
    }
}

void balm_to_camera_se3(const LidarPose* twl, int W, const SE3f& Tcl, double* JacT, double* H) {
    const int n = 6 * W;
    const BalmCameraFrame F = balm_camera_frame(Tcl);
    std::vector<double> DT(36 * (size_t)W);
    for (int i = 0; i < W; ++i) balm_camera_se3_D(twl[i], F, JacT + 6 * i, DT.data() + 36 * (size_t)i);
    for (int a = 0; a < W; ++a)
        for (int b = 0; b < W; ++b) balm_change_block(H, n, a, b, DT.data() + 36 * (size_t)a, DT.data() + 36 * (size_t)b);
}

void balm_to_body(const LidarPose* twl, int W, const SE3f& Tbl, double* JacT, double* H) {
    const int n = 6 * W;
    const float qi[4] = {-Tbl.q[0], -Tbl.q[1], -Tbl.q[2], Tbl.q[3]};  // mTlb = mTbl.inverse()
    double Rlb[9];
    quat_to_matrix_f(qi, Rlb);
    const double tbl[3] = {(double)Tbl.t[0], (double)Tbl.t[1], (double)Tbl.t[2]};
    std::vector<double> DT(36 * (size_t)W);
    for (int i = 0; i < W; ++i) balm_body_D(twl[i], Rlb, tbl, JacT + 6 * i, DT.data() + 36 * (size_t)i);
    for (int a = 0; a < W; ++a)
        for (int b = 0; b < W; ++b) balm_change_block(H, n, a, b, DT.data() + 36 * (size_t)a, DT.data() + 36 * (size_t)b);
}

// ---- the edge ---------------------------------------------------------------------------------------------------------
int BalmTerm::check_window(const tc2li_lidar_window* win, int n_poses) {
    const int W = win->n_keyframes;
    if (W < 1 || W > kMaxLidarWindow || !win->pose_index || !win->cloud_xyz || !win->cloud_offsets) {
        set_error("lidar window: n_keyframes must be in [1, %d] and the arrays non-null", kMaxLidarWindow);
        return TC2LI_ERR_INVALID;
    }
    if (win->cloud_offsets[0] != 0) { set_error("lidar window: cloud_offsets[0] must be 0"); return TC2LI_ERR_INVALID; }
    for (int i = 0; i < W; ++i) {
        const int k = win->pose_index[i];
        if (k < 0 || k >= n_poses) { set_error("lidar window: pose_index[%d] = %d out of range", i, k); return TC2LI_ERR_INVALID; }
        if (win->cloud_offsets[i + 1] <= win->cloud_offsets[i]) { set_error("lidar window: keyframe %d has no points", i); return TC2LI_ERR_INVALID; }
    }
    return 0;
}

int BalmTerm::window_poses(const double* poses7, int n_poses, const tc2li_lidar_window* win, std::vector<LidarPose>& twl) {
    const int rc = check_window(win, n_poses);
    if (rc < 0) return rc;
    const int W = win->n_keyframes;
    SE3f Tcl;
    memcpy(Tcl.q, win->Tcl, 4 * sizeof(float));
    memcpy(Tcl.t, win->Tcl + 4, 3 * sizeof(float));
    twl.resize(W);
    for (int i = 0; i < W; ++i) {
        const int k = win->pose_index[i];
        SE3f Tcw;  // KeyFrame::GetPose() is a Sophus::SE3f
        for (int c = 0; c < 4; ++c) Tcw.q[c] = (float)poses7[7 * k + c];
        for (int c = 0; c < 3; ++c) Tcw.t[c] = (float)poses7[7 * k + 4 + c];
        twl[i] = lidar_pose_from(Tcw, Tcl);
    }
    return 0;
}

int BalmTerm::build(const double* poses7, int n_poses, const tc2li_lidar_window* win, hipStream_t st, BalmCutTask* cut) {
    std::vector<LidarPose> twl;
    const int rcw = window_poses(poses7, n_poses, win, twl);
    if (rcw < 0) return rcw;
    body = false;
    return upload(twl, win, st, cut);
}

int BalmTerm::build_body(const void* kfs, size_t kf_bytes, int n_kfs, const tc2li_lidar_window* win, const float* Tbl7, size_t imu_pose_bytes,
                         hipStream_t st, BalmCutTask* cut) {
    const int rc = check_window(win, n_kfs);
    if (rc < 0) return rc;
    SE3f tcl;
    memcpy(tcl.q, win->Tcl, 4 * sizeof(float));
    memcpy(tcl.t, win->Tcl + 4, 3 * sizeof(float));
    std::vector<LidarPose> twl(win->n_keyframes);
    for (int i = 0; i < win->n_keyframes; ++i) {  // pKF->GetPoseInverse() * mTcl (SF/src/LidarRes.cc:44)
        const double* rt = reinterpret_cast<const double*>(static_cast<const char*>(kfs) + (size_t)win->pose_index[i] * kf_bytes);
        twl[i] = lidar_pose_from(se3f_from_rt(rt, rt + 9), tcl);
    }
    body = true;
    memcpy(Tbl.q, Tbl7, 4 * sizeof(float));
    memcpy(Tbl.t, Tbl7 + 4, 3 * sizeof(float));
    const int r = upload(twl, win, st, cut);
    dev.imu_pose_bytes = (int32_t)imu_pose_bytes;
    return r;
}

int BalmTerm::set_planes(int n) {
    n_planes = n;
    dev.n_planes = n;
    // a workgroup of the Hessian kernel takes whole batches of planes (8 for windows of <= 7 keyframes, else 4: balm_kernels.hip), at
    // most 1024 workgroups; a function of the window alone (the partial sums' grouping decides the bits)
    const int plane_batch = W <= 7 ? 8 : 4;
    dev.planes_per_chunk = plane_batch * std::max(1, (n_planes + plane_batch * 1024 - 1) / (plane_batch * 1024));
    dev.n_chunks = n_planes ? (n_planes + dev.planes_per_chunk - 1) / dev.planes_per_chunk : 1;
    TC2LI_HIP_CHECK(d_plane_res.ensure(std::max(n_planes, 1)));
    TC2LI_HIP_CHECK(d_eig.ensure((size_t)kBalmEig * std::max(n_planes, 1)));
    eig_at = nullptr;
    TC2LI_HIP_CHECK(d_part.ensure((size_t)dev.n_chunks * balm_part_stride(W)));
    dev.clusters = d_clusters.p; dev.coe = d_coe.p; dev.pose_index = d_pose_index.p; dev.twl = d_twl.p;
    dev.plane_res = d_plane_res.p; dev.eig = d_eig.p; dev.part = d_part.p; dev.out = h_out.p;
    return 0;
}

int BalmTerm::upload(const std::vector<LidarPose>& twl, const tc2li_lidar_window* win, hipStream_t st, BalmCutTask* cut) {
    W = win->n_keyframes;
    memcpy(Tcl.q, win->Tcl, 4 * sizeof(float));
    memcpy(Tcl.t, win->Tcl + 4, 3 * sizeof(float));
    information = win->weight;
    error = 0; r1 = 1000; r2 = 1000; is_calc_hess = true; hessian_evaluations = 0;
    JacT.assign(6 * (size_t)W, 0.0);
    Hessian.assign(36 * (size_t)W * W, 0.0);
    pose_index.assign(win->pose_index, win->pose_index + W);
    dev = BalmDev{};
    dev.W = W;
    dev.Tcl = Tcl;
    cut_pending = false;
    TC2LI_HIP_CHECK(d_pose_index.ensure(W));
    TC2LI_HIP_CHECK(d_twl.ensure(W));
    TC2LI_HIP_CHECK(h_out.ensure(balm_out_size(W)));
    TC2LI_HIP_CHECK(h_twl.ensure(W));
    const int total = win->cloud_offsets[W];
    if (cut) cut->n_points = 0;
    static const bool kHostCut = getenv("TC2LI_BALM_HOST_CUT") && atoi(getenv("TC2LI_BALM_HOST_CUT")) != 0;  // measurement: the round-3 path
    if (cut && !kHostCut && copy_sink_active() && W <= kBalmCutMaxW && total <= kBalmCutMaxPoints) {
        // ---- the extraction runs on the device: stage the clouds, carve the work space, describe the window ----
        const size_t cloud_bytes = ((size_t)total * 3 * sizeof(float) + 15) & ~(size_t)15;
        TC2LI_HIP_CHECK(h_upload.ensure(cloud_bytes + W * sizeof(int32_t) + 16));
        TC2LI_HIP_CHECK(h_cut_result.ensure(4));
        uint8_t* h = h_upload.p;
        memcpy(h, win->cloud_xyz, (size_t)total * 3 * sizeof(float));
        memcpy(h + cloud_bytes, pose_index.data(), W * sizeof(int32_t));
        int bits = 7;
        while (((size_t)1 << bits) < 2 * (size_t)total) ++bits;
        const size_t cap = (size_t)1 << bits, n = (size_t)total;
        size_t at = 0;
        auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 15) & ~(size_t)15; return o; };
        const size_t o_key = take(cap * 8), o_first = take(cap * 4), o_id = take(cap * 4), o_world = take(n * 24), o_cloud = take(cloud_bytes), o_oct = take(n),
                     o_slot = take(n * 4), o_ka = take(n * 4), o_kb = take(n * 4), o_va = take(n * 4), o_vb = take(n * 4), o_order = take(3 * n * 4),
                     o_begin = take(3 * (n + 1) * 4), o_ckey = take(3 * n * 4), o_flag = take(3 * n), o_ncells = take(16), o_plane = take(kBalmCutMaxPlanes * 4),
                     o_state = take(16);
        TC2LI_HIP_CHECK(d_cut.ensure(at));
        TC2LI_HIP_CHECK(d_clusters.ensure((size_t)kBalmCutMaxPlanes * W));
        TC2LI_HIP_CHECK(d_coe.ensure(kBalmCutMaxPlanes));
        uint8_t* d = d_cut.p;
        TC2LI_HIP_CHECK(upload_or_defer(d + o_cloud, h, cloud_bytes, st));
        TC2LI_HIP_CHECK(upload_or_defer(d_pose_index.p, h + cloud_bytes, W * sizeof(int32_t), st));
        BalmCutTask& T = *cut;
        T = BalmCutTask{};
        T.W = W; T.n_points = total; T.table_bits = bits;
        for (int i = 0; i <= kBalmCutMaxW; ++i) T.cloud_off[i] = win->cloud_offsets[std::min(i, W)];
        for (int i = 0; i < W; ++i) window_relative(twl.data(), i, T.rel[i].R, T.rel[i].p);
        T.cloud = (const float*)(d + o_cloud);
        T.table_key = (unsigned long long*)(d + o_key); T.table_first = (int32_t*)(d + o_first); T.table_id = (int32_t*)(d + o_id);
        T.world = (double*)(d + o_world); T.oct = d + o_oct; T.point_slot = (int32_t*)(d + o_slot);
        T.sort_key_a = (unsigned int*)(d + o_ka); T.sort_key_b = (unsigned int*)(d + o_kb); T.sort_val_a = (int32_t*)(d + o_va); T.sort_val_b = (int32_t*)(d + o_vb);
        T.order = (int32_t*)(d + o_order); T.cell_begin = (int32_t*)(d + o_begin); T.cell_key = (unsigned int*)(d + o_ckey); T.cell_flag = d + o_flag;
        T.n_cells = (int32_t*)(d + o_ncells); T.plane_cell = (int32_t*)(d + o_plane); T.state = (int32_t*)(d + o_state);
        T.result_host = h_cut_result.p;
        T.clusters = d_clusters.p; T.coe = d_coe.p;
        h_cut_result.p[0] = 0; h_cut_result.p[1] = 1;  // (a launch that never ran reads as "declined")
        twl_build = twl;
        win_build = win;
        cut_pending = true;
        // the pose-dependent buffers at their largest, so that finish_cut allocates nothing behind the batch's synchronisation
        TC2LI_HIP_CHECK(d_plane_res.ensure(kBalmCutMaxPlanes));
        TC2LI_HIP_CHECK(d_eig.ensure((size_t)kBalmEig * kBalmCutMaxPlanes));
        TC2LI_HIP_CHECK(d_part.ensure((size_t)(kBalmCutMaxPlanes / 8) * balm_part_stride(W)));
        return set_planes(0);
    }
    // members: the asynchronous uploads below read them, and a synchronisation here would make every window of a lock-step group
    // wait for everything the other windows have queued on the shared stream (their operand fills, their uploads)
    std::vector<PlaneCluster>& clusters = h_clusters;
    std::vector<double>& coe = h_coe;
    clusters.clear();
    coe.clear();
    balm_build_planes(twl.data(), W, win->cloud_xyz, win->cloud_offsets, clusters, coe);
    const int n = (int)coe.size();
    TC2LI_HIP_CHECK(d_clusters.ensure(std::max(clusters.size(), (size_t)1)));
    TC2LI_HIP_CHECK(d_coe.ensure(std::max(n, 1)));
    if (copy_sink_active()) {
        // a lock-step group gathers the uploads of all its windows into one launch that reads the sources in place: they must be pinned
        const size_t cb = (clusters.size() * sizeof(PlaneCluster) + 15) & ~(size_t)15, qb = (coe.size() * sizeof(double) + 15) & ~(size_t)15;
        TC2LI_HIP_CHECK(h_upload.ensure(cb + qb + W * sizeof(int32_t) + 16));
        uint8_t* h = h_upload.p;
        if (n) {
            memcpy(h, clusters.data(), clusters.size() * sizeof(PlaneCluster));
            memcpy(h + cb, coe.data(), coe.size() * sizeof(double));
            TC2LI_HIP_CHECK(upload_or_defer(d_clusters.p, h, clusters.size() * sizeof(PlaneCluster), st));
            TC2LI_HIP_CHECK(upload_or_defer(d_coe.p, h + cb, coe.size() * sizeof(double), st));
        }
        memcpy(h + cb + qb, pose_index.data(), W * sizeof(int32_t));
        TC2LI_HIP_CHECK(upload_or_defer(d_pose_index.p, h + cb + qb, W * sizeof(int32_t), st));
    } else {
        if (n) {
            TC2LI_HIP_CHECK(hipMemcpyAsync(d_clusters.p, clusters.data(), clusters.size() * sizeof(PlaneCluster), hipMemcpyHostToDevice, st));
            TC2LI_HIP_CHECK(hipMemcpyAsync(d_coe.p, coe.data(), coe.size() * sizeof(double), hipMemcpyHostToDevice, st));
        }
        TC2LI_HIP_CHECK(hipMemcpyAsync(d_pose_index.p, pose_index.data(), W * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    return set_planes(n);
}

int BalmTerm::finish_cut(hipStream_t st) {
    if (!cut_pending) return 0;
    cut_pending = false;
    const int32_t* r = h_cut_result.p;
    if (!r[1]) return set_planes(r[0]);
    // the kernels declined the window: the host extraction, its clusters up by a copy of their own (rare; the vectors are members)
    h_clusters.clear();
    h_coe.clear();
    balm_build_planes(twl_build.data(), W, win_build->cloud_xyz, win_build->cloud_offsets, h_clusters, h_coe);
    const int n = (int)h_coe.size();
    TC2LI_HIP_CHECK(d_clusters.ensure(std::max(h_clusters.size(), (size_t)1)));
    TC2LI_HIP_CHECK(d_coe.ensure(std::max(n, 1)));
    if (n) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(d_clusters.p, h_clusters.data(), h_clusters.size() * sizeof(PlaneCluster), hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(d_coe.p, h_coe.data(), h_coe.size() * sizeof(double), hipMemcpyHostToDevice, st));
    }
    return set_planes(n);
}

void BalmTerm::enqueue_error(const Se3* d_poses, hipStream_t st) {
    if (!n_planes) return;
    balm_launch_residual(dev, d_poses, st);
    eig_at = d_poses;
}

void BalmTerm::finish_error() {
    const double r = n_planes ? h_out.p[0] : 0.0;
    error = body ? std::sqrt(r) : r;  // G2oTypesWithLidar.cc:41-42 / G2oTypesWithLidar.h:131
    r1 = r2;
    r2 = r;
    is_calc_hess = !(r1 - r2 < 0);  // the Hessian is kept while the cost grows (G2oTypesWithLidar.h:130-139)
}

int BalmTerm::enqueue_linearization(const Se3* d_poses, hipStream_t st) {
    if (!n_planes) return 0;
    // the Hessian kernel starts from the eigen decompositions of a residual pass at the same poses: the optimisers run one right
    // before (computeActiveErrors, then linearizeOplus); any other caller gets it here
    if (eig_at != d_poses) { balm_launch_residual(dev, d_poses, st); eig_at = d_poses; }
    balm_launch_hessian(dev, d_poses, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    return 0;
}

void BalmTerm::finish_linearization() {
    if (!is_calc_hess) return;
    ++hessian_evaluations;
    if (!n_planes) {
        std::fill(JacT.begin(), JacT.end(), 0.0);
        std::fill(Hessian.begin(), Hessian.end(), 0.0);
        return;
    }
    const int n = 6 * W;
    memcpy(JacT.data(), h_out.p + 1, n * sizeof(double));
    memcpy(Hessian.data(), h_out.p + 1 + n, (size_t)n * n * sizeof(double));
    const LidarPose* at = reinterpret_cast<const LidarPose*>(h_out.p + 2 + n + (size_t)n * n);
    if (body) balm_to_body(at, W, Tbl, JacT.data(), Hessian.data());
    else balm_to_camera_se3(at, W, Tcl, JacT.data(), Hessian.data());
}

int BalmTerm::compute_error(const Se3* d_poses, hipStream_t st) {
    enqueue_error(d_poses, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    finish_error();
    return 0;
}

int BalmTerm::linearize(const Se3* d_poses, hipStream_t st) {
    if (!is_calc_hess) return 0;
    const int rc = enqueue_linearization(d_poses, st);
    if (rc < 0) return rc;
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    finish_linearization();
    return 0;
}

void BalmTerm::add_quadratic_form(const int* pose_var, int np /* row stride of Hpp */, double* Hpp, double* b) const {
    // The reference reads the 6x6 blocks at ELEMENT offsets (i, i) / (i, j) of the 6W x 6W Hessian and subtracts
    // information * J^T from b (G2oTypesWithLidar.h:168-236); kept as is so that the optimiser takes the same steps.
    const int n = 6 * W;
    for (int i = 0; i < W; ++i) {
        const int vi = pose_var[pose_index[i]];
        if (vi < 0) continue;
        for (int r = 0; r < 6; ++r) {
            b[6 * vi + r] -= information * JacT[6 * i + r];
            for (int c = 0; c < 6; ++c) Hpp[(size_t)(6 * vi + r) * np + 6 * vi + c] += Hessian[(size_t)(i + r) * n + i + c] * information;
        }
        for (int j = i + 1; j < W; ++j) {
            const int vj = pose_var[pose_index[j]];
            if (vj < 0) continue;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    const double h = Hessian[(size_t)(i + r) * n + j + c] * information;
                    Hpp[(size_t)(6 * vi + r) * np + 6 * vj + c] += h;
                    Hpp[(size_t)(6 * vj + c) * np + 6 * vi + r] += h;
                }
        }
    }
}

}  // namespace tc2li

extern "C" int tc2li_host_lidar_planes(const double* poses7, int n_poses, const tc2li_lidar_window* win, double* clusters, double* coe,
                                       int capacity) {
    using namespace tc2li;
    if (!poses7 || n_poses <= 0 || !win || capacity < 0 || (capacity > 0 && (!clusters || !coe))) {
        set_error("tc2li_host_lidar_planes: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    std::vector<LidarPose> twl;
    const int rc = BalmTerm::window_poses(poses7, n_poses, win, twl);
    if (rc < 0) return rc;
    std::vector<PlaneCluster> cl;
    std::vector<double> w;
    balm_build_planes(twl.data(), win->n_keyframes, win->cloud_xyz, win->cloud_offsets, cl, w);
    const int n = (int)w.size(), m = std::min(n, capacity);
    static_assert(sizeof(PlaneCluster) == 10 * sizeof(double), "cluster layout");
    if (m > 0) {
        memcpy(clusters, cl.data(), (size_t)m * win->n_keyframes * sizeof(PlaneCluster));
        memcpy(coe, w.data(), (size_t)m * sizeof(double));
    }
    return n;
}

// The device extraction of one window, for the tests: the clusters balm_cut_kernels.hip forms, read back.
extern "C" int tc2li_device_lidar_planes(const double* poses7, int n_poses, const tc2li_lidar_window* win, double* clusters, double* coe,
                                         int capacity, int32_t* info) {
    using namespace tc2li;
    if (!poses7 || n_poses <= 0 || !win || capacity < 0 || (capacity > 0 && (!clusters || !coe))) {
        set_error("tc2li_device_lidar_planes: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    note_hip_touched();
    BalmTerm term;
    BalmCutTask task{};
    std::vector<CopyTask> copies;
    int rc;
    {
        CopySink sink(&copies);
        rc = term.build(poses7, n_poses, win, nullptr, &task);
    }
    if (rc < 0) return rc;
    if (task.n_points <= 0) { set_error("tc2li_device_lidar_planes: the window is outside the kernels' range (keyframes %d, points %d)", win->n_keyframes, win->cloud_offsets[win->n_keyframes]); return TC2LI_ERR_INVALID; }
    for (const CopyTask& c : copies) TC2LI_HIP_CHECK(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, nullptr));
    DevBuf<BalmCutTask> d_task;
    TC2LI_HIP_CHECK(d_task.ensure(1));
    TC2LI_HIP_CHECK(hipMemcpyAsync(d_task.p, &task, sizeof(task), hipMemcpyHostToDevice, nullptr));
    launch_balm_cut(d_task.p, 1, task.n_points, 1 << task.table_bits, nullptr);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (info) memcpy(info, term.h_cut_result.p, 4 * sizeof(int32_t));
    if (term.h_cut_result.p[1]) { set_error("tc2li_device_lidar_planes: the kernels declined the window (%d planes found)", term.h_cut_result.p[3]); return TC2LI_ERR_INVALID; }
    const int n = term.h_cut_result.p[0], m = std::min(n, capacity);
    if (m > 0) {
        TC2LI_HIP_CHECK(hipMemcpy(clusters, term.d_clusters.p, (size_t)m * win->n_keyframes * sizeof(PlaneCluster), hipMemcpyDeviceToHost));
        TC2LI_HIP_CHECK(hipMemcpy(coe, term.d_coe.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    }
    return n;
}
