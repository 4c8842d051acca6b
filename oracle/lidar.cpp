// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
// See lidar.hpp for the reference locations each function follows.
#include "lidar.hpp"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace oracle {

// preprocess.cpp:145-166
PointVector preprocess_velodyne(const VelodynePoint* raw, int plsize, int point_filter_num, double blind,
                                float time_unit_scale) {
    PointVector pl_surf;
    if (plsize == 0) return pl_surf;
    pl_surf.reserve(plsize);
    for (int i = 0; i < plsize; i++) {
        PointXYZINormal added_pt;
        std::memset(&added_pt, 0, sizeof(added_pt));
        added_pt.pad0 = 1.0f;  // PCL_ADD_POINT4D initialises data[3] to 1
        added_pt.x = raw[i].x;
        added_pt.y = raw[i].y;
        added_pt.z = raw[i].z;
        added_pt.intensity = raw[i].intensity;
        added_pt.curvature = raw[i].time * time_unit_scale;
        if (i % point_filter_num == 0) {
            if (added_pt.x * added_pt.x + added_pt.y * added_pt.y + added_pt.z * added_pt.z > (blind * blind))
                pl_surf.push_back(added_pt);
        }
    }
    return pl_surf;
}

// pcl::VoxelGrid<PointT>::applyFilter (PCL 1.12 filters/impl/voxel_grid.hpp), all fields averaged.
PointVector voxel_grid_filter(const PointVector& in, float leaf) {
    PointVector out;
    if (in.empty()) return out;
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (const auto& p : in) {
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
        mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
    }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
                  dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT_MAX) return in;  // "leaf size too small": the filter hands the input back
    int min_b[3], max_b[3], div_b[3], mul[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)std::floor(mn[a] * inv);
        max_b[a] = (int)std::floor(mx[a] * inv);
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    std::vector<std::pair<int, int>> index;  // (voxel, point)
    index.reserve(in.size());
    for (int i = 0; i < (int)in.size(); ++i) {
        const auto& p = in[i];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        const int i0 = (int)(std::floor(p.x * inv) - (float)min_b[0]);
        const int i1 = (int)(std::floor(p.y * inv) - (float)min_b[1]);
        const int i2 = (int)(std::floor(p.z * inv) - (float)min_b[2]);
        index.emplace_back(i0 * mul[0] + i1 * mul[1] + i2 * mul[2], i);
    }
    std::sort(index.begin(), index.end());  // by voxel, then by point index (the order the sums are taken in)
    size_t first = 0;
    while (first < index.size()) {
        size_t last = first + 1;
        while (last < index.size() && index[last].first == index[first].first) ++last;
        // CentroidPoint: per-field running sums, divided by the count; the normal is re-normalised if non-zero
        float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = first; k < last; ++k) {
            const auto& p = in[index[k].second];
            s[0] += p.x; s[1] += p.y; s[2] += p.z;
            s[3] += p.normal_x; s[4] += p.normal_y; s[5] += p.normal_z;
            s[6] += p.intensity; s[7] += p.curvature;
        }
        const float n = (float)(last - first);
        PointXYZINormal o;
        std::memset(&o, 0, sizeof(o));
        o.pad0 = 1.0f;
        o.x = s[0] / n; o.y = s[1] / n; o.z = s[2] / n;
        float nx = s[3], ny = s[4], nz = s[5];
        const float nn = nx * nx + ny * ny + nz * nz;
        if (nn > 0) { const float r = std::sqrt(nn); nx /= r; ny /= r; nz /= r; }
        o.normal_x = nx; o.normal_y = ny; o.normal_z = nz;
        o.intensity = s[6] / n;
        o.curvature = s[7] / n;
        out.push_back(o);
        first = last;
    }
    return out;
}

// LidarFrontEnd.cpp:130-139
PointXYZINormal pointBodyToWorld(const PointXYZINormal& pi, const LidarState& s) {
    const double b[3] = {pi.x, pi.y, pi.z};
    double t[3], g[3];
    for (int r = 0; r < 3; ++r)
        t[r] = (s.offset_R_L_I[3 * r] * b[0] + s.offset_R_L_I[3 * r + 1] * b[1] + s.offset_R_L_I[3 * r + 2] * b[2]) + s.offset_T_L_I[r];
    for (int r = 0; r < 3; ++r) g[r] = (s.rot[3 * r] * t[0] + s.rot[3 * r + 1] * t[1] + s.rot[3 * r + 2] * t[2]) + s.pos[r];
    PointXYZINormal po;
    std::memset(&po, 0, sizeof(po));
    po.pad0 = 1.0f;
    po.x = (float)g[0]; po.y = (float)g[1]; po.z = (float)g[2];
    po.intensity = pi.intensity;
    return po;
}

// ---- k-d tree ------------------------------------------------------------------------------------------------
static inline float coord(const PointXYZINormal& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

// ikd_Tree.cpp:1750-1755
static inline float calc_dist(const PointXYZINormal& a, const PointXYZINormal& b) {
    return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
}

void KdTree::update(int n) {
    Node& nd = nodes[n];
    for (int a = 0; a < 3; ++a) nd.lo[a] = nd.hi[a] = coord(nd.p, a);
    for (int c : {nd.left, nd.right})
        if (c >= 0)
            for (int a = 0; a < 3; ++a) { nd.lo[a] = std::min(nd.lo[a], nodes[c].lo[a]); nd.hi[a] = std::max(nd.hi[a], nodes[c].hi[a]); }
}

// ikd_Tree.cpp:690-744
int KdTree::build(PointVector& s, int l, int r) {
    if (l > r) return -1;
    const int mid = (l + r) >> 1;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = l; i <= r; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], coord(s[i], a)); mx[a] = std::max(mx[a], coord(s[i], a)); }
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    std::nth_element(s.begin() + l, s.begin() + mid, s.begin() + r + 1,
                     [axis](const PointXYZINormal& a, const PointXYZINormal& b) { return coord(a, axis) < coord(b, axis); });
    const int id = (int)nodes.size();
    nodes.emplace_back();
    nodes[id].p = s[mid];
    nodes[id].axis = axis;
    const int lc = build(s, l, mid - 1);
    const int rc = build(s, mid + 1, r);
    nodes[id].left = lc;
    nodes[id].right = rc;
    update(id);
    return id;
}

void KdTree::Build(PointVector pts) {
    nodes.clear();
    nodes.reserve(pts.size());
    root = pts.empty() ? -1 : build(pts, 0, (int)pts.size() - 1);
}

// ikd_Tree.cpp:1007-1070 without re-balancing
void KdTree::Add_Point(const PointXYZINormal& p) {
    const int id = (int)nodes.size();
    nodes.emplace_back();
    nodes[id].p = p;
    if (root < 0) { nodes[id].axis = 0; root = id; update(id); return; }
    std::vector<int> path;
    int cur = root;
    for (;;) {
        path.push_back(cur);
        const int ax = nodes[cur].axis;
        int& child = coord(p, ax) < coord(nodes[cur].p, ax) ? nodes[cur].left : nodes[cur].right;
        if (child < 0) { child = id; nodes[id].axis = (ax + 1) % 3; break; }
        cur = child;
    }
    update(id);
    for (int i = (int)path.size() - 1; i >= 0; --i) update(path[i]);
}

// MANUAL_HEAP + PointType_CMP, ikd_Tree.h:93-201
struct KdTree::Heap {
    std::vector<HeapItem> h;
    int n = 0;
    explicit Heap(int cap) : h(cap) {}
    static bool less(const HeapItem& a, const HeapItem& b) {
        if (std::fabs(a.dist - b.dist) < 1e-10) return a.p.x < b.p.x;
        return a.dist < b.dist;
    }
    void pop() {
        if (!n) return;
        h[0] = h[n - 1];
        --n;
        int i = 0, l = 1;
        HeapItem tmp = h[0];
        while (l < n) {
            if (l + 1 < n && less(h[l], h[l + 1])) l++;
            if (less(tmp, h[l])) { h[i] = h[l]; i = l; l = 2 * i + 1; } else break;
        }
        h[i] = tmp;
    }
    void push(const HeapItem& it) {
        if (n >= (int)h.size()) return;
        int i = n, a = (i - 1) / 2;
        h[n] = it;
        HeapItem tmp = it;
        while (i > 0) {
            if (less(h[a], tmp)) { h[i] = h[a]; i = a; a = (i - 1) / 2; } else break;
        }
        h[i] = tmp;
        ++n;
    }
};

// ikd_Tree.cpp:1757-1776
float KdTree::box_dist(int n, const PointXYZINormal& q) const {
    if (n < 0) return INFINITY;
    const Node& nd = nodes[n];
    float d = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float c = coord(q, a);
        if (c < nd.lo[a]) d += (c - nd.lo[a]) * (c - nd.lo[a]);
        if (c > nd.hi[a]) d += (c - nd.hi[a]) * (c - nd.hi[a]);
    }
    return d;
}

// ikd_Tree.cpp:1074-1256 (max_dist = INFINITY, no deleted points, single thread)
void KdTree::search(int n, int k, const PointXYZINormal& q, Heap& hp) const {
    if (n < 0) return;
    const Node& nd = nodes[n];
    const float dist = calc_dist(q, nd.p);
    if (!nd.deleted && (hp.n < k || dist < hp.h[0].dist)) {  // ikd_Tree.cpp:1085 `if (!root->point_deleted)`
        if (hp.n >= k) hp.pop();
        hp.push(HeapItem{nd.p, dist});
    }
    const float dl = box_dist(nd.left, q), dr = box_dist(nd.right, q);
    if (hp.n < k || (dl < hp.h[0].dist && dr < hp.h[0].dist)) {
        if (dl <= dr) {
            search(nd.left, k, q, hp);
            if (hp.n < k || dr < hp.h[0].dist) search(nd.right, k, q, hp);
        } else {
            search(nd.right, k, q, hp);
            if (hp.n < k || dl < hp.h[0].dist) search(nd.left, k, q, hp);
        }
    } else {
        if (dl < hp.h[0].dist) search(nd.left, k, q, hp);
        if (dr < hp.h[0].dist) search(nd.right, k, q, hp);
    }
}

// ikd_Tree.cpp:426-461
void KdTree::Nearest_Search(const PointXYZINormal& q, int k, PointVector& near, std::vector<float>& sqdist) const {
    Heap hp(2 * k);
    search(root, k, q, hp);
    const int found = std::min(k, hp.n);
    near.assign(found, PointXYZINormal());
    sqdist.assign(found, 0.f);
    for (int i = found - 1; i >= 0; --i) { near[i] = hp.h[0].p; sqdist[i] = hp.h[0].dist; hp.pop(); }
}

// ---- plane fit ----------------------------------------------------------------------------------------------
// Least-squares solution of the 5x3 system A x = b by Householder QR with column pivoting (the method behind
// Eigen's colPivHouseholderQr().solve()), single precision, sums taken in index order.
static void qr_solve_5x3(float A[5][3], float b[5], float x[3]) {
    const int R = 5, C = 3;
    float normU[3], normD[3];
    for (int j = 0; j < C; ++j) {
        float s = 0;
        for (int i = 0; i < R; ++i) s += A[i][j] * A[i][j];
        normU[j] = normD[j] = std::sqrt(s);
    }
    float maxn = std::max(normU[0], std::max(normU[1], normU[2]));
    const float thr_helper = (maxn * FLT_EPSILON) * (maxn * FLT_EPSILON) / (float)R;
    const float downdate_thr = std::sqrt(FLT_EPSILON);
    int perm[3] = {0, 1, 2};
    int nonzero = C;
    float tau[3] = {0, 0, 0};
    for (int k = 0; k < C; ++k) {
        int big = k;
        for (int j = k + 1; j < C; ++j) if (normU[j] > normU[big]) big = j;
        const float big_sq = normU[big] * normU[big];
        if (nonzero == C && big_sq < thr_helper * (float)(R - k)) nonzero = k;
        if (big != k) {
            for (int i = 0; i < R; ++i) std::swap(A[i][k], A[i][big]);
            std::swap(normU[k], normU[big]);
            std::swap(normD[k], normD[big]);
            std::swap(perm[k], perm[big]);
        }
        // Householder vector of column k below the diagonal
        float tail = 0;
        for (int i = k + 1; i < R; ++i) tail += A[i][k] * A[i][k];
        const float c0 = A[k][k];
        float beta;
        if (tail <= FLT_MIN) {
            tau[k] = 0;
            beta = c0;
            for (int i = k + 1; i < R; ++i) A[i][k] = 0;
        } else {
            beta = std::sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < R; ++i) A[i][k] = A[i][k] / (c0 - beta);
            tau[k] = (beta - c0) / beta;
        }
        A[k][k] = beta;
        // apply H_k = I - tau v v^T (v = [1, essential]) to the remaining columns and to b
        if (tau[k] != 0) {
            for (int j = k + 1; j < C; ++j) {
                float t = 0;
                for (int i = k + 1; i < R; ++i) t += A[i][k] * A[i][j];
                t += A[k][j];
                A[k][j] -= tau[k] * t;
                for (int i = k + 1; i < R; ++i) A[i][j] -= tau[k] * A[i][k] * t;
            }
        }
        for (int j = k + 1; j < C; ++j) {
            if (normU[j] != 0) {
                float temp = std::fabs(A[k][j]) / normU[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0 ? 0 : temp;
                const float r = normU[j] / normD[j];
                const float temp2 = temp * (r * r);
                if (temp2 <= downdate_thr) {
                    float s = 0;
                    for (int i = k + 1; i < R; ++i) s += A[i][j] * A[i][j];
                    normD[j] = std::sqrt(s);
                    normU[j] = normD[j];
                } else {
                    normU[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // c = Q^T b
    for (int k = 0; k < nonzero; ++k) {
        if (tau[k] == 0) continue;
        float t = 0;
        for (int i = k + 1; i < R; ++i) t += A[i][k] * b[i];
        t += b[k];
        b[k] -= tau[k] * t;
        for (int i = k + 1; i < R; ++i) b[i] -= tau[k] * A[i][k] * t;
    }
    // back substitution on the leading nonzero x nonzero block of R
    float c[3] = {0, 0, 0};
    for (int i = nonzero - 1; i >= 0; --i) {
        float s = b[i];
        for (int j = i + 1; j < nonzero; ++j) s -= A[i][j] * c[j];
        c[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0;
    for (int i = 0; i < nonzero; ++i) x[perm[i]] = c[i];
}

// LidarFrontEnd.cpp:964-997
bool EstiPlane(float pca_result[4], const PointVector& point, float threshold) {
    float A[5][3], b[5], nv[3];
    for (int j = 0; j < 5; j++) {
        A[j][0] = point[j].x; A[j][1] = point[j].y; A[j][2] = point[j].z;
        b[j] = -1.0f;
    }
    qr_solve_5x3(A, b, nv);
    const float n = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    pca_result[0] = nv[0] / n;
    pca_result[1] = nv[1] / n;
    pca_result[2] = nv[2] / n;
    pca_result[3] = (float)(1.0 / n);
    for (int j = 0; j < 5; j++)
        if (std::fabs(pca_result[0] * point[j].x + pca_result[1] * point[j].y + pca_result[2] * point[j].z + pca_result[3]) > threshold)
            return false;
    return true;
}

// LidarFrontEnd.cpp:999-1073
FeatureExtraction feature_extraction(const PointVector& feats_down_body, const LidarState& st, const KdTree& tree) {
    const int N = (int)feats_down_body.size();
    FeatureExtraction fe;
    fe.feats_down_world.resize(N);
    fe.Nearest_Points.resize(N);
    fe.point_selected_surf.assign(N, 0);
    PointXYZINormal blank;
    std::memset(&blank, 0, sizeof(blank));
    blank.pad0 = 1.0f;  // pcl points are constructed with data[3] = 1
    fe.normvec.assign(N, blank);
    for (int i = 0; i < N; i++) {
        const PointXYZINormal& point_body = feats_down_body[i];
        PointXYZINormal& point_world = fe.feats_down_world[i];
        const double pb[3] = {point_body.x, point_body.y, point_body.z};
        point_world = pointBodyToWorld(point_body, st);
        std::vector<float> pointSearchSqDis(5);
        auto& points_near = fe.Nearest_Points[i];
        tree.Nearest_Search(point_world, 5, points_near, pointSearchSqDis);
        bool sel = points_near.size() < 5 ? false : (pointSearchSqDis[4] > 5 ? false : true);
        if (!sel) continue;
        float pabcd[4];
        if (EstiPlane(pabcd, points_near, 0.1f)) {
            const float pd2 = pabcd[0] * point_world.x + pabcd[1] * point_world.y + pabcd[2] * point_world.z + pabcd[3];
            const double pnorm = std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
            const float s = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(pnorm));
            if (s > 0.9) {
                fe.point_selected_surf[i] = 1;
                fe.normvec[i].x = pabcd[0]; fe.normvec[i].y = pabcd[1]; fe.normvec[i].z = pabcd[2];
                fe.normvec[i].intensity = pd2;
            }
        }
    }
    for (int i = 0; i < N; i++)
        if (fe.point_selected_surf[i]) {
            fe.laserCloudOri.push_back(feats_down_body[i]);
            fe.corr_normvect.push_back(fe.normvec[i]);
            fe.effct_feat_num++;
        }
    return fe;
}

// ---- b2: ImuProcess::UndistortPcl -------------------------------------------------------------------------------------
namespace {
void so3_Exp(const double w[3], double dt, double R[9]) {  // so3_math.h:63-85
    const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (n > 0.0000001) {
        const double a[3] = {w[0] / n, w[1] / n, w[2] / n};
        const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
        const double ang = n * dt, s = std::sin(ang), c1 = 1.0 - std::cos(ang);
        double cK[9], cKK[9];
        for (int k = 0; k < 9; ++k) cK[k] = c1 * K[k];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cKK[3 * r + c] = cK[3 * r] * K[c] + cK[3 * r + 1] * K[3 + c] + cK[3 * r + 2] * K[6 + c];
        for (int k = 0; k < 9; ++k) R[k] = (I[k] + s * K[k]) + cKK[k];
    } else {
        for (int k = 0; k < 9; ++k) R[k] = I[k];
    }
}
void mat3(const double* a, const double* b, double* o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c]; }
void matv(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[3 * r] * v[0] + a[3 * r + 1] * v[1] + a[3 * r + 2] * v[2]; }
void mattv(const double* a, const double* v, double* o) { for (int r = 0; r < 3; ++r) o[r] = a[r] * v[0] + a[3 + r] * v[1] + a[6 + r] * v[2]; }
bool time_list(const PointXYZINormal& x, const PointXYZINormal& y) { return x.curvature < y.curvature; }
}  // namespace

void UndistortPcl(PointVector& pcl, const std::vector<Pose6D>& IMUpose, const LidarState& end) {
    std::sort(pcl.begin(), pcl.end(), time_list);
    if (pcl.empty() || IMUpose.empty()) return;
    auto it_pcl = pcl.end() - 1;
    for (auto it_kp = IMUpose.end() - 1; it_kp != IMUpose.begin(); it_kp--) {
        auto head = it_kp - 1;
        auto tail = it_kp;
        for (; it_pcl->curvature / double(1000) > head->offset_time; it_pcl--) {
            const double dt = it_pcl->curvature / double(1000) - head->offset_time;
            double E[9], R_i[9];
            so3_Exp(tail->gyr, dt, E);
            mat3(head->rot, E, R_i);
            const double P_i[3] = {it_pcl->x, it_pcl->y, it_pcl->z};
            double T_ei[3], a[3], b[3], c[3], d[3];
            for (int k = 0; k < 3; ++k) T_ei[k] = ((head->pos[k] + head->vel[k] * dt) + ((0.5 * tail->acc[k]) * dt) * dt) - end.pos[k];
            matv(end.offset_R_L_I, P_i, a);
            for (int k = 0; k < 3; ++k) a[k] += end.offset_T_L_I[k];
            matv(R_i, a, b);
            for (int k = 0; k < 3; ++k) b[k] += T_ei[k];
            mattv(end.rot, b, c);
            for (int k = 0; k < 3; ++k) c[k] -= end.offset_T_L_I[k];
            mattv(end.offset_R_L_I, c, d);
            it_pcl->x = (float)d[0]; it_pcl->y = (float)d[1]; it_pcl->z = (float)d[2];
            if (it_pcl == pcl.begin()) break;
        }
    }
}

std::vector<Pose6D> ForwardPropagate(ImuState& st, const std::vector<ImuMeas>& v_imu, double pcl_beg_time, double pcl_end_time,
                                     double last_lidar_end_time, double acc_scale, const double acc_s_last[3], const double angvel_last[3]) {
    std::vector<Pose6D> out;
    auto save = [&](double t, const double* acc, const double* gyr) {
        Pose6D p;
        p.offset_time = t;
        std::memcpy(p.acc, acc, 24); std::memcpy(p.gyr, gyr, 24); std::memcpy(p.vel, st.vel, 24); std::memcpy(p.pos, st.pos, 24); std::memcpy(p.rot, st.rot, 72);
        out.push_back(p);
    };
    save(0.0, acc_s_last, angvel_last);
    auto predict = [&](double dt, const double* acc, const double* gyr) {  // state part of esekf::predict
        double w[3], am[3], Ra[3], E[9], Rn[9];
        for (int k = 0; k < 3; ++k) { w[k] = gyr[k] - st.bg[k]; am[k] = acc[k] - st.ba[k]; }
        matv(st.rot, am, Ra);
        for (int k = 0; k < 3; ++k) st.pos[k] += st.vel[k] * dt;
        const double wdt[3] = {w[0] * dt, w[1] * dt, w[2] * dt};
        so3_Exp(wdt, 1.0, E);
        mat3(st.rot, E, Rn);
        for (int k = 0; k < 3; ++k) st.vel[k] += (Ra[k] + st.grav[k]) * dt;
        std::memcpy(st.rot, Rn, sizeof(Rn));
    };
    double acc_avr[3] = {0, 0, 0}, angvel_avr[3] = {0, 0, 0};
    for (size_t i = 0; i + 1 < v_imu.size(); ++i) {
        const ImuMeas& head = v_imu[i];
        const ImuMeas& tail = v_imu[i + 1];
        if (tail.t < last_lidar_end_time) continue;
        for (int k = 0; k < 3; ++k) { angvel_avr[k] = 0.5 * (head.gyr[k] + tail.gyr[k]); acc_avr[k] = 0.5 * (head.acc[k] + tail.acc[k]) * acc_scale; }
        const double dt = head.t < last_lidar_end_time ? tail.t - last_lidar_end_time : tail.t - head.t;
        predict(dt, acc_avr, angvel_avr);
        double gl[3], al[3], am[3];
        for (int k = 0; k < 3; ++k) { gl[k] = angvel_avr[k] - st.bg[k]; am[k] = acc_avr[k] - st.ba[k]; }
        matv(st.rot, am, al);
        for (int k = 0; k < 3; ++k) al[k] += st.grav[k];
        save(tail.t - pcl_beg_time, al, gl);
    }
    const double imu_end_time = v_imu.back().t;
    const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;
    predict(note * (pcl_end_time - imu_end_time), acc_avr, angvel_avr);
    return out;
}

// ---- map maintenance ---------------------------------------------------------------------------------------------------
namespace {
inline bool same_point(const PointXYZINormal& a, const PointXYZINormal& b) {  // ikd_Tree.cpp: EPSS 1e-6
    return std::fabs(a.x - b.x) < 1e-6f && std::fabs(a.y - b.y) < 1e-6f && std::fabs(a.z - b.z) < 1e-6f;
}
inline bool in_box(const PointXYZINormal& p, const BoxPointType& b) {
    return b.vertex_min[0] <= p.x && b.vertex_max[0] > p.x && b.vertex_min[1] <= p.y && b.vertex_max[1] > p.y && b.vertex_min[2] <= p.z && b.vertex_max[2] > p.z;
}
}  // namespace

// ---- the same operations on the tree (ikd-Tree keeps deleted points as flagged nodes until a rebuild) --------------------
// One walk serves Search_by_range (storage != NULL) and Delete_by_range (n_del != NULL): prune on the node box, test the node's
// point against [min, max), descend.  The node boxes are those of all points ever inserted below (a superset of the live ones).
void KdTree::range_search(int n, const BoxPointType& box, PointVector* storage, int* n_del) {
    if (n < 0) return;
    Node& nd = nodes[n];
    for (int a = 0; a < 3; ++a) if (box.vertex_max[a] <= nd.lo[a] || box.vertex_min[a] > nd.hi[a]) return;
    if (!nd.deleted && in_box(nd.p, box)) {
        if (storage) storage->push_back(nd.p);
        if (n_del) { nd.deleted = true; ++n_deleted; ++*n_del; }
    }
    range_search(nd.left, box, storage, n_del);
    range_search(nd.right, box, storage, n_del);
}
void KdTree::Search_by_range(const BoxPointType& box, PointVector& storage) const {
    const_cast<KdTree*>(this)->range_search(root, box, &storage, nullptr);
}
int KdTree::Delete_by_range(const BoxPointType& box) {
    int n = 0;
    range_search(root, box, nullptr, &n);
    return n;
}
int KdTree::Delete_Point_Boxes(const std::vector<BoxPointType>& boxes) {
    int n = 0;
    for (const BoxPointType& b : boxes) n += Delete_by_range(b);
    return n;
}
PointVector KdTree::valid_points() const {
    PointVector out;
    out.reserve(valid_size());
    for (const Node& nd : nodes) if (!nd.deleted) out.push_back(nd.p);
    return out;
}
// ikd_Tree.cpp:478-584 with Rebuild_Ptr == nullptr (no background rebuild): per point the voxel box, the stored points inside it,
// the one nearest the voxel centre (the new point when it is nearer), and -- when the box held more than one point or the new
// point wins -- delete the box and insert the winner.
int KdTree::Add_Points(const PointVector& PointToAdd, bool downsample_on, float downsample_size) {
    int tmp_counter = 0;
    PointVector Downsample_Storage;
    for (size_t i = 0; i < PointToAdd.size(); i++) {
        if (!downsample_on) { Add_Point(PointToAdd[i]); continue; }
        BoxPointType Box;
        const float c[3] = {PointToAdd[i].x, PointToAdd[i].y, PointToAdd[i].z};
        PointXYZINormal mid = PointToAdd[i];
        float m[3];
        for (int k = 0; k < 3; ++k) {
            Box.vertex_min[k] = (float)(std::floor(c[k] / downsample_size) * downsample_size);
            Box.vertex_max[k] = Box.vertex_min[k] + downsample_size;
            m[k] = (float)(Box.vertex_min[k] + (Box.vertex_max[k] - Box.vertex_min[k]) / 2.0);
        }
        mid.x = m[0]; mid.y = m[1]; mid.z = m[2];
        Downsample_Storage.clear();
        Search_by_range(Box, Downsample_Storage);
        float min_dist = calc_dist(PointToAdd[i], mid);
        PointXYZINormal result = PointToAdd[i];
        for (const PointXYZINormal& q : Downsample_Storage) {
            const float d = calc_dist(q, mid);
            if (d < min_dist) { min_dist = d; result = q; }
        }
        if (Downsample_Storage.size() > 1 || same_point(PointToAdd[i], result)) {
            if (!Downsample_Storage.empty()) Delete_by_range(Box);
            Add_Point(result);
            tmp_counter++;
        }
    }
    return tmp_counter;
}

int MapPoints::Add_Points(const PointVector& PointToAdd, bool downsample_on, float downsample_size) {
    int tmp_counter = 0;
    for (size_t i = 0; i < PointToAdd.size(); i++) {
        if (!downsample_on) { pts.push_back(PointToAdd[i]); continue; }
        BoxPointType Box;
        const float c[3] = {PointToAdd[i].x, PointToAdd[i].y, PointToAdd[i].z};
        PointXYZINormal mid = PointToAdd[i];
        float m[3];
        for (int k = 0; k < 3; ++k) {
            Box.vertex_min[k] = (float)(std::floor(c[k] / downsample_size) * downsample_size);
            Box.vertex_max[k] = Box.vertex_min[k] + downsample_size;
            m[k] = (float)(Box.vertex_min[k] + (Box.vertex_max[k] - Box.vertex_min[k]) / 2.0);
        }
        mid.x = m[0]; mid.y = m[1]; mid.z = m[2];
        std::vector<size_t> storage;
        for (size_t k = 0; k < pts.size(); ++k) if (in_box(pts[k], Box)) storage.push_back(k);
        float min_dist = calc_dist(PointToAdd[i], mid);
        PointXYZINormal result = PointToAdd[i];
        for (size_t k : storage) {
            const float d = calc_dist(pts[k], mid);
            if (d < min_dist) { min_dist = d; result = pts[k]; }
        }
        if (storage.size() > 1 || same_point(PointToAdd[i], result)) {
            if (!storage.empty()) {
                PointVector kept;
                kept.reserve(pts.size());
                for (size_t k = 0; k < pts.size(); ++k) if (!in_box(pts[k], Box)) kept.push_back(pts[k]);
                pts.swap(kept);
            }
            pts.push_back(result);
            tmp_counter++;
        }
    }
    return tmp_counter;
}

int MapPoints::Delete_Point_Boxes(const std::vector<BoxPointType>& boxes) {
    int deleted = 0;
    for (const BoxPointType& b : boxes) {
        PointVector kept;
        kept.reserve(pts.size());
        for (const PointXYZINormal& p : pts) { if (in_box(p, b)) ++deleted; else kept.push_back(p); }
        pts.swap(kept);
    }
    return deleted;
}

MapIncrement map_incremental_lists(const PointVector& feats_down_body, const LidarState& st, const std::vector<PointVector>& Nearest_Points,
                                   bool flg_EKF_inited, double filter_size_map_min) {
    MapIncrement out;
    const int NUM_MATCH_POINTS = 5;
    for (size_t i = 0; i < feats_down_body.size(); i++) {
        const PointXYZINormal pw = pointBodyToWorld(feats_down_body[i], st);
        if (!Nearest_Points[i].empty() && flg_EKF_inited) {
            const PointVector& points_near = Nearest_Points[i];
            bool need_add = true;
            PointXYZINormal mid_point = pw;
            mid_point.x = (float)(std::floor(pw.x / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
            mid_point.y = (float)(std::floor(pw.y / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
            mid_point.z = (float)(std::floor(pw.z / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
            const float dist = calc_dist(pw, mid_point);
            if (std::fabs(points_near[0].x - mid_point.x) > 0.5 * filter_size_map_min && std::fabs(points_near[0].y - mid_point.y) > 0.5 * filter_size_map_min &&
                std::fabs(points_near[0].z - mid_point.z) > 0.5 * filter_size_map_min) {
                out.PointNoNeedDownsample.push_back(pw);
                continue;
            }
            for (int readd_i = 0; readd_i < NUM_MATCH_POINTS; readd_i++) {
                if ((int)points_near.size() < NUM_MATCH_POINTS) break;
                if (calc_dist(points_near[readd_i], mid_point) < dist) { need_add = false; break; }
            }
            if (need_add) out.PointToAdd.push_back(pw);
        } else {
            out.PointToAdd.push_back(pw);
        }
    }
    return out;
}

std::vector<BoxPointType> lasermap_fov_segment(LocalMapBox& lm, const double pos_LiD[3], double cube_len, double DET_RANGE, float MOV_THRESHOLD) {
    std::vector<BoxPointType> cub_needrm;
    if (!lm.initialized) {
        for (int i = 0; i < 3; i++) {
            lm.box.vertex_min[i] = (float)(pos_LiD[i] - cube_len / 2.0);
            lm.box.vertex_max[i] = (float)(pos_LiD[i] + cube_len / 2.0);
        }
        lm.initialized = true;
        return cub_needrm;
    }
    float dist_to_map_edge[3][2];
    bool need_move = false;
    for (int i = 0; i < 3; i++) {
        dist_to_map_edge[i][0] = (float)std::fabs(pos_LiD[i] - lm.box.vertex_min[i]);
        dist_to_map_edge[i][1] = (float)std::fabs(pos_LiD[i] - lm.box.vertex_max[i]);
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * DET_RANGE || dist_to_map_edge[i][1] <= MOV_THRESHOLD * DET_RANGE) need_move = true;
    }
    if (!need_move) return cub_needrm;
    BoxPointType New = lm.box, tmp;
    const float mov_dist = (float)std::max((cube_len - 2.0 * MOV_THRESHOLD * DET_RANGE) * 0.5 * 0.9, double(DET_RANGE * (MOV_THRESHOLD - 1)));
    for (int i = 0; i < 3; i++) {
        tmp = lm.box;
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * DET_RANGE) {
            New.vertex_max[i] -= mov_dist;
            New.vertex_min[i] -= mov_dist;
            tmp.vertex_min[i] = lm.box.vertex_max[i] - mov_dist;
            cub_needrm.push_back(tmp);
        } else if (dist_to_map_edge[i][1] <= MOV_THRESHOLD * DET_RANGE) {
            New.vertex_max[i] += mov_dist;
            New.vertex_min[i] += mov_dist;
            tmp.vertex_max[i] = lm.box.vertex_min[i] + mov_dist;
            cub_needrm.push_back(tmp);
        }
    }
    lm.box = New;
    return cub_needrm;
}

}  // namespace oracle
