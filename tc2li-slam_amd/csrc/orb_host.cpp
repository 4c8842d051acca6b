// Host orchestration of the gfx950 ORB extractor behind the C ABI of include/tc2li_hip.h.
// Mirrors TC2LI_SLAM::ORBextractor (SF/src/ORBextractor.cc): ctor tables (:383-443), ComputePyramid (:1143),
// ComputeKeyPointsOctTree (:755) and operator() (:1060).  Every stage is a kernel, the quadtree distribution (:529-753) included
// (quadtree_kernels.hip: one workgroup per (image, level), the reference's list order reproduced); the host queues the whole call,
// waits once and assembles the caller's arrays in the reference's output order.  The blur runs on a second stream.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "common.hpp"
#include "orb_device.hpp"
#include "quadtree.hpp"
#include "orb_handle.hpp"

using namespace tc2li;

namespace {

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeilD(double v) { int i = (int)v; return i + (i < v); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

namespace {

// cv::resize INTER_LINEAR coefficient tables (OpenCV imgproc resize(): 11-bit fixed point)
void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int>& xofs, std::vector<short>& ialpha,
                         std::vector<int>& yofs, std::vector<short>& ibeta) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    auto sat = [](float v) { int r = cvRoundF(v); return (short)std::min(32767, std::max(-32768, r)); };
    xofs.resize(dw); ialpha.resize(2 * dw); yofs.resize(dh); ibeta.resize(2 * dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorD(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorD(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat(fy * 2048);
    }
}

int setup_geometry(tc2li_orb* o, int w, int h) {
    if (o->cur_w == w && o->cur_h == h) return TC2LI_OK;
    const int L = o->prm.nlevels, M = o->max_images;
    o->geom.assign(L, LevelGeom());
    o->cells.clear();
    o->max_cell_w = o->max_cell_h = 0;
    std::vector<int> level_cell_begin(L + 1, 0), level_dense_off(L, 0);
    int slab = 0;
    for (int l = 0; l < L; ++l) {
        LevelGeom& g = o->geom[l];
        g.w = cvRoundF((float)w * o->inv_scale[l]);   // SF/src/ORBextractor.cc:1148
        g.h = cvRoundF((float)h * o->inv_scale[l]);
        g.pitch = align_up(g.w, 64);
        g.img_stride = (size_t)align_up(g.pitch * g.h, 256);
        // cell grid, SF/src/ORBextractor.cc:763-797
        g.min_bx = g.min_by = kMinBorder;
        g.max_bx = g.w - kEdgeThreshold + 3;
        g.max_by = g.h - kEdgeThreshold + 3;
        g.cell_begin = (int)o->cells.size();
        g.dense_off = slab;
        const float width = (float)(g.max_bx - g.min_bx), height = (float)(g.max_by - g.min_by);
        const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
        if (nCols > 0 && nRows > 0 && g.w > 2 * kEdgeThreshold && g.h > 2 * kEdgeThreshold) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; ++i) {
                const float iniY = (float)(g.min_by + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= g.max_by - 3) continue;
                if (maxY > g.max_by) maxY = (float)g.max_by;
                for (int j = 0; j < nCols; ++j) {
                    const float iniX = (float)(g.min_bx + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= g.max_bx - 6) continue;
                    if (maxX > g.max_bx) maxX = (float)g.max_bx;
                    FastCell c{};
                    c.level = (int16_t)l;
                    c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY;
                    c.w = (int16_t)((int)maxX - (int)iniX); c.h = (int16_t)((int)maxY - (int)iniY);
                    if (c.w > kFastTilePitch || c.h > kFastTileH) {
                        set_error("FAST cell window %dx%d exceeds the kernel tile", c.w, c.h);
                        return TC2LI_ERR_INVALID;
                    }
                    o->max_cell_w = std::max(o->max_cell_w, (int)c.w); o->max_cell_h = std::max(o->max_cell_h, (int)c.h);
                    const int ew = std::max(c.w - 6, 0), eh = std::max(c.h - 6, 0);
                    c.slab_off = slab;
                    c.slab_cap = ((ew + 1) / 2) * ((eh + 1) / 2);  // 3x3 strict maxima: at most one per 2x2 block
                    slab += c.slab_cap;
                    o->cells.push_back(c);
                }
            }
        }
        g.cell_end = (int)o->cells.size();
        if (g.cell_end - g.cell_begin >= kMaxCellsPerLevel) {
            set_error("too many FAST cells on level %d", l);
            return TC2LI_ERR_INVALID;
        }
        level_cell_begin[l] = g.cell_begin;
        level_dense_off[l] = g.dense_off;
        if (g.w >= 4096 + kMinBorder || g.h >= 4096 + kMinBorder) {
            set_error("image too large for the 12-bit candidate packing");
            return TC2LI_ERR_INVALID;
        }
    }
    level_cell_begin[L] = (int)o->cells.size();
    {
        std::vector<int> ids, large;
        for (int k = 0; k < (int)o->cells.size(); ++k) (o->cells[k].w <= 48 && o->cells[k].h <= 48 ? ids : large).push_back(k);
        o->n_small_cells = (int)ids.size(); o->n_large_cells = (int)large.size();
        ids.insert(ids.end(), large.begin(), large.end());
        if (ids.empty()) ids.push_back(0);
        TC2LI_HIP_CHECK(o->d_cell_ids.upload(ids));
    }
    o->slab_per_image = align_up(std::max(slab, 1), 64);
    o->kp_cap_per_image = o->prm.nfeatures + 4 * L;
    // one distribution job per (image, level): the work space is sized for the level's candidate capacity, the node arrays for the
    // largest list the walk can reach (N + 3 after the closing division, or the 4 * nIni children of the first round)
    std::vector<QuadJob> jobs((size_t)M * L);
    size_t scratch_per_image = 0, picked_per_image = 0;
    {
        std::vector<size_t> scratch_off(L), picked_off(L);
        std::vector<int> max_keys(L), max_nodes(L);
        for (int l = 0; l < L; ++l) {
            const LevelGeom& g = o->geom[l];
            max_keys[l] = (l + 1 < L ? o->geom[l + 1].dense_off : slab) - g.dense_off;
            if (max_keys[l] >= (1 << 20)) { set_error("level %d holds more FAST candidates than the distribution kernel packs", l); return TC2LI_ERR_INVALID; }
            const int bw = g.max_bx - g.min_bx, bh = g.max_by - g.min_by;
            const int n_ini = bh > 0 && bw > 0 ? (int)std::round(static_cast<float>(bw) / bh) : 0;
            max_nodes[l] = std::max(o->features_per_level[l] + 3, 4 * std::max(n_ini, 1)) + 8;
            scratch_off[l] = scratch_per_image; scratch_per_image += quadtree_scratch_bytes(max_keys[l], max_nodes[l]);
            picked_off[l] = picked_per_image; picked_per_image += (size_t)max_nodes[l];
        }
        for (int i = 0; i < M; ++i)
            for (int l = 0; l < L; ++l) {
                const LevelGeom& g = o->geom[l];
                QuadJob& j = jobs[(size_t)i * L + l];
                j.cand_off = (int64_t)i * o->slab_per_image + g.dense_off;
                j.scratch_off = (int64_t)((size_t)i * scratch_per_image + scratch_off[l]);
                j.out_off = (int64_t)((size_t)i * picked_per_image + picked_off[l]);
                j.count_idx = i * L + l;
                j.out_cap = max_nodes[l]; j.max_keys = max_keys[l]; j.max_nodes = max_nodes[l];
                j.min_x = g.min_bx; j.max_x = g.max_bx; j.min_y = g.min_by; j.max_y = g.max_by;
                j.n_target = o->features_per_level[l]; j.pad_ = 0;
            }
    }

    // device allocations
    TC2LI_HIP_CHECK(o->d_level0.alloc((size_t)M * o->geom[0].img_stride));
    for (int l = 0; l < L; ++l) {
        if (l > 0) TC2LI_HIP_CHECK(o->d_levels[l].alloc((size_t)M * o->geom[l].img_stride));
        TC2LI_HIP_CHECK(o->d_blur[l].alloc((size_t)M * o->geom[l].img_stride));
        if (l > 0) {
            std::vector<int> xofs, yofs;
            std::vector<short> ia, ib;
            build_resize_tables(o->geom[l - 1].w, o->geom[l - 1].h, o->geom[l].w, o->geom[l].h, xofs, ia, yofs, ib);
            TC2LI_HIP_CHECK(o->d_xofs[l].upload(xofs));
            TC2LI_HIP_CHECK(o->d_yofs[l].upload(yofs));
            TC2LI_HIP_CHECK(o->d_ialpha[l].upload(ia));
            TC2LI_HIP_CHECK(o->d_ibeta[l].upload(ib));
        }
    }
    TC2LI_HIP_CHECK(o->d_cells.upload(o->cells));
    TC2LI_HIP_CHECK(o->d_level_cell_begin.upload(level_cell_begin));
    TC2LI_HIP_CHECK(o->d_level_dense_off.upload(level_dense_off));
    TC2LI_HIP_CHECK(o->d_cell_counts.alloc((size_t)M * std::max<size_t>(o->cells.size(), 1)));
    TC2LI_HIP_CHECK(o->d_slab.alloc((size_t)M * o->slab_per_image));
    TC2LI_HIP_CHECK(o->d_dense.alloc((size_t)M * o->slab_per_image));
    TC2LI_HIP_CHECK(o->d_jobs.upload(jobs));
    TC2LI_HIP_CHECK(o->d_qscratch.alloc((size_t)M * scratch_per_image));
    TC2LI_HIP_CHECK(o->d_picked.alloc((size_t)M * picked_per_image));
    TC2LI_HIP_CHECK(o->d_picked_count.alloc((size_t)M * L));
    TC2LI_HIP_CHECK(o->d_level_counts.alloc((size_t)M * L));
    TC2LI_HIP_CHECK(o->d_kps.alloc((size_t)M * o->kp_cap_per_image));
    TC2LI_HIP_CHECK(o->d_nkp.alloc((size_t)M));
    TC2LI_HIP_CHECK(o->d_status.alloc(1));
    TC2LI_HIP_CHECK(o->h_nkp.alloc((size_t)M));
    TC2LI_HIP_CHECK(o->h_status.alloc(1));
    TC2LI_HIP_CHECK(o->d_mkeys.alloc((size_t)M * o->kp_cap_per_image));
    TC2LI_HIP_CHECK(o->d_desc.alloc((size_t)M * o->kp_cap_per_image * 32));
    TC2LI_HIP_CHECK(o->d_angles.alloc((size_t)M * o->kp_cap_per_image));
    TC2LI_HIP_CHECK(o->h_level_counts.alloc((size_t)M * L));
    TC2LI_HIP_CHECK(o->h_kps.alloc((size_t)M * o->kp_cap_per_image));
    TC2LI_HIP_CHECK(o->h_angles.alloc((size_t)M * o->kp_cap_per_image));
    TC2LI_HIP_CHECK(o->h_desc.alloc((size_t)M * o->kp_cap_per_image * 32));
    o->cur_w = w;
    o->cur_h = h;
    return TC2LI_OK;
}

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return -1.f; }
    return ms;
}

}  // namespace

extern "C" {

int tc2li_orb_create(const tc2li_orb_params* p, int max_width, int max_height, int max_images, tc2li_orb** out) {
    if (!p || !out || max_width <= 0 || max_height <= 0 || max_images <= 0 || p->nlevels < 1 ||
        p->nlevels > kMaxLevels || p->nfeatures < 1 || !(p->scale_factor > 1.f)) {
        set_error("tc2li_orb_create: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    std::unique_ptr<tc2li_orb> o(new tc2li_orb());
    o->prm = *p;
    if (const char* taps = getenv("TC2LI_GAUSS_TAPS")) o->gauss_rounded_taps = strcmp(taps, "rounded") == 0;
    o->max_w = max_width; o->max_h = max_height; o->max_images = max_images;
    const int L = p->nlevels;
    // SF/src/ORBextractor.cc:388-419; the member scaleFactor is a double holding the float argument
    const double scaleFactor = (double)p->scale_factor;
    o->scale.resize(L); o->sigma2.resize(L); o->inv_scale.resize(L); o->inv_sigma2.resize(L);
    o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        o->scale[i] = (float)(o->scale[i - 1] * scaleFactor);
        o->sigma2[i] = o->scale[i] * o->scale[i];
    }
    for (int i = 0; i < L; i++) {
        o->inv_scale[i] = 1.0f / o->scale[i];
        o->inv_sigma2[i] = 1.0f / o->sigma2[i];
    }
    o->features_per_level.resize(L);
    const float factor = (float)(1.0f / scaleFactor);
    float desired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        o->features_per_level[l] = cvRoundF(desired);
        sum += o->features_per_level[l];
        desired *= factor;
    }
    o->features_per_level[L - 1] = std::max(p->nfeatures - sum, 0);
    // umax, :427-442
    {
        const int HP = 15;
        int v, v0, vmax = cvFloorD(HP * sqrt(2.f) / 2 + 1), vmin = cvCeilD(HP * sqrt(2.f) / 2);
        const double hp2 = HP * HP;
        for (v = 0; v <= vmax; ++v) o->umax[v] = cvRoundD(sqrt(hp2 - v * v));
        for (v = HP, v0 = 0; v >= vmin; --v) {
            while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
            o->umax[v] = v0;
            ++v0;
        }
    }
    TC2LI_HIP_CHECK(upload_umax(o->umax));
    for (int i = 0; i < L; ++i) { o->scale_tab.scale[i] = o->scale[i]; o->scale_tab.inv_scale[i] = o->inv_scale[i]; }
    o->d_levels.resize(L); o->d_blur.resize(L);
    o->d_xofs.resize(L); o->d_yofs.resize(L); o->d_ialpha.resize(L); o->d_ibeta.resize(L);
    TC2LI_HIP_CHECK(hipStreamCreateWithFlags(&o->side_stream, hipStreamNonBlocking));
    TC2LI_HIP_CHECK(hipStreamCreateWithFlags(&o->chunk_stream, hipStreamNonBlocking));
    TC2LI_HIP_CHECK(hipEventCreateWithFlags(&o->ev_fork, hipEventDisableTiming));
    for (auto& e : o->ev) TC2LI_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventBlockingSync));  // timed, and waited on without spinning
    int rc = setup_geometry(o.get(), max_width, max_height);
    if (rc != TC2LI_OK) return rc;
    *out = o.release();
    return TC2LI_OK;
}

void tc2li_orb_destroy(tc2li_orb* orb) {
    if (!orb) return;
    tc2li::stereo_release_workspace(orb);
    delete orb;
}

int tc2li_orb_levels(const tc2li_orb* o) { return o ? o->prm.nlevels : TC2LI_ERR_INVALID; }

int tc2li_orb_scale_factors(const tc2li_orb* o, float* s, float* is, float* s2, float* is2) {
    if (!o) return TC2LI_ERR_INVALID;
    const size_t n = o->scale.size() * sizeof(float);
    if (s) memcpy(s, o->scale.data(), n);
    if (is) memcpy(is, o->inv_scale.data(), n);
    if (s2) memcpy(s2, o->sigma2.data(), n);
    if (is2) memcpy(is2, o->inv_sigma2.data(), n);
    return o->prm.nlevels;
}

int tc2li_orb_last_chunks(const tc2li_orb* o) { return o ? o->last_chunks : TC2LI_ERR_INVALID; }

int tc2li_orb_features_per_level(const tc2li_orb* o, int32_t* per_level) {
    if (!o || !per_level) return TC2LI_ERR_INVALID;
    memcpy(per_level, o->features_per_level.data(), o->features_per_level.size() * sizeof(int32_t));
    return o->prm.nlevels;
}

int tc2li_orb_level_size(const tc2li_orb* o, int level, int* w, int* h) {
    if (!o || level < 0 || level >= o->prm.nlevels) return TC2LI_ERR_INVALID;
    if (w) *w = o->geom[level].w;
    if (h) *h = o->geom[level].h;
    return TC2LI_OK;
}

int tc2li_orb_extract_batch(tc2li_orb* o, const uint8_t* dev_images, int n_images, int width, int height, int stride,
                            size_t image_pitch_bytes, const int32_t lapping_area[2], tc2li_keypoint* keypoints,
                            uint8_t* descriptors, int capacity, int32_t* n_keypoints, int32_t* mono_index,
                            void* stream_) {
    if (!o || !keypoints || !descriptors || !n_keypoints || !lapping_area || capacity < 0 || n_images < 0) {
        set_error("tc2li_orb_extract_batch: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_images == 0) return 0;
    if (width <= 0 || height <= 0 || !dev_images) {
        for (int i = 0; i < n_images; ++i) { n_keypoints[i] = 0; if (mono_index) mono_index[i] = -1; }
        return TC2LI_ERR_EMPTY;  // SF/src/ORBextractor.cc:1063-1064
    }
    if (n_images > o->max_images || width > o->max_w || height > o->max_h || stride < width) {
        set_error("tc2li_orb_extract_batch: %d images of %dx%d (stride %d) exceed the handle (%d of %dx%d)", n_images,
                  width, height, stride, o->max_images, o->max_w, o->max_h);
        return TC2LI_ERR_INVALID;
    }
    int rc = setup_geometry(o, width, height);
    if (rc != TC2LI_OK) return rc;
    hipStream_t st = (hipStream_t)stream_;
    const auto t_begin = std::chrono::steady_clock::now();
    const int L = o->prm.nlevels, M = n_images;
    const int ncells = (int)o->cells.size();

    LevelTable& raw = o->raw_tab;
    LevelTable& blur = o->blur_tab;
    for (int l = 0; l < L; ++l) {
        const LevelGeom& g = o->geom[l];
        raw.lv[l] = LevelDesc{l == 0 ? dev_images : o->d_levels[l].p, l == 0 ? image_pitch_bytes : g.img_stride,
                              l == 0 ? stride : g.pitch, g.w, g.h, 0};
        blur.lv[l] = LevelDesc{o->d_blur[l].p, g.img_stride, g.pitch, g.w, g.h, 0};
    }
    o->last_nimg = M;

    // The whole call is queued at once (pyramid, FAST, compaction, keypoint distribution, then orientation + descriptors once the blur
    // -- on the side stream -- is done) and the host waits once, at the end.  TC2LI_ORB_CHUNKS > 1 queues the images in that many
    // chunks instead; measured on MI355X that is slower now that no host stage sits between the kernels (128 KITTI images: 2.4 ms in one
    // chunk, 2.9 ms in two, 3.9 ms in four: every chunk pays the tail of its distribution kernel).  With profiling on every kernel is on
    // the caller's stream, which gives clean per-stage durations.
    const char* chunk_env = getenv("TC2LI_ORB_CHUNKS");  // read per call: the tests switch it
    const int kChunkEnv = chunk_env ? atoi(chunk_env) : 0;
    // lanes per distribution job: a batch fills the GPU with jobs and 256 lanes per job are best (0.56 ms per 128 images; 512: 0.76, 1024:
    // 1.21); a stereo pair alone is 16 jobs, each a long chain of rounds, and more lanes shorten a round (0.265 / 0.223 / 0.214 ms)
    static const int kQuadThreadsEnv = getenv("TC2LI_QUADTREE_THREADS") ? atoi(getenv("TC2LI_QUADTREE_THREADS")) : 0;
    const int kQuadThreads = kQuadThreadsEnv > 0 ? kQuadThreadsEnv : 0;  // 0: keys sorted by path in LDS (quadtree_kernels.hip); > 0: the global-memory form with that many lanes
    // Round 4: a large batch goes in TWO chunks on two streams -- the tail of a chunk (k_orient_describe and k_quadtree_gather write the host
    // mirrors of 92 MB per 1024 images over the bus; the distribution kernel's last workgroups) runs under the other chunk's FAST and blur
    // instead of holding the extraction stream alone.  (On one stream chunking only added tails: 2.4 / 2.9 / 3.9 ms for 1 / 2 / 4 chunks.)
    // TC2LI_ORB_CHUNK_STREAMS=0: the chunks one after the other on the caller's stream, as before (A/B measurements).
    const char* cs_env = getenv("TC2LI_ORB_CHUNK_STREAMS");
    const bool two_streams = !(cs_env && atoi(cs_env) == 0);
    const int want_chunks = kChunkEnv > 0 ? kChunkEnv : (M >= 256 && two_streams ? 2 : 1);
    const int n_chunks = o->profiling ? 1 : std::max(1, std::min(std::min(want_chunks, (int)tc2li_orb::kMaxChunks), M));
    o->last_chunks = n_chunks;
    hipStream_t blur_st = o->profiling ? st : o->side_stream;
    auto EV = [&](int chunk, int k) { return o->ev[chunk * tc2li_orb::kEvPerChunk + k]; };
    auto chunk_begin = [&](int c) { return (int)((long)M * c / n_chunks); };
    const int kp_stride = o->kp_cap_per_image;
    TC2LI_HIP_CHECK(hipMemsetAsync(o->d_status.p, 0, sizeof(int), st));
    if (ncells == 0) TC2LI_HIP_CHECK(hipMemsetAsync(o->d_level_counts.p, 0, (size_t)M * L * sizeof(int), st));
    if (n_chunks > 1 && two_streams && !o->profiling) TC2LI_HIP_CHECK(hipEventRecord(o->ev_fork, st));
    const int chunk_jobs_max = ((M + n_chunks - 1) / n_chunks + 1) * L;  // jobs of the largest chunk
    TC2LI_HIP_CHECK(o->d_qclass.ensure((size_t)n_chunks * quadtree_class_work_ints(chunk_jobs_max)));
    for (int c = 0; c < n_chunks; ++c) {
        const int i0 = chunk_begin(c), m = chunk_begin(c + 1) - i0;
        hipStream_t cst = ((c & 1) && two_streams && !o->profiling) ? o->chunk_stream : st;  // this chunk's stream
        if (cst != st) TC2LI_HIP_CHECK(hipStreamWaitEvent(cst, o->ev_fork, 0));  // behind whatever the caller's stream held when the call began
        LevelTable craw = raw, cblur = blur;  // this chunk's images
        for (int l = 0; l < L; ++l) {
            craw.lv[l].img = raw.lv[l].img + (size_t)i0 * raw.lv[l].img_stride;
            cblur.lv[l].img = blur.lv[l].img + (size_t)i0 * blur.lv[l].img_stride;
        }
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 0), cst));
        // the pyramid: the large levels a launch each over the whole GPU, the small ones (from level kTailFrom on) in ONE launch, a workgroup per
        // image -- seven dependent launches cost the chunk's stream ~0.6 ms each beside the other stages' kernels, whatever their size
        static const int kTailFromEnv = getenv("TC2LI_RESIZE_TAIL_FROM") ? atoi(getenv("TC2LI_RESIZE_TAIL_FROM")) : 3;
        const int tail_from = (m >= 32 && kTailFromEnv >= 1) ? std::min(kTailFromEnv, L) : L;
        for (int l = 1; l < tail_from; ++l)
            launch_resize(craw.lv[l - 1], craw.lv[l], o->d_xofs[l].p, o->d_ialpha[l].p, o->d_yofs[l].p, o->d_ibeta[l].p, m, cst);
        if (tail_from < L) {
            const int* xo[kMaxLevels]; const short* ia[kMaxLevels]; const int* yo[kMaxLevels]; const short* ib[kMaxLevels];
            for (int l = 0; l < L; ++l) { xo[l] = o->d_xofs[l].p; ia[l] = o->d_ialpha[l].p; yo[l] = o->d_yofs[l].p; ib[l] = o->d_ibeta[l].p; }
            launch_resize_tail(craw, xo, ia, yo, ib, tail_from, L, m, cst);
        }
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 1), cst));
        if (!o->profiling) TC2LI_HIP_CHECK(hipStreamWaitEvent(blur_st, EV(c, 1), 0));
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 4), blur_st));
        launch_blur_all(craw, cblur, L, m, o->gauss_rounded_taps, blur_st);
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 5), blur_st));
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 8), cst));
        if (ncells > 0) {
            launch_fast(craw, o->d_cells.p, ncells, o->prm.ini_th_fast, o->prm.min_th_fast, o->d_slab.p + (size_t)i0 * o->slab_per_image,
                        (size_t)o->slab_per_image, o->d_cell_counts.p + (size_t)i0 * ncells, m, o->d_cell_ids.p, o->n_small_cells,
                        o->d_cell_ids.p + o->n_small_cells, o->n_large_cells, cst);
            TC2LI_HIP_CHECK(hipEventRecord(EV(c, 3), cst));
            launch_compact(o->d_cells.p, o->d_level_cell_begin.p, o->d_cell_counts.p + (size_t)i0 * ncells, ncells,
                           o->d_slab.p + (size_t)i0 * o->slab_per_image, (size_t)o->slab_per_image,
                           o->d_dense.p + (size_t)i0 * o->slab_per_image, o->d_level_dense_off.p, o->d_level_counts.p + (size_t)i0 * L, L, m, cst);
        } else {
            TC2LI_HIP_CHECK(hipEventRecord(EV(c, 3), cst));
        }
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 2), cst));
        // ---- stage 2: keypoint distribution per (image, level), and the per-image keypoint lists ----
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 9), cst));
        launch_quadtree(o->d_jobs.p, i0 * L, m * L, o->d_dense.p, o->d_level_counts.p, o->d_qscratch.p, o->d_picked.p, o->d_picked_count.p, o->d_status.p,
                        kQuadThreads, L, cst, o->d_qclass.p + (size_t)c * quadtree_class_work_ints(chunk_jobs_max));
        // (the results reach the pinned mirrors through the kernels' own stores.  Device copies only + five hipMemcpyAsync per chunk behind the
        // kernels -- the copy engines instead of bus-bound wavefronts -- was measured in round 4: ORB stage alone 9.8 -> 12.3 ms per 1024 images,
        // the whole loop 29.1 -> 29.9 ms per step over three A/B pairs; the same with ONE k_copy_tasks launch of 16 / 64 workgroups behind the
        // chunk's kernels instead of the copy engines: 11.5 ms alone, 29.7-30.2 against 28.6-29.2 in the loop -- the bus time of the results
        // is hidden best inside the descriptor kernel itself)
        launch_quadtree_gather(o->d_jobs.p, o->d_picked.p, o->d_picked_count.p, o->d_level_counts.p, i0, m, L, kp_stride, o->d_kps.p, o->h_kps.p, o->d_nkp.p,
                               o->h_nkp.p, o->h_level_counts.p, o->d_status.p, cst);
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 10), cst));
        // ---- stage 3: orientation + descriptors of the chunk's keypoints ----
        TC2LI_HIP_CHECK(hipStreamWaitEvent(cst, EV(c, 5), 0));
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 6), cst));
        launch_orient_describe(raw, blur, o->scale_tab, o->d_kps.p, o->d_nkp.p, i0, m, kp_stride, o->h_angles.p, o->d_angles.p, o->h_desc.p, o->d_mkeys.p, o->d_desc.p, cst);
        TC2LI_HIP_CHECK(hipEventRecord(EV(c, 7), cst));
    }
    for (int c = 1; c < n_chunks; c += 2)  // the odd chunks' work joins the caller's stream
        if (two_streams && !o->profiling) TC2LI_HIP_CHECK(hipStreamWaitEvent(st, EV(c, 7), 0));
    TC2LI_HIP_CHECK(hipMemcpyAsync(o->h_status.p, o->d_status.p, sizeof(int), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipGetLastError());
    WorkerPool& pool = global_pool();
    const auto t_queued = std::chrono::steady_clock::now();
    TC2LI_HIP_CHECK(stream_wait_blocking(st));
    if (o->h_status.p[0] != 0) {
        static const char* const what[] = {"", "more candidates than the level's region holds", "more initial nodes than the node arrays hold",
                                           "the node list outgrew its arrays", "more keypoints than the level's pick region holds", "keypoint capacity exceeded"};
        set_error("keypoint distribution: %s", what[std::min(std::max(o->h_status.p[0], 0), 5)]);
        return TC2LI_ERR_CAPACITY;
    }
    o->last_level_counts.assign(o->h_level_counts.p, o->h_level_counts.p + (size_t)M * L);
    o->last_kp_off.resize(M); o->last_kp_cnt.resize(M);
    for (int i = 0; i < M; ++i) { o->last_kp_off[i] = i * kp_stride; o->last_kp_cnt[i] = o->h_nkp.p[i]; }
    o->last_plain_order = true;
    const std::vector<int>& img_kp_off = o->last_kp_off;
    const std::vector<int>& img_kp_cnt = o->last_kp_cnt;

    // ---- assemble in the reference's output order (SF/src/ORBextractor.cc:1093-1137) -----------------------
    std::atomic<int> status{TC2LI_OK};
    pool.parallel_for(M, [&](int i) {
        const int n = img_kp_cnt[i];
        n_keypoints[i] = n;
        if (n > capacity) { status = TC2LI_ERR_CAPACITY; if (mono_index) mono_index[i] = -1; return; }
        tc2li_keypoint* kout = keypoints + (size_t)i * capacity;
        uint8_t* dout = descriptors + (size_t)i * capacity * 32;
        int mono = 0, stereo = n - 1;
        for (int k = 0; k < n; ++k) {
            const int g = img_kp_off[i] + k;
            const DevKeypoint& dk = o->h_kps.p[g];
            const int l = dk.img_level & 0xff;
            tc2li_keypoint kp;
            kp.x = (float)((dk.packed >> 8) & 0xfff);
            kp.y = (float)(dk.packed >> 20);
            kp.response = (float)(dk.packed & 0xff);
            kp.octave = l;
            kp.size = (float)(int)(31 * o->scale[l]);  // scaledPatchSize, :851
            kp.angle = o->h_angles.p[g];
            if (l != 0) { kp.x *= o->scale[l]; kp.y *= o->scale[l]; }
            int dsti;
            if (kp.x >= lapping_area[0] && kp.x <= lapping_area[1]) { dsti = stereo--; o->last_plain_order = false; }
            else dsti = mono++;
            kout[dsti] = kp;
            memcpy(dout + (size_t)dsti * 32, o->h_desc.p + (size_t)g * 32, 32);
        }
        if (mono_index) mono_index[i] = mono;
    });
    for (int k = 0; k < 5; ++k) o->timings[k] = 0;
    for (int c = 0; c < n_chunks; ++c) {  // device stages: the chunks' durations added up
        o->timings[0] += elapsed(EV(c, 0), EV(c, 1));
        o->timings[1] += elapsed(EV(c, 8), EV(c, 3));
        o->timings[2] += elapsed(EV(c, 3), EV(c, 2));
        o->timings[3] += elapsed(EV(c, 4), EV(c, 5));
        o->timings[4] += elapsed(EV(c, 6), EV(c, 7));
    }
    o->timings[5] = 0;
    for (int c = 0; c < n_chunks; ++c) o->timings[5] += elapsed(EV(c, 9), EV(c, 10));
    o->timings[6] = std::chrono::duration<float, std::milli>(t_queued - t_begin).count();
    o->timings[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (status.load() != TC2LI_OK) { set_error("keypoint capacity %d too small", capacity); return status.load(); }
    return n_images;
}

int tc2li_orb_extract(tc2li_orb* o, const uint8_t* image, int width, int height, int stride,
                      const int32_t lapping_area[2], tc2li_keypoint* keypoints, uint8_t* descriptors, int capacity,
                      int32_t* n_keypoints) {
    if (!o || !n_keypoints) { set_error("tc2li_orb_extract: invalid argument"); return TC2LI_ERR_INVALID; }
    if (!image || width <= 0 || height <= 0) { *n_keypoints = 0; return TC2LI_ERR_EMPTY; }
    if (width > o->max_w || height > o->max_h || stride < width) {
        set_error("tc2li_orb_extract: image %dx%d exceeds the handle", width, height);
        return TC2LI_ERR_INVALID;
    }
    int rc = setup_geometry(o, width, height);
    if (rc != TC2LI_OK) return rc;
    const LevelGeom& g = o->geom[0];
    hipStream_t ps = private_stream();  // not the NULL stream: it would serialise against every blocking stream of the process
    TC2LI_HIP_CHECK(hipMemcpy2DAsync(o->d_level0.p, g.pitch, image, stride, width, height, hipMemcpyHostToDevice, ps));
    TC2LI_HIP_CHECK(hipStreamSynchronize(ps));
    int32_t mono = -1;
    rc = tc2li_orb_extract_batch(o, o->d_level0.p, 1, width, height, g.pitch, g.img_stride, lapping_area, keypoints,
                                 descriptors, capacity, n_keypoints, &mono, ps);
    if (rc < 0) return rc;
    return mono;
}

static int download_plane(const LevelDesc& d, int image_index, uint8_t* dst) {
    TC2LI_HIP_CHECK(hipMemcpy2D(dst, d.w, d.img + (size_t)image_index * d.img_stride, d.pitch, d.w, d.h,
                                hipMemcpyDeviceToHost));
    return TC2LI_OK;
}

int tc2li_orb_download_level(tc2li_orb* o, int image_index, int level, uint8_t* dst) {
    if (!o || !dst || level < 0 || level >= o->prm.nlevels || image_index < 0 || image_index >= o->last_nimg)
        return TC2LI_ERR_INVALID;
    return download_plane(o->raw_tab.lv[level], image_index, dst);
}

int tc2li_orb_download_blurred(tc2li_orb* o, int image_index, int level, uint8_t* dst) {
    if (!o || !dst || level < 0 || level >= o->prm.nlevels || image_index < 0 || image_index >= o->last_nimg)
        return TC2LI_ERR_INVALID;
    return download_plane(o->blur_tab.lv[level], image_index, dst);
}

int tc2li_orb_download_candidates(tc2li_orb* o, int image_index, int level, float* xyr, int capacity) {
    if (!o || level < 0 || level >= o->prm.nlevels || image_index < 0 || image_index >= o->last_nimg)
        return TC2LI_ERR_INVALID;
    const int L = o->prm.nlevels;
    const int n = o->last_level_counts[image_index * L + level];
    if (!xyr) return n;
    if (n > capacity) return TC2LI_ERR_CAPACITY;
    std::vector<uint32_t> c((size_t)std::max(n, 1));
    TC2LI_HIP_CHECK(copy_sync(c.data(), o->d_dense.p + (size_t)image_index * o->slab_per_image + o->geom[level].dense_off, (size_t)n * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, private_stream()));
    for (int k = 0; k < n; ++k) {
        xyr[3 * k] = (float)(((c[k] >> 8) & 0xfff) + kMinBorder);
        xyr[3 * k + 1] = (float)((c[k] >> 20) + kMinBorder);
        xyr[3 * k + 2] = (float)(c[k] & 0xff);
    }
    return n;
}

int tc2li_host_distribute_quadtree(const float* xyr, int n, int min_x, int max_x, int min_y, int max_y, int n_target,
                                   float* out_xyr, int capacity) {
    if (!xyr || !out_xyr || n < 0) return TC2LI_ERR_INVALID;
    std::vector<uint32_t> cand(n);
    for (int i = 0; i < n; ++i) {
        const int x = (int)xyr[3 * i], y = (int)xyr[3 * i + 1], r = (int)xyr[3 * i + 2];
        if (x < 0 || x > 4095 || y < 0 || y > 4095 || r < 0 || r > 255) return TC2LI_ERR_INVALID;
        cand[i] = ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)r;
    }
    QuadtreeScratch scratch;
    std::vector<int32_t> picked;
    distribute_quadtree(cand.data(), n, min_x, max_x, min_y, max_y, n_target, scratch, picked);
    if ((int)picked.size() > capacity) return TC2LI_ERR_CAPACITY;
    for (size_t k = 0; k < picked.size(); ++k) memcpy(out_xyr + 3 * k, xyr + 3 * picked[k], 3 * sizeof(float));
    return (int)picked.size();
}

int tc2li_device_distribute_quadtree(const float* xyr, int n, int min_x, int max_x, int min_y, int max_y, int n_target, float* out_xyr, int capacity,
                                     int threads) {
    if (!xyr || !out_xyr || n < 0 || n >= (1 << 20) || n_target < 0 || capacity < 0) return TC2LI_ERR_INVALID;
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    std::vector<uint32_t> cand((size_t)std::max(n, 1));
    for (int i = 0; i < n; ++i) {
        const int x = (int)xyr[3 * i], y = (int)xyr[3 * i + 1], r = (int)xyr[3 * i + 2];
        if (x < 0 || x > 4095 || y < 0 || y > 4095 || r < 0 || r > 255) return TC2LI_ERR_INVALID;
        cand[i] = ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)r;
    }
    const int bw = max_x - min_x, bh = max_y - min_y;
    const int n_ini = bh > 0 && bw > 0 ? (int)std::round(static_cast<float>(bw) / bh) : 0;
    QuadJob job{};
    job.max_keys = std::max(n, 1);
    job.max_nodes = std::max(n_target + 3, 4 * std::max(n_ini, 1)) + 8;
    job.out_cap = job.max_nodes;
    job.min_x = min_x; job.max_x = max_x; job.min_y = min_y; job.max_y = max_y; job.n_target = n_target;
    DevBuf<uint32_t> d_cand, d_picked;
    DevBuf<int> d_counts;  // [0] candidates, [1] picks... the job's count_idx is 0 for both arrays
    DevBuf<int> d_pick_count, d_status;
    DevBuf<QuadJob> d_job;
    DevBuf<uint8_t> d_scratch;
    hipStream_t ps = private_stream();
    TC2LI_HIP_CHECK(d_cand.upload(cand));
    TC2LI_HIP_CHECK(d_counts.upload(std::vector<int>{n}));
    TC2LI_HIP_CHECK(d_pick_count.upload(std::vector<int>{0}));
    TC2LI_HIP_CHECK(d_status.upload(std::vector<int>{0}));
    TC2LI_HIP_CHECK(d_job.upload(std::vector<QuadJob>{job}));
    TC2LI_HIP_CHECK(d_scratch.alloc(quadtree_scratch_bytes(job.max_keys, job.max_nodes)));
    TC2LI_HIP_CHECK(d_picked.alloc((size_t)job.out_cap));
    launch_quadtree(d_job.p, 0, 1, d_cand.p, d_counts.p, d_scratch.p, d_picked.p, d_pick_count.p, d_status.p, threads, 1, ps);
    TC2LI_HIP_CHECK(hipGetLastError());
    int count = 0, status = 0;
    TC2LI_HIP_CHECK(copy_sync(&count, d_pick_count.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    TC2LI_HIP_CHECK(copy_sync(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost, ps));
    if (status != 0) { set_error("keypoint distribution kernel: status %d", status); return TC2LI_ERR_CAPACITY; }
    if (count > capacity) return TC2LI_ERR_CAPACITY;
    std::vector<uint32_t> picked((size_t)std::max(count, 1));
    TC2LI_HIP_CHECK(copy_sync(picked.data(), d_picked.p, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, ps));
    for (int k = 0; k < count; ++k) {
        out_xyr[3 * k] = (float)((picked[k] >> 8) & 0xfff);
        out_xyr[3 * k + 1] = (float)(picked[k] >> 20);
        out_xyr[3 * k + 2] = (float)(picked[k] & 0xff);
    }
    return count;
}

int tc2li_orb_set_profiling(tc2li_orb* o, int enabled) {
    if (!o) return TC2LI_ERR_INVALID;
    o->profiling = enabled != 0;
    return TC2LI_OK;
}

int tc2li_orb_last_timings(const tc2li_orb* o, float ms[8]) {
    if (!o || !ms) return TC2LI_ERR_INVALID;
    memcpy(ms, o->timings, sizeof(o->timings));
    return TC2LI_OK;
}

}  // extern "C"
