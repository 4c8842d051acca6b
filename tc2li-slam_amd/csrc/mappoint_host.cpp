// tc2li_map_points_refresh (include/tc2li_hip.h): uploads, one launch, downloads.
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "mappoint_device.hpp"

using namespace tc2li;

namespace {
struct Workspace {
    DevBuf<int32_t> d_off, d_best;
    DevBuf<uint8_t> d_desc;
    DevBuf<float> d_centres, d_pos, d_ref, d_scale, d_normals, d_min, d_max;
    std::mutex mu;
};
Workspace& ws() { return shutdown_owned<Workspace>(); }
}  // namespace

extern "C" int tc2li_map_points_refresh(int n_points, const int32_t* obs_offsets, const uint8_t* obs_descriptors, const float* obs_centres,
                                        const float* positions, const float* ref_centres, const float* ref_level_scale, float last_level_scale,
                                        int32_t* best_obs, float* normals, float* min_distance, float* max_distance, void* stream_) {
    if (n_points < 0 || (n_points > 0 && (!obs_offsets || !positions || !ref_centres || !ref_level_scale || !best_obs || !normals || !min_distance ||
                                          !max_distance)) || !(last_level_scale > 0)) {
        set_error("tc2li_map_points_refresh: invalid argument");
        return TC2LI_ERR_INVALID;
    }
    if (n_points == 0) return 0;
    if (obs_offsets[0] != 0) { set_error("tc2li_map_points_refresh: obs_offsets[0] must be 0"); return TC2LI_ERR_INVALID; }
    for (int p = 0; p < n_points; ++p) {
        const int n = obs_offsets[p + 1] - obs_offsets[p];
        if (n < 0) { set_error("tc2li_map_points_refresh: obs_offsets must not decrease"); return TC2LI_ERR_INVALID; }
        if (n > kMaxObservations) { set_error("point %d has %d observations, at most %d are supported", p, n, kMaxObservations); return TC2LI_ERR_CAPACITY; }
    }
    const int total = obs_offsets[n_points];
    if (total > 0 && (!obs_descriptors || !obs_centres)) { set_error("tc2li_map_points_refresh: null observation arrays"); return TC2LI_ERR_INVALID; }
    if (!device_ready()) return TC2LI_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream_;
    Workspace& w = ws();
    std::lock_guard<std::mutex> lk(w.mu);
    const size_t P = n_points, T = std::max(total, 1);
    TC2LI_HIP_CHECK(w.d_off.ensure(P + 1)); TC2LI_HIP_CHECK(w.d_best.ensure(P)); TC2LI_HIP_CHECK(w.d_desc.ensure(32 * T));
    TC2LI_HIP_CHECK(w.d_centres.ensure(3 * T)); TC2LI_HIP_CHECK(w.d_pos.ensure(3 * P)); TC2LI_HIP_CHECK(w.d_ref.ensure(3 * P));
    TC2LI_HIP_CHECK(w.d_scale.ensure(P)); TC2LI_HIP_CHECK(w.d_normals.ensure(3 * P)); TC2LI_HIP_CHECK(w.d_min.ensure(P)); TC2LI_HIP_CHECK(w.d_max.ensure(P));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_off.p, obs_offsets, (P + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (total > 0) {
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_desc.p, obs_descriptors, 32 * (size_t)total, hipMemcpyHostToDevice, st));
        TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_centres.p, obs_centres, 3 * (size_t)total * sizeof(float), hipMemcpyHostToDevice, st));
    }
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_pos.p, positions, 3 * P * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_ref.p, ref_centres, 3 * P * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_scale.p, ref_level_scale, P * sizeof(float), hipMemcpyHostToDevice, st));
    // points without observations keep what the caller passed in (the reference returns early for them)
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_normals.p, normals, 3 * P * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_min.p, min_distance, P * sizeof(float), hipMemcpyHostToDevice, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(w.d_max.p, max_distance, P * sizeof(float), hipMemcpyHostToDevice, st));
    MapPointRefresh a{w.d_off.p, w.d_desc.p, w.d_centres.p, w.d_pos.p, w.d_ref.p, w.d_scale.p, last_level_scale, 0, w.d_best.p, w.d_normals.p, w.d_min.p, w.d_max.p};
    launch_map_points_refresh(a, n_points, st);
    TC2LI_HIP_CHECK(hipGetLastError());
    TC2LI_HIP_CHECK(hipMemcpyAsync(best_obs, w.d_best.p, P * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(normals, w.d_normals.p, 3 * P * sizeof(float), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(min_distance, w.d_min.p, P * sizeof(float), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipMemcpyAsync(max_distance, w.d_max.p, P * sizeof(float), hipMemcpyDeviceToHost, st));
    TC2LI_HIP_CHECK(hipStreamSynchronize(st));
    return n_points;
}
