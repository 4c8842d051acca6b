// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  See localmap.hpp.
#include "localmap.hpp"

#include <cstddef>
#include <map>

namespace oracle {

LocalMap UpdateLocalMap(const MapGraph& g, const int32_t* frame_points, int n_frame_points, int temporal_last_kf) {
    LocalMap out;
    out.frame_cleared.assign(n_frame_points, 0);
    // ---- UpdateLocalKeyFrames (Tracking.cc:3326-3476) ----
    std::map<int, int> keyframeCounter;  // :3329 (keyed by address there, by index here)
    for (int i = 0; i < n_frame_points; ++i) {
        const int p = frame_points[i];
        if (p < 0) continue;
        if (!g.point_bad[p]) {
            for (int k = g.obs_off[p]; k < g.obs_off[p + 1]; ++k) keyframeCounter[g.obs_kf[k]]++;  // :3339-3341
        } else {
            out.frame_cleared[i] = 1;  // :3345
        }
    }
    std::vector<uint8_t> marked(g.n_keyframes, 0);  // mnTrackReferenceForFrame == mCurrentFrame.mnId
    int max = 0, kf_max = -1;
    for (const auto& it : keyframeCounter) {  // :3382-3397
        const int kf = it.first;
        if (g.kf_bad[kf]) continue;
        if (it.second > max) { max = it.second; kf_max = kf; }
        out.keyframes.push_back(kf);
        marked[kf] = 1;
    }
    // :3400-3451 -- the loop's end iterator is taken before anything is appended: only the voted keyframes are visited
    const size_t n_voted = out.keyframes.size();
    for (size_t j = 0; j < n_voted; ++j) {
        if (out.keyframes.size() > 80) break;  // :3404
        const int kf = out.keyframes[j];
        int taken = 0;
        for (int k = g.covis_off[kf]; k < g.covis_off[kf + 1] && taken < 10; ++k, ++taken) {  // GetBestCovisibilityKeyFrames(10)
            const int n = g.covis[k];
            if (!g.kf_bad[n] && !marked[n]) { out.keyframes.push_back(n); marked[n] = 1; break; }
        }
        for (int k = g.child_off[kf]; k < g.child_off[kf + 1]; ++k) {  // :3426-3439
            const int c = g.children[k];
            if (!g.kf_bad[c] && !marked[c]) { out.keyframes.push_back(c); marked[c] = 1; break; }
        }
        const int par = g.parent[kf];  // :3441-3450: no isBad() test, and the `break` leaves the keyframe loop
        if (par >= 0 && !marked[par]) { out.keyframes.push_back(par); marked[par] = 1; break; }
    }
    if (temporal_last_kf >= 0 && out.keyframes.size() < 80) {  // :3454-3469
        int t = temporal_last_kf;
        for (int i = 0; i < 20; ++i) {
            if (t < 0) break;
            if (!marked[t]) { out.keyframes.push_back(t); marked[t] = 1; t = g.prev_kf[t]; }  // advances only after a push
        }
    }
    out.reference_kf = kf_max;  // :3471-3475
    // ---- UpdateLocalPoints (Tracking.cc:3296-3323) ----
    std::vector<uint8_t> seen(g.n_points, 0);  // mnTrackReferenceForFrame of the points
    for (size_t j = out.keyframes.size(); j-- > 0;) {  // reverse iteration (:3302)
        const int kf = out.keyframes[j];
        for (int k = g.match_off[kf]; k < g.match_off[kf + 1]; ++k) {
            const int p = g.matches[k];
            if (p < 0 || seen[p]) continue;
            if (!g.point_bad[p]) { out.points.push_back(p); seen[p] = 1; }
        }
    }
    return out;
}

}  // namespace oracle
