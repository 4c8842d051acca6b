"""tc2li-slam_amd -- host-side mirror (ctypes) of the gfx950 C-ABI library ``libtc2li_hip.so``.

The directory name carries a hyphen (it is fixed by the project layout), so import it through
``tc2li_loader.load()`` at the repository root, which registers it as module ``tc2li_slam_amd``.

Only plumbing lives here: every computation happens inside the shared library (hand-written HIP kernels plus
the C++ host stages); there is no Python or CPU fallback -- without the built library, or without a GPU, the
compute entry points raise.
"""
from .capi import (  # noqa: F401
    LIB_PATH,
    Tc2liError,
    OrbParams,
    OrbExtractor,
    abi_version,
    search_by_projection,
    project_last_frame,
    project_local_map,
    QUERY_DTYPE,
    MAP_POINT_DTYPE,
    local_bundle_adjustment,
    BA_EDGE_DTYPE,
    pack_ba_edges,
    pose_optimization,
    pose_optimization_batch,
    LidarFrontEnd,
    LidarMap,
    pack_lidar_state,
    POINT_DTYPE,
    VELODYNE_DTYPE,
    compute_stereo_matches,
    stereo_match_batch,
    device_count,
    distribute_quadtree_host,
    distribute_quadtree_device,
    exported_symbols,
    lib,
)
