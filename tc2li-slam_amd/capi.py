"""ctypes binding of include/tc2li_hip.h and thin classes that mirror the reference's C++ interface
(``ORBextractor`` -> :class:`OrbExtractor`).  Mirrors names, argument meaning and error behaviour of
SF/include/ORBextractor.h:46-121 so that the parity tests read like tests of the reference class."""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtc2li_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tc2li_hip.h")

_lib = None


class Tc2liError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("tc2li error %d: %s" % (code, text))
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4")])


def lib():
    """Loads libtc2li_hip.so (built in-tree by ``__graft_entry__.build()``); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch first makes the
    # dynamic loader satisfy our DT_NEEDED libamdhip64.so.7 with that already-loaded copy, so device pointers and
    # streams can be shared with torch.  Without torch the system runtime under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    if not os.path.exists(LIB_PATH):
        raise Tc2liError(-3, "native library %s not built (run __graft_entry__.build()); there is no fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.tc2li_last_error.restype = C.c_char_p
    L.tc2li_orb_create.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tc2li_orb_destroy.argtypes = [C.c_void_p]
    L.tc2li_orb_destroy.restype = None
    L.tc2li_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_void_p,
                                    C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
    L.tc2li_orb_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                          C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
    L.tc2li_orb_levels.argtypes = [C.c_void_p]
    L.tc2li_orb_scale_factors.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    L.tc2li_orb_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
    L.tc2li_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tc2li_orb_download_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.tc2li_orb_download_blurred.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.tc2li_orb_download_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.tc2li_orb_last_timings.argtypes = [C.c_void_p, C.c_void_p]
    L.tc2li_orb_last_chunks.argtypes = [C.c_void_p]
    L.tc2li_orb_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.tc2li_host_distribute_quadtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_int]
    L.tc2li_device_distribute_quadtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.tc2li_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_stereo_match_batch.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p]
    L.tc2li_lidar_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tc2li_lidar_destroy.argtypes = [C.c_void_p]
    L.tc2li_lidar_destroy.restype = None
    L.tc2li_lidar_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_float, C.c_void_p, C.c_int]
    L.tc2li_lidar_voxel_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
    L.tc2li_lidar_map_create.argtypes = [C.POINTER(C.c_void_p)]
    L.tc2li_lidar_map_destroy.argtypes = [C.c_void_p]
    L.tc2li_lidar_map_destroy.restype = None
    L.tc2li_lidar_map_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tc2li_lidar_map_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tc2li_lidar_map_size.argtypes = [C.c_void_p]
    L.tc2li_lidar_feature_extraction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 8 + [C.c_int]
    L.tc2li_lidar_frontend_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p]
    L.tc2li_pose_optimization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.tc2li_pose_optimization_batch.argtypes = [C.c_int] + [C.c_void_p] * 8
    L.tc2li_local_bundle_adjustment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_local_lv_bundle_adjustment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                   C.c_int, C.c_double] + [C.c_void_p] * 7
    L.tc2li_local_lv_bundle_adjustment_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                           C.c_int, C.c_double] + [C.c_void_p] * 8
    L.tc2li_ba_shard_select.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.tc2li_rccl_unique_id.argtypes = [C.c_void_p]
    L.tc2li_rccl_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tc2li_rccl_comm_destroy.argtypes = [C.c_void_p]
    L.tc2li_rccl_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.tc2li_local_map_create.argtypes = [C.POINTER(C.c_void_p)]
    L.tc2li_local_map_destroy.argtypes = [C.c_void_p]
    L.tc2li_local_map_destroy.restype = None
    L.tc2li_local_map_set_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_local_map_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_local_map_device_points.argtypes = [C.c_void_p]
    L.tc2li_local_map_device_points.restype = C.c_void_p
    L.tc2li_lidar_window_evaluate.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.tc2li_track_motion_model_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_float, C.c_float] + [C.c_void_p] * 5
    L.tc2li_local_bundle_adjustment_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_local_bundle_adjustment_batch_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_ba_engine_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.tc2li_ba_engine_destroy.argtypes = [C.c_void_p]
    L.tc2li_ba_engine_destroy.restype = None
    L.tc2li_ba_engine_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_ba_engine_submit.restype = C.c_int64
    L.tc2li_ba_engine_wait.argtypes = [C.c_void_p, C.c_int64]
    L.tc2li_ba_engine_poll.argtypes = [C.c_void_p, C.c_int64]
    L.tc2li_lidar_last_timings.argtypes = [C.c_void_p, C.c_void_p]
    L.tc2li_imu_preintegrated_init.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.tc2li_imu_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    L.tc2li_imu_preintegrate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    L.tc2li_imu_delta.argtypes = [C.c_void_p] * 5
    L.tc2li_imu_predict_state.argtypes = [C.c_void_p] * 8
    L.tc2li_lidar_undistort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_lidar_imu_propagate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int]
    L.tc2li_lidar_map_incremental.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_lidar_map_delete_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_lidar_map_delete_boxes_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tc2li_lidar_map_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.tc2li_lidar_fov_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    L.tc2li_lidar_fov_segment_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.tc2li_local_inertial_bundle_adjustment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                         C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double] + [C.c_void_p] * 5
    L.tc2li_host_lidar_planes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.tc2li_device_lidar_planes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.tc2li_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    L.tc2li_project_last_frame.argtypes = [C.c_void_p] * 3 + [C.c_float, C.c_float, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_void_p]
    L.tc2li_project_local_map.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.tc2li_device_time_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise Tc2liError(rc, lib().tc2li_last_error().decode("utf-8", "replace"))
    return rc


def abi_version():
    return lib().tc2li_abi_version()


def device_count():
    return lib().tc2li_device_count()


def set_hardware_queues(n):
    """tc2li_set_hardware_queues: only effective before the process's first HIP call."""
    _check(lib().tc2li_set_hardware_queues(int(n)))


def lib_loaded():
    return _lib is not None


def shutdown():
    """tc2li_shutdown: joins the library's worker pools and releases its process-wide work spaces; no other thread may be inside the
    library.  Handles stay valid, the next call that needs a pool makes it again."""
    _check(lib().tc2li_shutdown())


def set_host_thread_budget(threads):
    """tc2li_set_host_thread_budget: host threads this process may keep busy (cores / ranks per node); before the pools exist."""
    _check(lib().tc2li_set_host_thread_budget(int(threads)))


def host_threads():
    """{budget, extractor_pool, tracking_pool, lidar_pool, ba_group_pool, ba_groups_max}: the sizes the pools have (or will get)."""
    out = (C.c_int32 * 6)()
    _check(lib().tc2li_host_threads(out, 6))
    return dict(zip(("budget", "extractor_pool", "tracking_pool", "lidar_pool", "ba_group_pool", "ba_groups_max"), [int(v) for v in out]))


def exported_symbols():
    """Names of the functions include/tc2li_hip.h declares (used by the CPU-side ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tc2li_[a-z0-9_]+)\s*\(", text)))


def diag_clocks():
    """(GHz inside the f64 MFMA loop, GHz inside the f64 FMA loop) of tc2li_diag_peaks' kernels."""
    a, b = C.c_double(0), C.c_double(0)
    _check(lib().tc2li_diag_clocks(C.byref(a), C.byref(b)))
    return a.value, b.value


def profile_enable(on=True):
    _check(lib().tc2li_profile_enable(int(bool(on))))


def ba_options():
    """tc2li_ba_options -> dict of the bundle-adjustment switches in effect."""
    import json
    f = lib().tc2li_ba_options
    f.argtypes = [C.c_char_p, C.c_int]
    buf = C.create_string_buffer(512)
    f(buf, len(buf))
    return json.loads(buf.value.decode())


def profile_report():
    """{kernel name: (launches, total ms)} of the launches since the last report (call with idle streams)."""
    f = lib().tc2li_profile_report
    f.argtypes = [C.c_char_p, C.c_int]
    buf = C.create_string_buffer(1 << 18)
    f(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split("\t")
        name = name.strip("() ").replace(" ", "")
        c0, m0 = out.get(name, (0, 0.0))
        out[name] = (c0 + int(calls), m0 + float(ms))
    return out


def diag_peaks():
    """(f64 MFMA TFLOP/s, f64 vector FMA TFLOP/s, HBM copy GB/s) measured on the current device."""
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    f = lib().tc2li_diag_peaks
    f.argtypes = [C.c_void_p] * 3
    _check(f(C.addressof(a), C.addressof(b), C.addressof(c)))
    return a.value, b.value, c.value


def distribute_quadtree_host(xyr, min_x, max_x, min_y, max_y, n_target):
    xyr = np.ascontiguousarray(xyr, dtype=np.float32).reshape(-1, 3)
    out = np.empty((max(len(xyr), 1), 3), np.float32)
    n = _check(lib().tc2li_host_distribute_quadtree(xyr.ctypes.data, len(xyr), min_x, max_x, min_y, max_y, n_target,
                                                    out.ctypes.data, len(out)))
    return out[:n].copy()


def distribute_quadtree_device(xyr, min_x, max_x, min_y, max_y, n_target, threads=0):
    """One job of the extractor's keypoint-distribution kernel (k_quadtree) on the given candidates."""
    xyr = np.ascontiguousarray(xyr, dtype=np.float32).reshape(-1, 3)
    out = np.empty((max(len(xyr), n_target + 16, 1), 3), np.float32)
    n = _check(lib().tc2li_device_distribute_quadtree(xyr.ctypes.data, len(xyr), min_x, max_x, min_y, max_y, n_target, out.ctypes.data, len(out),
                                                      threads))
    return out[:n].copy()


class OrbExtractor:
    """Mirror of ``TC2LI_SLAM::ORBextractor`` (SF/include/ORBextractor.h:46).

    ``extract(image)`` is ``operator()``: returns ``(monoIndex, keypoints, descriptors)``; an empty image gives
    ``(-1, [], [])`` like the reference (SF/src/ORBextractor.cc:1063-1064).
    """

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7,
                 max_width=1242, max_height=376, max_images=2):
        self._h = C.c_void_p()
        self.params = OrbParams(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast)
        self.capacity = nfeatures + 4 * nlevels
        self.max_images = max_images
        _check(lib().tc2li_orb_create(C.byref(self.params), max_width, max_height, max_images, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().tc2li_orb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # ---- accessors (ORBextractor.h:70-90) ----
    def GetLevels(self):
        return lib().tc2li_orb_levels(self._h)

    def _scales(self):
        n = self.GetLevels()
        arrs = [np.empty(n, np.float32) for _ in range(4)]
        _check(lib().tc2li_orb_scale_factors(self._h, *[a.ctypes.data for a in arrs]))
        return arrs

    def GetScaleFactors(self):
        return self._scales()[0]

    def GetInverseScaleFactors(self):
        return self._scales()[1]

    def GetScaleSigmaSquares(self):
        return self._scales()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._scales()[3]

    def features_per_level(self):
        a = np.empty(self.GetLevels(), np.int32)
        _check(lib().tc2li_orb_features_per_level(self._h, a.ctypes.data))
        return a

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(lib().tc2li_orb_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    # ---- operator() ----
    def extract(self, image, lapping_area=(0, 0)):
        image = np.asarray(image)
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int32(0)
        lap = (C.c_int32 * 2)(*lapping_area)
        if image.size == 0:
            rc = lib().tc2li_orb_extract(self._h, None, 0, 0, 0, lap, kps.ctypes.data, desc.ctypes.data, self.capacity,
                                         C.byref(n))
            return rc, kps[:0], desc[:0]
        assert image.dtype == np.uint8 and image.ndim == 2
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        rc = lib().tc2li_orb_extract(self._h, image.ctypes.data, image.shape[1], image.shape[0], image.strides[0], lap,
                                     kps.ctypes.data, desc.ctypes.data, self.capacity, C.byref(n))
        _check(rc)
        return rc, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_dev(self, dev_ptr, n_images, width, height, stride, image_pitch, stream=0, lapping_area=(0, 0),
                          out=None):
        """Images resident in device memory (``dev_ptr`` = integer device address). Returns per-image arrays."""
        if out is None:
            out = (np.zeros((n_images, self.capacity), KEYPOINT_DTYPE), np.zeros((n_images, self.capacity, 32), np.uint8),
                   np.zeros(n_images, np.int32), np.zeros(n_images, np.int32))
        kps, desc, counts, mono = out
        lap = (C.c_int32 * 2)(*lapping_area)
        _check(lib().tc2li_orb_extract_batch(self._h, C.c_void_p(dev_ptr), n_images, width, height, stride, image_pitch,
                                             lap, kps.ctypes.data, desc.ctypes.data, self.capacity, counts.ctypes.data,
                                             mono.ctypes.data, C.c_void_p(stream)))
        return kps, desc, counts, mono

    # ---- mvImagePyramid and diagnostics ----
    def pyramid_level(self, image_index, level):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        _check(lib().tc2li_orb_download_level(self._h, image_index, level, out.ctypes.data))
        return out

    def blurred_level(self, image_index, level):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        _check(lib().tc2li_orb_download_blurred(self._h, image_index, level, out.ctypes.data))
        return out

    def candidates(self, image_index, level):
        n = _check(lib().tc2li_orb_download_candidates(self._h, image_index, level, None, 0))
        out = np.empty((max(n, 1), 3), np.float32)
        n = _check(lib().tc2li_orb_download_candidates(self._h, image_index, level, out.ctypes.data, len(out)))
        return out[:n]

    def set_profiling(self, enabled):
        _check(lib().tc2li_orb_set_profiling(self._h, int(enabled)))

    def last_timings(self):
        t = np.zeros(8, np.float32)
        _check(lib().tc2li_orb_last_timings(self._h, t.ctypes.data))
        return t

    def last_chunks(self):
        """Chunks of images the last batch call was pipelined over (every device stage is launched once per chunk)."""
        return _check(lib().tc2li_orb_last_chunks(self._h))


def compute_stereo_matches(ext_left, ext_right, kps_l, desc_l, kps_r, desc_r, bf, b):
    """``Frame::ComputeStereoMatches`` (SF/src/Frame.cc:841): returns (mvuRight, mvDepth, bestSAD)."""
    kps_l = np.ascontiguousarray(kps_l, KEYPOINT_DTYPE)
    kps_r = np.ascontiguousarray(kps_r, KEYPOINT_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8)
    desc_r = np.ascontiguousarray(desc_r, np.uint8)
    n = len(kps_l)
    u = np.full(max(n, 1), -1, np.float32)
    d = np.full(max(n, 1), -1, np.float32)
    s = np.full(max(n, 1), -1, np.int32)
    _check(lib().tc2li_stereo_match(ext_left._h, ext_right._h, kps_l.ctypes.data, desc_l.ctypes.data, n, kps_r.ctypes.data,
                                    desc_r.ctypes.data, len(kps_r), bf, b, u.ctypes.data, d.ctypes.data, s.ctypes.data))
    return u[:n], d[:n], s[:n]


def stereo_match_batch(ext, n_frames, bf, b, stream=0, out=None):
    """Batched stereo matching on the device-resident features of the preceding ``extract_batch_dev`` call."""
    if out is None:
        out = (np.full((n_frames, ext.capacity), -1, np.float32), np.full((n_frames, ext.capacity), -1, np.float32),
               np.full((n_frames, ext.capacity), -1, np.int32))
    u, d, s = out
    _check(lib().tc2li_stereo_match_batch(ext._h, n_frames, bf, b, u.ctypes.data, d.ctypes.data, s.ctypes.data, ext.capacity,
                                          C.c_void_p(stream)))
    return u, d, s


# ---- LiDAR front end ---------------------------------------------------------------------------------------------
VELODYNE_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"),
                           ("time", "<f4"), ("ring", "<u2"), ("pad1", "<u2"), ("pad2", "<f4")])
POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("normal_x", "<f4"),
                        ("normal_y", "<f4"), ("normal_z", "<f4"), ("pad1", "<f4"), ("intensity", "<f4"),
                        ("curvature", "<f4"), ("pad2", "<f4"), ("pad3", "<f4")])


def pack_lidar_state(rot, pos, offset_r=None, offset_t=None):
    """tc2li_lidar_state as 24 float64 (rot row-major, pos, offset_R_L_I row-major, offset_T_L_I)."""
    offset_r = np.eye(3) if offset_r is None else offset_r
    offset_t = np.zeros(3) if offset_t is None else offset_t
    return np.ascontiguousarray(np.concatenate([np.ravel(rot), np.ravel(pos), np.ravel(offset_r), np.ravel(offset_t)]), np.float64)


class LidarMap:
    """The incremental LiDAR map (global ``ikdtree`` of LidarFrontEnd.cpp:96): Build / Add_Points / size."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib().tc2li_lidar_map_create(C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().tc2li_lidar_map_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def Build(self, points):
        points = np.ascontiguousarray(points, POINT_DTYPE)
        return _check(lib().tc2li_lidar_map_build(self._h, points.ctypes.data, len(points)))

    def Add_Points(self, points):
        points = np.ascontiguousarray(points, POINT_DTYPE)
        return _check(lib().tc2li_lidar_map_add(self._h, points.ctypes.data, len(points)))

    def size(self):
        return _check(lib().tc2li_lidar_map_size(self._h))

    def points(self):
        n = self.size()
        out = np.zeros(max(n, 1), POINT_DTYPE)
        _check(lib().tc2li_lidar_map_download(self._h, out.ctypes.data, len(out)))
        return out[:n].copy()

    def stats(self):
        """tc2li_lidar_map_stats -> dict(points, slots, grid_builds, grid_updates, tombstones, cells)."""
        out = (C.c_int32 * 6)()
        f = lib().tc2li_lidar_map_stats
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _check(f(self._h, out, 6))
        return dict(zip(("points", "slots", "grid_builds", "grid_updates", "tombstones", "cells"), [int(v) for v in out]))

    def grid(self):
        """tc2li_lidar_map_grid_download: the grid walked and checked on the host -> (cells [n], point indices [n]) of its live entries."""
        n = self.size()
        cells, idx = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        f = lib().tc2li_lidar_map_grid_download
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        live = _check(f(self._h, cells.ctypes.data, idx.ctypes.data, len(cells)))
        if live != n:
            raise Tc2liError(-2, "the grid holds %d live entries for %d points" % (live, n))
        return cells[:n].copy(), idx[:n].copy()

    def Delete_Point_Boxes(self, boxes6, stream=0):
        """boxes6: [n, 6] = min x y z, max x y z; returns the number of removed points."""
        b = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        return _check(lib().tc2li_lidar_map_delete_boxes(self._h, b.ctypes.data, len(b), C.c_void_p(stream)))

    def map_incremental(self, front_end, scan, state24, ekf_inited=True, filter_size_map_min=0.5, stream=0):
        """``map_incremental()`` for scan slot ``scan`` of ``front_end``'s last feature extraction against this map ->
        (map size, n_to_add, n_no_need)."""
        st = np.ascontiguousarray(state24, np.float64)
        na, nn = C.c_int32(0), C.c_int32(0)
        n = _check(lib().tc2li_lidar_map_incremental(front_end._h, scan, self._h, st.ctypes.data, int(ekf_inited), filter_size_map_min,
                                                     C.addressof(na), C.addressof(nn), C.c_void_p(stream)))
        return n, na.value, nn.value


def delete_point_boxes_batch(maps, boxes_per_map, stream=0):
    """``Delete_Point_Boxes`` on several maps at once: ``boxes_per_map[i]`` = [k_i, 6] boxes of map i -> removed points per map."""
    n = len(maps)
    offs = np.zeros(n + 1, np.int32)
    offs[1:] = np.cumsum([len(np.asarray(b).reshape(-1, 6)) for b in boxes_per_map])
    allb = np.ascontiguousarray(np.concatenate([np.asarray(b, np.float32).reshape(-1, 6) for b in boxes_per_map] + [np.zeros((0, 6), np.float32)]), np.float32)
    handles = (C.c_void_p * max(n, 1))(*[m._h for m in maps])
    removed = np.zeros(max(n, 1), np.int32)
    _check(lib().tc2li_lidar_map_delete_boxes_batch(n, handles, allb.ctypes.data, offs.ctypes.data, removed.ctypes.data, C.c_void_p(stream)))
    return removed[:n]


def map_incremental_batch(front_end, scans, maps, states24, ekf_inited=True, filter_size_map_min=0.5, stream=0):
    """``tc2li_lidar_map_incremental_batch`` -> (n_to_add [n], n_no_need [n], map sizes [n])."""
    n = len(maps)
    sc = np.ascontiguousarray(scans, np.int32)
    st = np.ascontiguousarray(states24, np.float64).reshape(n, 24)
    handles = (C.c_void_p * n)(*[m._h for m in maps])
    na, nn, sz = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    f = lib().tc2li_lidar_map_incremental_batch
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(front_end._h, n, sc.ctypes.data, handles, st.ctypes.data, int(ekf_inited), filter_size_map_min, na.ctypes.data, nn.ctypes.data,
             sz.ctypes.data, C.c_void_p(stream)))
    return na, nn, sz


# ---- pose plumbing between the camera thread and the LiDAR front end (row b4) ----
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def lidar_update_pose(Tcw_last7, velocity7, time_from_last_frame, Tcl7, state24):
    """``UpdateLidarPose`` -> (state24 with rot / pos replaced, pos_lid)."""
    st, pos = np.ascontiguousarray(state24, np.float64).copy(), np.zeros(3)
    f = lib().tc2li_lidar_update_pose
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(_f32(Tcw_last7).ctypes.data, _f32(velocity7).ctypes.data, time_from_last_frame, _f32(Tcl7).ctypes.data, st.ctypes.data, pos.ctypes.data))
    return st, pos


def se3_interpolate(a7, b7, t):
    out = np.zeros(7, np.float32)
    f = lib().tc2li_se3_interpolate
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    _check(f(_f32(a7).ctypes.data, _f32(b7).ctypes.data, t, out.ctypes.data))
    return out


def lidar_sync_transform(Tcw_frame7, Tcw_last7, Tcw_cur7, ratio, Tlc7, Tcl7):
    out = np.zeros(7, np.float32)
    f = lib().tc2li_lidar_sync_transform
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(_f32(Tcw_frame7).ctypes.data, _f32(Tcw_last7).ctypes.data, _f32(Tcw_cur7).ctypes.data, ratio, _f32(Tlc7).ctypes.data, _f32(Tcl7).ctypes.data,
             out.ctypes.data))
    return out


def lidar_keyframe_transform(Tcw_cur7, rel7, Tcw_refkf7, Tlc7, Tcl7):
    out = np.zeros(7, np.float32)
    f = lib().tc2li_lidar_keyframe_transform
    f.argtypes = [C.c_void_p] * 6
    _check(f(_f32(Tcw_cur7).ctypes.data, _f32(rel7).ctypes.data, _f32(Tcw_refkf7).ctypes.data, _f32(Tlc7).ctypes.data, _f32(Tcl7).ctypes.data, out.ctypes.data))
    return out


def transform_point_cloud(points, T7, stream=0):
    p = np.ascontiguousarray(points, POINT_DTYPE)
    out = np.zeros(len(p), POINT_DTYPE)
    f = lib().tc2li_transform_point_cloud
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(p.ctypes.data if len(p) else None, len(p), _f32(T7).ctypes.data, out.ctypes.data if len(p) else None, C.c_void_p(stream)))
    return out


def lidar_transform_features_batch(front_end, scans, T7, capacity=None, stream=0):
    """The selected feature clouds of scan slots ``scans`` of the last ``frontend_batch``, each moved by T7[i] -> list of POINT_DTYPE arrays."""
    sc = np.ascontiguousarray(scans, np.int32)
    T = _f32(T7).reshape(len(sc), 7)
    capacity = capacity or front_end.cap
    out = np.zeros((len(sc), capacity), POINT_DTYPE)
    npts = np.zeros(len(sc), np.int32)
    f = lib().tc2li_lidar_transform_features_batch
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _check(f(front_end._h, len(sc), sc.ctypes.data, T.ctypes.data, out.ctypes.data, capacity, npts.ctypes.data, C.c_void_p(stream)))
    return [out[i, :npts[i]].copy() for i in range(len(sc))]


class LocalMapBox(C.Structure):
    _fields_ = [("vertex_min", C.c_float * 3), ("vertex_max", C.c_float * 3), ("initialized", C.c_int32)]


def lidar_fov_segment(local_map, pos_lid, cube_len=200.0, det_range=100.0):
    """``lasermap_fov_segment`` (host): updates ``local_map`` (a LocalMapBox) and returns the boxes [k, 6] to delete."""
    pos = np.ascontiguousarray(pos_lid, np.float64)
    boxes = np.zeros((3, 6), np.float32)
    k = _check(lib().tc2li_lidar_fov_segment(C.addressof(local_map), pos.ctypes.data, cube_len, det_range, boxes.ctypes.data))
    return boxes[:k].copy()


def host_reduced_solve(Hi, S, lam, rhs):
    """``tc2li_host_reduced_solve``: the host's envelope LDL^T of an inertial window's reduced system (no GPU) -> x, or None when a pivot fails."""
    Hi = np.ascontiguousarray(Hi, np.float64)
    S = np.ascontiguousarray(S, np.float64)
    rhs = np.ascontiguousarray(rhs, np.float64)
    n, npose = len(Hi), len(S)
    x = np.zeros(n)
    f = lib().tc2li_host_reduced_solve
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    ok = _check(f(Hi.ctypes.data, S.ctypes.data if npose else None, n, npose, float(lam), rhs.ctypes.data, x.ctypes.data))
    return x if ok else None


def device_reduced_solve(Hi, S, lam, rhs, stream=0):
    """``tc2li_device_reduced_solve``: the same system through k_lvi_solve -> x, or None when a pivot fails."""
    Hi = np.ascontiguousarray(Hi, np.float64)
    S = np.ascontiguousarray(S, np.float64)
    rhs = np.ascontiguousarray(rhs, np.float64)
    n, npose = len(Hi), len(S)
    x = np.zeros(n)
    f = lib().tc2li_device_reduced_solve
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    ok = _check(f(Hi.ctypes.data, S.ctypes.data, n, npose, float(lam), rhs.ctypes.data, x.ctypes.data, C.c_void_p(stream)))
    return x if ok else None


def lidar_fov_segment_batch(local_maps, pos_lid3, cube_len=200.0, det_range=100.0):
    """``lasermap_fov_segment`` for n sensors in one call: ``local_maps`` a ctypes array ``(LocalMapBox * n)()``, ``pos_lid3`` [n, 3]
    -> (boxes [n, 3, 6] float32, counts [n] int32)."""
    n = len(local_maps)
    pos = np.ascontiguousarray(pos_lid3, np.float64).reshape(n, 3)
    boxes = np.zeros((n, 3, 6), np.float32)
    counts = np.zeros(n, np.int32)
    _check(lib().tc2li_lidar_fov_segment_batch(C.addressof(local_maps), pos.ctypes.data, n, cube_len, det_range, boxes.ctypes.data, counts.ctypes.data))
    return boxes, counts


class LidarFrontEnd:
    """Stage functions of the camera-LiDAR front end (Preprocess::process, VoxelGrid::filter, feature_extraction)."""

    def __init__(self, max_points_per_scan=140000, max_scans=1):
        self._h = C.c_void_p()
        self.cap = max_points_per_scan
        self.max_scans = max_scans
        _check(lib().tc2li_lidar_create(max_points_per_scan, max_scans, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().tc2li_lidar_destroy(self._h)
            self._h = C.c_void_p()

    def last_timings(self):
        """Device ms of the last frontend_batch: preprocess, voxel hashing, voxel centroids, 5-NN + plane fit, selection, total,
        then the 5-NN stage split into k_knn_plane and k_knn_hard."""
        t = np.zeros(8, np.float32)
        _check(lib().tc2li_lidar_last_timings(self._h, t.ctypes.data))
        return t

    __del__ = close

    def process(self, raw, point_filter_num=2, blind=2.0, time_unit_scale=1e-3):
        raw = np.ascontiguousarray(raw, VELODYNE_DTYPE)
        out = np.zeros(max(len(raw), 1), POINT_DTYPE)
        n = _check(lib().tc2li_lidar_preprocess(self._h, raw.ctypes.data, len(raw), point_filter_num, blind, time_unit_scale,
                                                out.ctypes.data, len(out)))
        return out[:n].copy()

    def voxel_filter(self, points, leaf=0.5):
        points = np.ascontiguousarray(points, POINT_DTYPE)
        out = np.zeros(max(len(points), 1), POINT_DTYPE)
        n = _check(lib().tc2li_lidar_voxel_filter(self._h, points.ctypes.data, len(points), leaf, out.ctypes.data, len(out)))
        return out[:n].copy()

    def undistort(self, points, imu_poses22, end_state24):
        """ImuProcess::UndistortPcl, the point part: time sort + compensation into the scan-end frame."""
        pts = np.ascontiguousarray(points, POINT_DTYPE).copy()
        poses = np.ascontiguousarray(imu_poses22, np.float64).reshape(-1, 22)
        st = np.ascontiguousarray(end_state24, np.float64)
        _check(lib().tc2li_lidar_undistort(self._h, pts.ctypes.data, len(pts), poses.ctypes.data, len(poses), st.ctypes.data))
        return pts

    def eskf_update(self, lidar_map, feats_down_body, state36, P, R=0.001, max_iter=4, limit=None, extrinsic_est_en=False):
        """``esekf::update_iterated_dyn_share_modified`` with ``h_share_model`` -> (state36, P [23, 23], EskfStats).  state36: pos 3,
        rot 9, vel 3, bg 3, ba 3, grav 3, offset_R_L_I 9, offset_T_L_I 3."""
        body = np.ascontiguousarray(feats_down_body, POINT_DTYPE)
        st = np.ascontiguousarray(state36, np.float64).copy()
        Pm = np.ascontiguousarray(P, np.float64).reshape(23, 23).copy()
        lim = np.ascontiguousarray(np.full(23, 0.001) if limit is None else limit, np.float64)
        stats = EskfStats()
        f = lib().tc2li_lidar_eskf_update
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _check(f(self._h, lidar_map._h, body.ctypes.data, len(body), st.ctypes.data, Pm.ctypes.data, R, max_iter, lim.ctypes.data,
                 int(extrinsic_est_en), C.addressof(stats)))
        return st, Pm, stats

    def time_sort(self, points, depth_limit=-1):
        """The permutation ``std::sort(points, time_list)`` leaves (UndistortPcl's time sort), computed by the device kernel of the
        inertial batch -> (perm [n], reached_depth_limit)."""
        pts = np.ascontiguousarray(points, POINT_DTYPE)
        perm = np.zeros(max(len(pts), 1), np.int32)
        fb = _check(lib().tc2li_device_time_sort(self._h, pts.ctypes.data, len(pts), int(depth_limit), perm.ctypes.data))
        return perm[:len(pts)], bool(fb)

    def inertial_prepare_batch(self, dev_raw_ptr, raw_offsets, point_filter_num=2, blind=2.0, time_unit_scale=1e-3, stream=0):
        """``tc2li_lidar_inertial_prepare_batch``: Preprocess::process + the order of UndistortPcl's time sort for the scans, kept in this handle
        for the next ``inertial_frontend_batch(None, ...)``."""
        raw_offsets = np.ascontiguousarray(raw_offsets, np.int32)
        f = lib().tc2li_lidar_inertial_prepare_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_void_p]
        return _check(f(self._h, len(raw_offsets) - 1, C.c_void_p(dev_raw_ptr), raw_offsets.ctypes.data, point_filter_num, blind, time_unit_scale,
                        C.c_void_p(stream)))

    def inertial_frontend_batch(self, dev_raw_ptr, raw_offsets, maps, states36, Ps, imus, times, cov12, last6=None, point_filter_num=2, blind=2.0,
                                time_unit_scale=1e-3, leaf=0.5, R=0.001, max_iter=3, limit=None, extrinsic_est_en=False, stream=0):
        """``LidarInertialProcess`` for a batch of sequences (tc2li_lidar_inertial_frontend_batch).  states36 [S, 36] (pos 3, rot 9, vel 3, bg 3,
        ba 3, grav 3, offset_R_L_I 9, offset_T_L_I 3), Ps [S, 23, 23], imus: list of [K, 7] sample arrays (t, acc, gyr), times [S, 4] =
        pcl_beg_time, pcl_end_time, last_lidar_end_time, acc_scale, last6 [S, 6] = acc_s_last, angvel_last ->
        (states36, Ps, list of EskfStats, n_preprocessed, n_downsampled, last6).  dev_raw_ptr None: the scans inertial_prepare_batch left in
        this handle."""
        S = len(raw_offsets) - 1
        raw_offsets = np.ascontiguousarray(raw_offsets, np.int32)
        st = np.ascontiguousarray(states36, np.float64).reshape(S, 36).copy()
        Pm = np.ascontiguousarray(Ps, np.float64).reshape(S, 529).copy()
        times = np.ascontiguousarray(times, np.float64).reshape(S, 4)
        last = np.zeros((S, 6)) if last6 is None else np.ascontiguousarray(last6, np.float64).reshape(S, 6).copy()
        cov = np.ascontiguousarray(cov12, np.float64)
        lim = np.ascontiguousarray(np.full(23, 0.001) if limit is None else limit, np.float64)
        imus = [np.ascontiguousarray(x, np.float64).reshape(-1, 7) for x in imus]
        scans = (LidarInertialScan * S)()
        for s in range(S):
            sc = scans[s]
            sc.imu, sc.n_imu = imus[s].ctypes.data, len(imus[s])
            sc.pcl_beg_time, sc.pcl_end_time, sc.last_lidar_end_time, sc.acc_scale = [float(v) for v in times[s]]
            for k in range(3):
                sc.acc_s_last[k], sc.angvel_last[k] = last[s, k], last[s, 3 + k]
            C.memmove(C.addressof(sc.state), st[s].ctypes.data, 36 * 8)
            sc.P = Pm[s].ctypes.data
        handles = (C.c_void_p * S)(*[m._h for m in maps])
        f = lib().tc2li_lidar_inertial_frontend_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _check(f(self._h, S, C.c_void_p(dev_raw_ptr or 0), raw_offsets.ctypes.data, point_filter_num, blind, time_unit_scale, leaf, handles, scans,
                 cov.ctypes.data, R, max_iter, lim.ctypes.data, int(extrinsic_est_en), C.c_void_p(stream)))
        stats, n_pre, n_down = [], np.zeros(S, np.int32), np.zeros(S, np.int32)
        for s in range(S):
            sc = scans[s]
            C.memmove(st[s].ctypes.data, C.addressof(sc.state), 36 * 8)
            e = EskfStats()
            C.memmove(C.addressof(e), C.addressof(sc.stats), C.sizeof(EskfStats))
            stats.append(e)
            n_pre[s], n_down[s] = sc.n_preprocessed, sc.n_downsampled
            for k in range(3):
                last[s, k], last[s, 3 + k] = sc.acc_s_last[k], sc.angvel_last[k]
        return st, Pm.reshape(S, 23, 23), stats, n_pre, n_down, last

    def feature_extraction(self, lidar_map, feats_down_body, state24):
        body = np.ascontiguousarray(feats_down_body, POINT_DTYPE)
        n = len(body)
        m1 = max(n, 1)
        world = np.zeros(m1, POINT_DTYPE)
        sel = np.zeros(m1, np.uint8)
        normvec = np.zeros(m1, POINT_DTYPE)
        near = np.zeros((m1, 5), POINT_DTYPE)
        dist = np.zeros((m1, 5), np.float32)
        nfound = np.zeros(m1, np.int32)
        ori = np.zeros(m1, POINT_DTYPE)
        corr = np.zeros(m1, POINT_DTYPE)
        state24 = np.ascontiguousarray(state24, np.float64)
        m = _check(lib().tc2li_lidar_feature_extraction(self._h, lidar_map._h, body.ctypes.data, n, state24.ctypes.data,
                                                        world.ctypes.data, sel.ctypes.data, normvec.ctypes.data, near.ctypes.data,
                                                        dist.ctypes.data, nfound.ctypes.data, ori.ctypes.data, corr.ctypes.data, m1))
        return dict(world=world[:n], selected=sel[:n], normvec=normvec[:n], nearest=near[:n], sqdist=dist[:n], nfound=nfound[:n],
                    cloud_ori=ori[:m], corr_normvect=corr[:m], effct_feat_num=m)

    def frontend_batch(self, dev_raw_ptr, raw_offsets, maps, states, point_filter_num=2, blind=2.0, time_unit_scale=1e-3,
                       leaf=0.5, stream=0, want_points=True, capacity=None):
        n_scans = len(raw_offsets) - 1
        raw_offsets = np.ascontiguousarray(raw_offsets, np.int32)
        states = np.ascontiguousarray(states, np.float64).reshape(n_scans, 24)
        handles = (C.c_void_p * n_scans)(*[m._h for m in maps])
        counts = np.zeros((3, n_scans), np.int32)
        capacity = capacity or self.cap
        ori = corr = None
        if want_points:
            ori = np.zeros((n_scans, capacity), POINT_DTYPE)
            corr = np.zeros((n_scans, capacity), POINT_DTYPE)
        _check(lib().tc2li_lidar_frontend_batch(self._h, n_scans, C.c_void_p(dev_raw_ptr), raw_offsets.ctypes.data,
                                                point_filter_num, blind, time_unit_scale, leaf, handles, states.ctypes.data,
                                                counts[0].ctypes.data, counts[1].ctypes.data, counts[2].ctypes.data,
                                                ori.ctypes.data if want_points else None,
                                                corr.ctypes.data if want_points else None, capacity, C.c_void_p(stream)))
        return counts, ori, corr


# ---- optimisation back end ----------------------------------------------------------------------------------------
BA_EDGE_DTYPE = np.dtype([("point", "<i4"), ("pose", "<i4"), ("u", "<f8"), ("v", "<f8"), ("u_right", "<f8"), ("inv_sigma2", "<f8")])


def pack_ba_edges(edges6):
    """[E, 6] float array (point, pose, u, v, uR, invSigma2) -> structured tc2li_ba_edge array."""
    e = np.asarray(edges6, np.float64).reshape(-1, 6)
    out = np.zeros(len(e), BA_EDGE_DTYPE)
    out["point"], out["pose"] = e[:, 0].astype(np.int32), e[:, 1].astype(np.int32)
    out["u"], out["v"], out["u_right"], out["inv_sigma2"] = e[:, 2], e[:, 3], e[:, 4], e[:, 5]
    return out


def pose_optimization(pose7, Xw, edges, cam5):
    """``Optimizer::PoseOptimization`` -> (pose7, outlier mask, inliers)."""
    pose = np.ascontiguousarray(pose7, np.float64).copy()
    Xw = np.ascontiguousarray(Xw, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    out = np.zeros(max(len(edges), 1), np.uint8)
    inl = _check(lib().tc2li_pose_optimization(pose.ctypes.data, Xw.ctypes.data, edges.ctypes.data, len(edges), cam5.ctypes.data,
                                               out.ctypes.data))
    return pose, out[:len(edges)], inl


def pose_optimization_batch(poses7, edge_offsets, Xw, edges, cam5, stream=0):
    poses = np.ascontiguousarray(poses7, np.float64).copy()
    offs = np.ascontiguousarray(edge_offsets, np.int32)
    Xw = np.ascontiguousarray(Xw, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    n = len(offs) - 1
    out = np.zeros(max(len(edges), 1), np.uint8)
    inl = np.zeros(n, np.int32)
    _check(lib().tc2li_pose_optimization_batch(n, poses.ctypes.data, offs.ctypes.data, Xw.ctypes.data, edges.ctypes.data,
                                               cam5.ctypes.data, out.ctypes.data, inl.ctypes.data, C.c_void_p(stream)))
    return poses, out[:len(edges)], inl


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("n_free_poses", C.c_int32), ("pad_", C.c_int32),
                ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("final_lambda", C.c_double)]


def local_bundle_adjustment(poses7, fixed, points3, edges, cam5, iterations=10, lambda_init=0.0, stop_flag=None, stream=0):
    """The optimisation of ``Optimizer::LocalBundleAdjustment`` -> (poses7, points3, chi2, depth_positive, stats)."""
    poses = np.ascontiguousarray(poses7, np.float64).copy()
    pts = np.ascontiguousarray(points3, np.float64).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    chi2 = np.zeros(max(len(edges), 1))
    dpos = np.zeros(max(len(edges), 1), np.uint8)
    stats = BaStats()
    stop_ptr = stop_flag.ctypes.data if stop_flag is not None else None
    _check(lib().tc2li_local_bundle_adjustment(poses.ctypes.data, fixed.ctypes.data, len(poses), pts.ctypes.data, len(pts),
                                               edges.ctypes.data, len(edges), cam5.ctypes.data, iterations, lambda_init, stop_ptr,
                                               chi2.ctypes.data, dpos.ctypes.data, C.byref(stats), C.c_void_p(stream)))
    return poses, pts, chi2[:len(edges)], dpos[:len(edges)], stats


# ---- IMU pre-integration (IMU::Preintegrated, Tracking::PreintegrateIMU / PredictStateIMU) -----------------------------
IMU_SAMPLE_DTYPE = np.dtype([("t", "<f8"), ("a", "<f4", (3,)), ("w", "<f4", (3,))])


class ImuBias(C.Structure):
    _fields_ = [("bax", C.c_float), ("bay", C.c_float), ("baz", C.c_float), ("bwx", C.c_float), ("bwy", C.c_float), ("bwz", C.c_float)]


class InertialInitStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("final_lambda", C.c_double)]


def imu_init_gravity(Rwb, twb, pres):
    """First estimate of ``LocalMapping::InitializeIMU``: keyframes in temporal order, ``pres[i]`` = Preintegrated of keyframe i from i - 1
    (``pres[0]`` ignored) -> (velocities [N, 3] float32, Rwg [3, 3] float32)."""
    R, t = np.ascontiguousarray(Rwb, np.float32).reshape(-1, 9), np.ascontiguousarray(twb, np.float32).reshape(-1, 3)
    n = len(R)
    ptrs = (C.c_void_p * n)(*[None if (i == 0 or pres[i] is None) else C.addressof(pres[i].p) for i in range(n)])
    vel, Rwg = np.zeros((n, 3), np.float32), np.zeros(9, np.float32)
    f = lib().tc2li_imu_init_gravity
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(n, R.ctypes.data, t.ctypes.data, ptrs, vel.ctypes.data, Rwg.ctypes.data))
    return vel, Rwg.reshape(3, 3)


def inertial_scale_refinement(Rwb, twb, vel, bg, ba, pres, Rwg, scale, iterations=10):
    """``Optimizer::InertialOptimization(pMap, Rwg, scale)`` (ScaleRefinement) -> (Rwg, scale, iterations, (chi2 before, after))."""
    R, t = np.ascontiguousarray(Rwb, np.float64).reshape(-1, 9), np.ascontiguousarray(twb, np.float64).reshape(-1, 3)
    n = len(R)
    v, g, a = [np.ascontiguousarray(x, np.float64).reshape(n, 3) for x in (vel, bg, ba)]
    ptrs = (C.c_void_p * n)(*[None if (i == 0 or pres[i] is None) else C.addressof(pres[i].p) for i in range(n)])
    Rg, s, chi = np.ascontiguousarray(Rwg, np.float64).reshape(9).copy(), C.c_double(scale), np.zeros(2)
    f = lib().tc2li_inertial_scale_refinement
    f.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
    it = _check(f(n, R.ctypes.data, t.ctypes.data, v.ctypes.data, g.ctypes.data, a.ctypes.data, ptrs, Rg.ctypes.data, C.addressof(s), iterations, chi.ctypes.data))
    return Rg.reshape(3, 3), s.value, it, (chi[0], chi[1])


def inertial_optimization(Rwb, twb, vel, pres, Rwg, scale, bg, ba, mono=False, fixed_vel=False, prior_g=1e2, prior_a=1e6, iterations=200):
    """``Optimizer::InertialOptimization`` (IMU initialisation) -> (velocities [N, 3], Rwg, scale, bg, ba, InertialInitStats)."""
    R, t = np.ascontiguousarray(Rwb, np.float64).reshape(-1, 9), np.ascontiguousarray(twb, np.float64).reshape(-1, 3)
    n = len(R)
    v = np.ascontiguousarray(vel, np.float64).reshape(n, 3).copy()
    ptrs = (C.c_void_p * n)(*[None if (i == 0 or pres[i] is None) else C.addressof(pres[i].p) for i in range(n)])
    Rg, g, a = np.ascontiguousarray(Rwg, np.float64).reshape(9).copy(), np.ascontiguousarray(bg, np.float64).copy(), np.ascontiguousarray(ba, np.float64).copy()
    s, st = C.c_double(scale), InertialInitStats()
    f = lib().tc2li_inertial_optimization
    f.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]
    _check(f(n, R.ctypes.data, t.ctypes.data, v.ctypes.data, ptrs, Rg.ctypes.data, C.addressof(s), g.ctypes.data, a.ctypes.data, int(mono), int(fixed_vel),
             prior_g, prior_a, iterations, C.addressof(st)))
    return v, Rg.reshape(3, 3), s.value, g, a, st


class PreintegratedPOD(C.Structure):
    _fields_ = [("dT", C.c_float), ("n_measurements", C.c_int32), ("dR", C.c_float * 9), ("dV", C.c_float * 3), ("dP", C.c_float * 3),
                ("JRg", C.c_float * 9), ("JVg", C.c_float * 9), ("JVa", C.c_float * 9), ("JPg", C.c_float * 9), ("JPa", C.c_float * 9),
                ("avgA", C.c_float * 3), ("avgW", C.c_float * 3), ("C", C.c_float * 225), ("noise", C.c_float * 6),
                ("noise_walk", C.c_float * 6), ("bias", ImuBias)]


class Preintegrated:
    """Mirror of ``IMU::Preintegrated`` (SF/include/ImuTypes.h:141-235)."""

    def __init__(self, bias6, ng, na, ngw, naw):
        self.p = PreintegratedPOD()
        b = ImuBias(*[float(x) for x in bias6])
        _check(lib().tc2li_imu_preintegrated_init(C.addressof(self.p), C.addressof(b), ng, na, ngw, naw))

    def IntegrateNewMeasurement(self, acc, ang_vel, dt):
        a, w = np.ascontiguousarray(acc, np.float32), np.ascontiguousarray(ang_vel, np.float32)
        _check(lib().tc2li_imu_integrate(C.addressof(self.p), a.ctypes.data, w.ctypes.data, float(dt)))

    def preintegrate(self, samples, t_prev, t_cur):
        """The loop of Tracking::PreintegrateIMU over mvImuFromLastFrame -> number of integration steps."""
        s = np.ascontiguousarray(samples, IMU_SAMPLE_DTYPE)
        return _check(lib().tc2li_imu_preintegrate(C.addressof(self.p), s.ctypes.data, len(s), float(t_prev), float(t_cur)))

    def delta(self, bias6):
        """GetDeltaRotation / GetDeltaVelocity / GetDeltaPosition at another bias."""
        b = ImuBias(*[float(x) for x in bias6])
        dR, dV, dP = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        _check(lib().tc2li_imu_delta(C.addressof(self.p), C.addressof(b), dR.ctypes.data, dV.ctypes.data, dP.ctypes.data))
        return dR.reshape(3, 3), dV, dP

    def predict_state(self, bias6, Rwb1, twb1, Vwb1):
        """Tracking::PredictStateIMU -> (Rwb2, twb2, Vwb2)."""
        b = ImuBias(*[float(x) for x in bias6])
        R1, t1, v1 = [np.ascontiguousarray(a, np.float32) for a in (Rwb1, twb1, Vwb1)]
        R2, t2, v2 = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        _check(lib().tc2li_imu_predict_state(C.addressof(self.p), C.addressof(b), R1.ctypes.data, t1.ctypes.data, v1.ctypes.data,
                                             R2.ctypes.data, t2.ctypes.data, v2.ctypes.data))
        return R2.reshape(3, 3), t2, v2

    def fields(self):
        g = lambda name, shape: np.array(getattr(self.p, name), np.float32).reshape(shape)
        return dict(dT=self.p.dT, dR=g("dR", (3, 3)), dV=g("dV", 3), dP=g("dP", 3), JRg=g("JRg", (3, 3)), JVg=g("JVg", (3, 3)),
                    JVa=g("JVa", (3, 3)), JPg=g("JPg", (3, 3)), JPa=g("JPa", (3, 3)), avgA=g("avgA", 3), avgW=g("avgW", 3), C=g("C", (15, 15)))


def lidar_imu_propagate(state36, imu7, beg, end, last_end, acc_scale, last6):
    """Forward propagation of UndistortPcl (host) -> (end state [36], poses [K, 22], updated last6)."""
    st = np.ascontiguousarray(state36, np.float64).copy()
    imu = np.ascontiguousarray(imu7, np.float64).reshape(-1, 7)
    last = np.ascontiguousarray(last6, np.float64).copy()
    poses = np.zeros((len(imu) + 2, 22))
    k = _check(lib().tc2li_lidar_imu_propagate(st.ctypes.data, imu.ctypes.data, len(imu), beg, end, last_end, acc_scale, last.ctypes.data,
                                               last.ctypes.data + 24, poses.ctypes.data, len(poses)))
    return st, poses[:k], last


class KeyframeView(C.Structure):
    """tc2li_keyframe_view"""
    _fields_ = [("n", C.c_int32), ("n_nodes", C.c_int32), ("keys", C.c_void_p), ("descriptors", C.c_void_p), ("u_right", C.c_void_p),
                ("depth", C.c_void_p), ("has_point", C.c_void_p), ("fv_node", C.c_void_p), ("fv_offset", C.c_void_p), ("fv_index", C.c_void_p),
                ("pose7", C.c_float * 7), ("pad_", C.c_float)]


class NewMapPoint(C.Structure):
    """tc2li_new_map_point"""
    _fields_ = [("idx1", C.c_int32), ("neighbour", C.c_int32), ("idx2", C.c_int32), ("stereo", C.c_int32), ("x3D", C.c_float * 3), ("pad_", C.c_float)]


def pack_keyframe_views(items):
    """items: dicts with keys (KEYPOINT_DTYPE), descriptors, u_right, depth, has_point, fv_node, fv_offset, fv_index, pose7 ->
    (ctypes array of tc2li_keyframe_view, keep-alive list)."""
    arr = (KeyframeView * max(len(items), 1))()
    keep = []
    for i, it in enumerate(items):
        k = np.ascontiguousarray(it["keys"], KEYPOINT_DTYPE)
        d = np.ascontiguousarray(it["descriptors"], np.uint8).reshape(-1, 32)
        ur, z = np.ascontiguousarray(it["u_right"], np.float32), np.ascontiguousarray(it["depth"], np.float32)
        hp = np.ascontiguousarray(it["has_point"], np.uint8)
        fn, fo, fi = [np.ascontiguousarray(it[f], np.int32) for f in ("fv_node", "fv_offset", "fv_index")]
        keep.append((k, d, ur, z, hp, fn, fo, fi))
        arr[i].n, arr[i].n_nodes = len(k), len(fn)
        arr[i].keys, arr[i].descriptors, arr[i].u_right, arr[i].depth, arr[i].has_point = (k.ctypes.data, d.ctypes.data, ur.ctypes.data, z.ctypes.data,
                                                                                          hp.ctypes.data)
        arr[i].fv_node, arr[i].fv_offset, arr[i].fv_index = fn.ctypes.data, fo.ctypes.data, fi.ctypes.data
        arr[i].pose7 = (C.c_float * 7)(*[float(v) for v in it["pose7"]])
    return arr, keep


def search_for_triangulation(kf1, kf2, cam5, scale_factors, level_sigma2, only_stereo=False, coarse=False, check_orientation=False, stream=0):
    """``ORBmatcher::SearchForTriangulation`` -> (nmatches, match12 [n1])."""
    arr, keep = pack_keyframe_views([kf1, kf2])
    cam5 = np.ascontiguousarray(cam5, np.float64)
    sf, sg = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
    match = np.full(max(arr[0].n, 1), -1, np.int32)
    f = lib().tc2li_search_for_triangulation
    f.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
    n = _check(f(C.addressof(arr), C.addressof(arr) + C.sizeof(KeyframeView), cam5.ctypes.data, sf.ctypes.data, sg.ctypes.data, len(sf), int(only_stereo),
                 int(coarse), int(check_orientation), match.ctypes.data, C.c_void_p(stream)))
    del keep
    return n, match[:arr[0].n]


def create_new_map_points(cur, neighbours, cam5, mb, scale_factors, level_sigma2, scale_factor=1.2, inertial=False, far_points=False,
                          th_far_points=0.0, coarse=False, stream=0):
    """The geometric loop of ``LocalMapping::CreateNewMapPoints`` -> (idx [k, 4] = idx1, neighbour, idx2, stereo; x3D [k, 3])."""
    arr, keep = pack_keyframe_views([cur] + list(neighbours))
    cam5 = np.ascontiguousarray(cam5, np.float64)
    sf, sg = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
    cap = max(arr[0].n, 1)
    pts = (NewMapPoint * cap)()
    f = lib().tc2li_create_new_map_points
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float,
                  C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = _check(f(C.addressof(arr), C.addressof(arr) + C.sizeof(KeyframeView), len(neighbours), cam5.ctypes.data, mb, sf.ctypes.data, sg.ctypes.data,
                 len(sf), scale_factor, int(inertial), int(far_points), th_far_points, int(coarse), C.addressof(pts), cap, C.c_void_p(stream)))
    del keep
    idx = np.array([[pts[k].idx1, pts[k].neighbour, pts[k].idx2, pts[k].stereo] for k in range(n)], np.int32).reshape(-1, 4)
    x3D = np.array([list(pts[k].x3D) for k in range(n)], np.float32).reshape(-1, 3)
    return idx, x3D


def track_local_map_batch(ext, n_frames, keypoints, u_right, poses7, held, held_Xw, local_points, local_offsets, cam5, th=1.0, far_points=False,
                          th_far=0.0, stream=0, out=None):
    """``tc2li_track_local_map_batch`` on the features of the last ``extract_batch_dev`` call ->
    (poses7 double [F, 7], local_of_keypoint [F, cap], outlier [F, cap], n_matches [F], n_inliers [F])."""
    kps = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE)
    cap = kps.shape[1]
    ur = np.ascontiguousarray(u_right, np.float32)
    p7 = np.ascontiguousarray(poses7, np.float32).reshape(n_frames, 7)
    h = np.ascontiguousarray(held, np.uint8).reshape(n_frames, cap)
    hx = np.ascontiguousarray(held_Xw, np.float32).reshape(n_frames, cap, 3)
    pts = np.ascontiguousarray(local_points, MAP_POINT_DTYPE)
    off = np.ascontiguousarray(local_offsets, np.int32)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    if out is None:
        out = (np.zeros((n_frames, 7)), np.full((n_frames, cap), -1, np.int32), np.zeros((n_frames, cap), np.uint8), np.zeros(n_frames, np.int32),
               np.zeros(n_frames, np.int32))
    out_p, lk, ol, nm, inl = out
    f = lib().tc2li_track_local_map_batch
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                  C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(ext._h, n_frames, kps.ctypes.data, ur.ctypes.data, cap, p7.ctypes.data, h.ctypes.data, hx.ctypes.data, pts.ctypes.data if len(pts) else None,
             off.ctypes.data, cam5.ctypes.data, th, int(far_points), th_far, out_p.ctypes.data, lk.ctypes.data, ol.ctypes.data, nm.ctypes.data,
             inl.ctypes.data, C.c_void_p(stream)))
    return out_p, lk, ol, nm, inl


def search_local_points_batch(ext, n_frames, keypoints, u_right, poses7, held, held_Xw, local_points, local_offsets, cam5, th=1.0, far_points=False,
                              th_far=0.0, stream=0, out=None):
    """``tc2li_search_local_points_batch`` (Tracking::SearchLocalPoints alone: the camera-LiDAR-inertial configuration's TrackLocalMap optimises with
    PoseInertialOptimization) on the features of the last ``extract_batch_dev`` call -> (local_of_keypoint [F, cap], n_matches [F])."""
    kps = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE)
    cap = kps.shape[1]
    ur = np.ascontiguousarray(u_right, np.float32)
    p7 = np.ascontiguousarray(poses7, np.float32).reshape(n_frames, 7)
    h = np.ascontiguousarray(held, np.uint8).reshape(n_frames, cap)
    hx = np.ascontiguousarray(held_Xw, np.float32).reshape(n_frames, cap, 3)
    pts = np.ascontiguousarray(local_points, MAP_POINT_DTYPE)
    off = np.ascontiguousarray(local_offsets, np.int32)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    if out is None:
        out = (np.full((n_frames, cap), -1, np.int32), np.zeros(n_frames, np.int32))
    lk, nm = out
    f = lib().tc2li_search_local_points_batch
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                  C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(ext._h, n_frames, kps.ctypes.data, ur.ctypes.data, cap, p7.ctypes.data, h.ctypes.data, hx.ctypes.data, pts.ctypes.data if len(pts) else None,
             off.ctypes.data, cam5.ctypes.data, th, int(far_points), th_far, lk.ctypes.data, nm.ctypes.data, C.c_void_p(stream)))
    return lk, nm


def fuse_search(keys, desc, u_right, cols, rows, pose7, cam4, bf, scale_factors, inv_level_sigma2, log_scale_factor, points, valid, th=3.0, stream=0):
    """``ORBmatcher::Fuse``, the search part -> (n_fused, best_idx [m], best_dist [m]); points: MAP_POINT_DTYPE."""
    k = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    d = np.ascontiguousarray(desc, np.uint8)
    ur = np.ascontiguousarray(u_right, np.float32)
    fv = FrameView(k.ctypes.data, d.ctypes.data, ur.ctypes.data, None, len(k), 0.0, float(cols), 0.0, float(rows))
    pts = np.ascontiguousarray(points, MAP_POINT_DTYPE)
    val = np.ascontiguousarray(valid, np.uint8)
    sf, isg = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(inv_level_sigma2, np.float32)
    p7, c4 = np.ascontiguousarray(pose7, np.float32), np.ascontiguousarray(cam4, np.float32)
    m = len(pts)
    bi, bd = np.full(max(m, 1), -1, np.int32), np.zeros(max(m, 1), np.int32)
    f = lib().tc2li_fuse_search
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                  C.c_void_p, C.c_void_p, C.c_void_p]
    nf = _check(f(C.addressof(fv), p7.ctypes.data, c4.ctypes.data, bf, sf.ctypes.data, isg.ctypes.data, len(sf), log_scale_factor, pts.ctypes.data,
                  val.ctypes.data, m, th, bi.ctypes.data, bd.ctypes.data, C.c_void_p(stream)))
    return nf, bi[:m], bd[:m]


def map_points_refresh(obs_off, descriptors, centres, positions, ref_centres, level_scale, last_scale, stream=0):
    """``MapPoint::ComputeDistinctiveDescriptors`` + ``UpdateNormalAndDepth`` for a flat list of points ->
    (best_obs, normals [n, 3], min_distance, max_distance)."""
    off = np.ascontiguousarray(obs_off, np.int32)
    n = len(off) - 1
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
    c = np.ascontiguousarray(centres, np.float32).reshape(-1, 3)
    pos, ref = np.ascontiguousarray(positions, np.float32).reshape(-1, 3), np.ascontiguousarray(ref_centres, np.float32).reshape(-1, 3)
    ls = np.ascontiguousarray(level_scale, np.float32)
    best, normals, mn, mx = np.full(max(n, 1), -1, np.int32), np.zeros((max(n, 1), 3), np.float32), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    f = lib().tc2li_map_points_refresh
    f.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 5
    _check(f(n, off.ctypes.data, d.ctypes.data if len(d) else None, c.ctypes.data if len(c) else None, pos.ctypes.data, ref.ctypes.data, ls.ctypes.data,
             float(last_scale), best.ctypes.data, normals.ctypes.data, mn.ctypes.data, mx.ctypes.data, C.c_void_p(stream)))
    return best[:n], normals[:n], mn[:n], mx[:n]


class EskfStats(C.Structure):
    """tc2li_eskf_stats"""
    _fields_ = [("calls", C.c_int32), ("effct_feat_num", C.c_int32), ("searches", C.c_int32), ("converged", C.c_int32), ("finished", C.c_int32),
                ("pad_", C.c_int32), ("res_mean_last", C.c_double)]


class LidarInertialScan(C.Structure):
    """tc2li_lidar_inertial_scan (state = tc2li_imu_state: pos 3, rot 9, vel 3, bg 3, ba 3, grav 3, offset_R_L_I 9, offset_T_L_I 3)"""
    _fields_ = [("imu", C.c_void_p), ("n_imu", C.c_int32), ("pad_", C.c_int32), ("pcl_beg_time", C.c_double), ("pcl_end_time", C.c_double),
                ("last_lidar_end_time", C.c_double), ("acc_scale", C.c_double), ("acc_s_last", C.c_double * 3), ("angvel_last", C.c_double * 3),
                ("state", C.c_double * 36), ("P", C.c_void_p), ("stats", EskfStats), ("n_preprocessed", C.c_int32), ("n_downsampled", C.c_int32)]


def eskf_predict(state36, P, Q, acc, gyr, dt):
    """One ``esekf::predict`` step (host) -> (state36, P)."""
    st = np.ascontiguousarray(state36, np.float64).copy()
    Pm = np.ascontiguousarray(P, np.float64).reshape(23, 23).copy()
    Qm = np.ascontiguousarray(Q, np.float64).reshape(12, 12)
    a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
    f = lib().tc2li_eskf_predict
    f.argtypes = [C.c_void_p] * 5 + [C.c_double]
    _check(f(st.ctypes.data, Pm.ctypes.data, Qm.ctypes.data, a.ctypes.data, g.ctypes.data, dt))
    return st, Pm


def lidar_imu_propagate_cov(state36, P, cov12, imu7, beg, end, last_end, acc_scale, last6):
    """Forward propagation of UndistortPcl with the covariance (host) -> (end state [36], P, poses [K, 22], updated last6)."""
    st = np.ascontiguousarray(state36, np.float64).copy()
    Pm = np.ascontiguousarray(P, np.float64).reshape(23, 23).copy()
    cov = np.ascontiguousarray(cov12, np.float64)
    imu = np.ascontiguousarray(imu7, np.float64).reshape(-1, 7)
    last = np.ascontiguousarray(last6, np.float64).copy()
    poses = np.zeros((len(imu) + 2, 22))
    f = lib().tc2li_lidar_imu_propagate_cov
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_int]
    k = _check(f(st.ctypes.data, Pm.ctypes.data, cov.ctypes.data, imu.ctypes.data, len(imu), beg, end, last_end, acc_scale, last.ctypes.data,
                 last.ctypes.data + 24, poses.ctypes.data, len(poses)))
    return st, Pm, poses[:k], last


class InertialLink(C.Structure):
    """tc2li_inertial_link"""
    _fields_ = [("kf1", C.c_int32), ("kf2", C.c_int32), ("robust", C.c_int32), ("pad_", C.c_int32), ("info_scale", C.c_double),
                ("preintegrated", C.c_void_p)]


def local_inertial_bundle_adjustment(kf33, fixed, has_imu, calib24, points3, edges, link4, preintegrated, cam5, iterations=10, lambda_init=1.0,
                                     stop_flag=None, stream=0):
    """The optimisation of ``Optimizer::LocalInertialBA``.  kf33: [K, 33] = Rcw 9, tcw 3, Rwb 9, twb 3, velocity 3, gyro bias 3,
    acc bias 3 per keyframe; link4: [L, 4] = kf1, kf2, robust, info_scale; preintegrated: list of ``Preintegrated`` (one per
    link) -> (kf33, points3, chi2, depth_positive, stats)."""
    kf = np.ascontiguousarray(kf33, np.float64).copy()
    pts = np.ascontiguousarray(points3, np.float64).copy()
    fixed, has_imu = np.ascontiguousarray(fixed, np.uint8), np.ascontiguousarray(has_imu, np.uint8)
    calib24, cam5 = np.ascontiguousarray(calib24, np.float64), np.ascontiguousarray(cam5, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    link4 = np.ascontiguousarray(link4, np.float64).reshape(-1, 4)
    links = (InertialLink * max(len(link4), 1))()
    for l, row in enumerate(link4):
        links[l] = InertialLink(int(row[0]), int(row[1]), int(row[2] != 0), 0, float(row[3]), C.addressof(preintegrated[l].p))
    chi2 = np.zeros(max(len(edges), 1))
    dpos = np.zeros(max(len(edges), 1), np.uint8)
    stats = BaStats()
    stop_ptr = stop_flag.ctypes.data if stop_flag is not None else None
    _check(lib().tc2li_local_inertial_bundle_adjustment(kf.ctypes.data, fixed.ctypes.data, has_imu.ctypes.data, len(kf), calib24.ctypes.data,
                                                        pts.ctypes.data, len(pts), edges.ctypes.data, len(edges), C.addressof(links), len(link4),
                                                        cam5.ctypes.data, iterations, lambda_init, stop_ptr, chi2.ctypes.data, dpos.ctypes.data,
                                                        C.addressof(stats), C.c_void_p(stream)))
    return kf, pts, chi2[:len(edges)], dpos[:len(edges)], stats


class PoseImuPrior(C.Structure):
    """tc2li_pose_imu_prior"""
    _fields_ = [("Rwb", C.c_double * 9), ("twb", C.c_double * 3), ("vwb", C.c_double * 3), ("bg", C.c_double * 3), ("ba", C.c_double * 3),
                ("H", C.c_double * 225)]


class PoseInertialProblem(C.Structure):
    """tc2li_pose_inertial_problem"""
    _fields_ = [("frame", C.c_double * 33), ("other", C.c_double * 33), ("prior", C.c_void_p), ("preintegrated", C.c_void_p),
                ("preintegrated_rw", C.c_void_p), ("Xw", C.c_void_p), ("edges", C.c_void_p), ("close_point", C.c_void_p), ("outlier", C.c_void_p),
                ("prior_out", C.c_void_p), ("n_edges", C.c_int32), ("last_frame", C.c_int32), ("rec_init", C.c_int32), ("n_initial", C.c_int32),
                ("n_bad", C.c_int32), ("n_inliers", C.c_int32), ("solver_failed", C.c_int32), ("pad_", C.c_int32)]


def pose_inertial_optimization_batch(problems, calib24, cam5, stream=0):
    """``tc2li_pose_inertial_optimization_batch``.  problems: dicts with cur33, other33 (kf33 layout), last_frame, prior246 (or None),
    pre / pre_rw (``Preintegrated``), Xw [E, 3], edges (BA_EDGE_DTYPE), close [E], rec_init
    -> per frame (cur33, other33, outlier, prior246, return value, (n_initial, n_bad, n_inliers), solver_failed)."""
    n = len(problems)
    arr = (PoseInertialProblem * max(n, 1))()
    keep = []
    for f, pr in enumerate(problems):
        Xw = np.ascontiguousarray(pr["Xw"], np.float64).reshape(-1, 3)
        e = np.ascontiguousarray(pr["edges"], BA_EDGE_DTYPE)
        cl = np.ascontiguousarray(pr["close"], np.uint8)
        out = np.zeros(max(len(e), 1), np.uint8)
        prior_in, prior_out = PoseImuPrior(), PoseImuPrior()
        if pr.get("prior246") is not None:
            C.memmove(C.addressof(prior_in), np.ascontiguousarray(pr["prior246"], np.float64).ctypes.data, 246 * 8)
        keep.append((Xw, e, cl, out, prior_in, prior_out))
        a = arr[f]
        C.memmove(a.frame, np.ascontiguousarray(pr["cur33"], np.float64).ctypes.data, 33 * 8)
        C.memmove(a.other, np.ascontiguousarray(pr["other33"], np.float64).ctypes.data, 33 * 8)
        a.prior = C.addressof(prior_in) if pr.get("prior246") is not None else None
        a.preintegrated, a.preintegrated_rw = C.addressof(pr["pre"].p), C.addressof(pr.get("pre_rw", pr["pre"]).p)
        a.Xw, a.edges, a.close_point, a.outlier, a.prior_out = Xw.ctypes.data, e.ctypes.data, cl.ctypes.data, out.ctypes.data, C.addressof(prior_out)
        a.n_edges, a.last_frame, a.rec_init = len(e), int(bool(pr.get("last_frame"))), int(bool(pr.get("rec_init")))
    res = np.zeros(max(n, 1), np.int32)
    f_ = lib().tc2li_pose_inertial_optimization_batch
    f_.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f_(C.addressof(arr), n, np.ascontiguousarray(calib24, np.float64).ctypes.data, np.ascontiguousarray(cam5, np.float64).ctypes.data, res.ctypes.data,
              C.c_void_p(stream)))
    outs = []
    for f in range(n):
        a, (Xw, e, cl, out, prior_in, prior_out) = arr[f], keep[f]
        outs.append((np.array(a.frame), np.array(a.other), out[:len(e)].copy(), np.frombuffer(bytes(prior_out), np.float64).copy(), int(res[f]),
                     (a.n_initial, a.n_bad, a.n_inliers), bool(a.solver_failed)))
    return outs


class PoseInertialBatch:
    """The IMU pre-integration between two frames and ``PoseInertialOptimizationLastFrame / LastKeyFrame`` for one frame of each of n sequences,
    packed once: ``preintegrate()`` = tc2li_imu_preintegrate_frames, ``run()`` = tc2li_pose_inertial_optimization_batch from the initial states.
    problems: dicts as for pose_inertial_optimization_batch plus samples (IMU_SAMPLE_DTYPE), t1, t2, bias6; noise4 = ng, na, ngw, naw."""

    def __init__(self, problems, calib24, cam5, noise4):
        n = self.n = len(problems)
        self.calib24, self.cam5, self.noise4 = np.ascontiguousarray(calib24, np.float64), np.ascontiguousarray(cam5, np.float64), [float(v) for v in noise4]
        self.arr = (PoseInertialProblem * n)()
        self.pre = (PreintegratedPOD * n)()
        self.bias = (ImuBias * n)()
        self.samples = np.concatenate([np.ascontiguousarray(p["samples"], IMU_SAMPLE_DTYPE) for p in problems])
        self.offsets = np.concatenate([[0], np.cumsum([len(p["samples"]) for p in problems])]).astype(np.int32)
        self.t1, self.t2 = np.array([p["t1"] for p in problems], np.float64), np.array([p["t2"] for p in problems], np.float64)
        self.keep, self.init = [], []
        self.results = np.zeros(n, np.int32)
        for f, pr in enumerate(problems):
            Xw = np.ascontiguousarray(pr["Xw"], np.float64).reshape(-1, 3)
            e = np.ascontiguousarray(pr["edges"], BA_EDGE_DTYPE)
            cl = np.ascontiguousarray(pr["close"], np.uint8)
            out = np.zeros(max(len(e), 1), np.uint8)
            prior_in, prior_out = PoseImuPrior(), PoseImuPrior()
            if pr.get("prior246") is not None:
                C.memmove(C.addressof(prior_in), np.ascontiguousarray(pr["prior246"], np.float64).ctypes.data, 246 * 8)
            self.bias[f] = ImuBias(*[float(x) for x in pr["bias6"]])
            cur, oth = np.ascontiguousarray(pr["cur33"], np.float64).copy(), np.ascontiguousarray(pr["other33"], np.float64).copy()
            self.keep.append((Xw, e, cl, out, prior_in, prior_out))
            self.init.append((cur, oth))
            a = self.arr[f]
            a.prior = C.addressof(prior_in) if pr.get("prior246") is not None else None
            a.preintegrated = a.preintegrated_rw = C.addressof(self.pre) + f * C.sizeof(PreintegratedPOD)
            a.Xw, a.edges, a.close_point, a.outlier, a.prior_out = Xw.ctypes.data, e.ctypes.data, cl.ctypes.data, out.ctypes.data, C.addressof(prior_out)
            a.n_edges, a.last_frame, a.rec_init = len(e), int(bool(pr.get("last_frame"))), int(bool(pr.get("rec_init")))
        self.init33 = np.stack([np.concatenate(x) for x in self.init])  # [n, 66]

    def preintegrate(self):
        f = lib().tc2li_imu_preintegrate_frames
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(f(self.n, C.addressof(self.pre), C.addressof(self.bias), *self.noise4, self.samples.ctypes.data, self.offsets.ctypes.data,
                        self.t1.ctypes.data, self.t2.ctypes.data))

    def run(self, stream=0):
        """Resets every frame to its initial state and optimises all of them -> results [n] (initial correspondences - bad)."""
        frame_off, stride = PoseInertialProblem.frame.offset, C.sizeof(PoseInertialProblem)
        base = C.addressof(self.arr)
        # frame | other are adjacent: every record's 66 doubles in one strided assignment (a loop of memmoves was interpreter time on the stage thread)
        np.frombuffer(self.arr, np.uint8).reshape(self.n, stride)[:, frame_off:frame_off + 66 * 8] = np.ascontiguousarray(self.init33).view(np.uint8).reshape(self.n, 66 * 8)
        f_ = lib().tc2li_pose_inertial_optimization_batch
        f_.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f_(base, self.n, self.calib24.ctypes.data, self.cam5.ctypes.data, self.results.ctypes.data, C.c_void_p(stream)))
        return self.results

    def inliers(self):
        return np.array([self.arr[f].n_inliers for f in range(self.n)])


class LidarInertialBatch:
    """``tc2li_lidar_inertial_frontend_batch`` with its per-scan records packed once: ``run()`` starts every sequence from its initial filter
    state and covariance again (a benchmark's repeated step).  Arguments as LidarFrontEnd.inertial_frontend_batch."""

    def __init__(self, fe, raw_offsets, maps, states36, Ps, imus, times, cov12, R=0.001, max_iter=3, limit=None, extrinsic_est_en=False,
                 point_filter_num=2, blind=2.0, time_unit_scale=1e-3, leaf=0.5):
        S = self.S = len(raw_offsets) - 1
        self.fe = fe
        self.raw_offsets = np.ascontiguousarray(raw_offsets, np.int32)
        self.st0 = np.ascontiguousarray(states36, np.float64).reshape(S, 36).copy()
        self.P0 = np.ascontiguousarray(Ps, np.float64).reshape(S, 529).copy()
        self.P = self.P0.copy()
        self.cov, self.lim = np.ascontiguousarray(cov12, np.float64), np.ascontiguousarray(np.full(23, 0.001) if limit is None else limit, np.float64)
        self.imus = [np.ascontiguousarray(x, np.float64).reshape(-1, 7) for x in imus]
        times = np.ascontiguousarray(times, np.float64).reshape(S, 4)
        self.scans = (LidarInertialScan * S)()
        for s in range(S):
            sc = self.scans[s]
            sc.imu, sc.n_imu = self.imus[s].ctypes.data, len(self.imus[s])
            sc.pcl_beg_time, sc.pcl_end_time, sc.last_lidar_end_time, sc.acc_scale = [float(v) for v in times[s]]
            sc.P = self.P[s].ctypes.data
        self.handles = (C.c_void_p * S)(*[m._h for m in maps])
        self.args = (point_filter_num, blind, time_unit_scale, leaf)
        self.tail = (R, max_iter, int(extrinsic_est_en))
        self.state_off, self.stride = LidarInertialScan.state.offset, C.sizeof(LidarInertialScan)
        self.last_off = LidarInertialScan.acc_s_last.offset
        self.zero6 = np.zeros(6)

    def prepare(self, dev_raw_ptr, stream=0, fe=None):
        """Preprocess + time-sort order of the scans into the handle ``fe`` (default: the batch's own), ahead of ``run(None, ..., fe=fe)``."""
        point_filter_num, blind, time_unit_scale, _ = self.args
        return (fe or self.fe).inertial_prepare_batch(dev_raw_ptr, self.raw_offsets, point_filter_num, blind, time_unit_scale, stream)

    def run(self, dev_raw_ptr, stream=0, fe=None):
        """dev_raw_ptr None: the scans ``prepare`` left in the handle."""
        self.P[...] = self.P0
        raw = np.frombuffer(self.scans, np.uint8).reshape(self.S, self.stride)   # the records' bytes: states and last accelerations in two strided assignments
        raw[:, self.state_off:self.state_off + 36 * 8] = self.st0.view(np.uint8).reshape(self.S, 36 * 8)
        raw[:, self.last_off:self.last_off + 48] = 0
        f = lib().tc2li_lidar_inertial_frontend_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        R, max_iter, ext = self.tail
        return _check(f((fe or self.fe)._h, self.S, C.c_void_p(dev_raw_ptr or 0), self.raw_offsets.ctypes.data, *self.args, self.handles, self.scans,
                        self.cov.ctypes.data, R, max_iter, self.lim.ctypes.data, ext, C.c_void_p(stream)))

    def states36(self):
        raw = np.frombuffer(self.scans, np.uint8).reshape(self.S, self.stride)
        return np.ascontiguousarray(raw[:, self.state_off:self.state_off + 36 * 8]).view(np.float64).reshape(self.S, 36)

    def stats(self):
        return [(sc.stats.calls, sc.stats.searches, sc.stats.effct_feat_num, sc.n_preprocessed, sc.n_downsampled) for sc in self.scans]


class LastFrame(C.Structure):
    """tc2li_last_frame: what SearchByProjection(F, LastFrame) reads of mLastFrame."""
    _fields_ = [("n", C.c_int32), ("pad_", C.c_int32), ("has_point", C.c_void_p), ("outlier", C.c_void_p), ("Xw", C.c_void_p),
                ("keys", C.c_void_p), ("descriptors", C.c_void_p), ("pose7", C.c_float * 7), ("pad2_", C.c_float)]


def pack_last_frames(items):
    """items: list of dicts with has_point, outlier, Xw, keys, descriptors, pose7 -> (ctypes array, keep-alive list)."""
    arr = (LastFrame * len(items))()
    keep = []
    for i, it in enumerate(items):
        hp = np.ascontiguousarray(it["has_point"], np.uint8); ol = np.ascontiguousarray(it["outlier"], np.uint8)
        Xw = np.ascontiguousarray(it["Xw"], np.float32); keys = np.ascontiguousarray(it["keys"], KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(it["descriptors"], np.uint8)
        keep.append((hp, ol, Xw, keys, desc))
        arr[i] = LastFrame(len(keys), 0, hp.ctypes.data, ol.ctypes.data, Xw.ctypes.data, keys.ctypes.data, desc.ctypes.data,
                           (C.c_float * 7)(*[float(x) for x in it["pose7"]]), 0.0)
    return arr, keep


def track_motion_model_batch(ext, n_frames, keypoints, u_right, last_frames, pose_pred7, cam5, b, th=7.0, stream=0, out=None):
    """``Tracking::TrackWithMotionModel`` data path for the frames of the preceding ``extract_batch_dev`` /
    ``stereo_match_batch`` calls -> (poses7 [F, 7], map_point_of_keypoint [F, capacity], n_matches [F], n_inliers [F]).
    last_frames: result of ``pack_last_frames``."""
    keypoints = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE)
    u_right = np.ascontiguousarray(u_right, np.float32)
    pp = np.ascontiguousarray(pose_pred7, np.float32)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    cap = keypoints.shape[1]
    if out is None:
        out = (np.zeros((n_frames, 7)), np.full((n_frames, cap), -1, np.int32), np.zeros(n_frames, np.int32), np.zeros(n_frames, np.int32))
    poses, mp, nm, inl = out
    arr = last_frames[0] if isinstance(last_frames, tuple) else last_frames
    _check(lib().tc2li_track_motion_model_batch(ext._h, n_frames, keypoints.ctypes.data, u_right.ctypes.data, cap, C.addressof(arr),
                                                pp.ctypes.data, cam5.ctypes.data, b, th, poses.ctypes.data, mp.ctypes.data,
                                                nm.ctypes.data, inl.ctypes.data, C.c_void_p(stream)))
    return poses, mp, nm, inl


class LidarWindow(C.Structure):
    """tc2li_lidar_window: the co-visibility window of LocalLVBundleAdjustment (SF/src/OptimizerWithLidar.cc:226-260)."""
    _fields_ = [("n_keyframes", C.c_int32), ("pad_", C.c_int32), ("pose_index", C.c_void_p), ("cloud_xyz", C.c_void_p),
                ("cloud_offsets", C.c_void_p), ("Tcl", C.c_float * 7), ("pad2_", C.c_float), ("weight", C.c_double)]


class LidarBaStats(C.Structure):
    _fields_ = [("n_planes", C.c_int32), ("hessian_evaluations", C.c_int32), ("residual", C.c_double), ("chi2", C.c_double)]


def _pack_lidar_window(win_pose, clouds, Tcl7, weight):
    win = np.ascontiguousarray(win_pose, np.int32)
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32)
    cl = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32).reshape(-1, 3) for c in clouds]), np.float32)
    w = LidarWindow(len(win), 0, win.ctypes.data, cl.ctypes.data, off.ctypes.data, (C.c_float * 7)(*[float(x) for x in Tcl7]), 0.0,
                    float(weight))
    return w, (win, off, cl)  # keep the arrays alive


def local_lv_bundle_adjustment(poses7, fixed, points3, edges, cam5, win_pose, clouds, Tcl7, weight, iterations=10, lambda_init=0.0,
                               stop_flag=None, stream=0):
    """``OptimizerWithLidar::LocalLVBundleAdjustment`` with the LiDAR edge over the keyframes ``win_pose`` (rows of
    poses7) and their surface clouds -> (poses7, points3, chi2, depth_positive, stats, lidar_stats)."""
    poses = np.ascontiguousarray(poses7, np.float64).copy()
    pts = np.ascontiguousarray(points3, np.float64).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    chi2 = np.zeros(max(len(edges), 1))
    dpos = np.zeros(max(len(edges), 1), np.uint8)
    stats, lstats = BaStats(), LidarBaStats()
    stop_ptr = stop_flag.ctypes.data if stop_flag is not None else None
    w, keep = _pack_lidar_window(win_pose, clouds, Tcl7, weight)
    _check(lib().tc2li_local_lv_bundle_adjustment(poses.ctypes.data, fixed.ctypes.data, len(poses), pts.ctypes.data, len(pts),
                                                  edges.ctypes.data, len(edges), cam5.ctypes.data, iterations, lambda_init, stop_ptr,
                                                  chi2.ctypes.data, dpos.ctypes.data, C.addressof(stats), C.addressof(w),
                                                  C.addressof(lstats), C.c_void_p(stream)))
    del keep
    return poses, pts, chi2[:len(edges)], dpos[:len(edges)], stats, lstats


REDUCE_SUM, REDUCE_MAX = 0, 1
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class BaShard(C.Structure):
    """tc2li_ba_shard"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("allreduce", C.c_void_p), ("ctx", C.c_void_p)]


class RcclComm:
    """An RCCL communicator made by the library (``tc2li_rccl_comm_create``): rank 0 calls ``RcclComm.unique_id()`` and hands
    the 128 bytes to the other ranks (bench.py broadcasts them over torch.distributed)."""

    def __init__(self, unique_id, rank, world):
        uid = np.ascontiguousarray(np.frombuffer(bytes(unique_id), np.uint8))
        assert uid.size == 128
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        _check(lib().tc2li_rccl_comm_create(uid.ctypes.data, rank, world, C.byref(self.h)))

    @staticmethod
    def unique_id():
        uid = np.zeros(128, np.uint8)
        _check(lib().tc2li_rccl_unique_id(uid.ctypes.data))
        return uid.tobytes()

    def shard(self):
        """tc2li_ba_shard whose all-reduce is ``tc2li_rccl_allreduce`` on this communicator (no Python on the data path)."""
        fn = C.cast(lib().tc2li_rccl_allreduce, C.c_void_p).value
        return BaShard(self.rank, self.world, fn, self.h.value)

    def close(self):
        if self.h:
            lib().tc2li_rccl_comm_destroy(self.h)
            self.h = C.c_void_p()


def torch_allreduce_shard(rank, world, group=None):
    """tc2li_ba_shard whose all-reduce is ``torch.distributed.all_reduce`` (any backend that takes device tensors) -> (shard,
    keep-alive).  The library hands the callback a device pointer; the tensor that aliases it goes through
    ``__cuda_array_interface__``."""
    import torch
    import torch.distributed as dist

    class _Alias:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def cb(ctx, ptr, count, op, stream):
        try:
            ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
            rop = dist.ReduceOp.SUM if op == REDUCE_SUM else dist.ReduceOp.MAX
            with torch.cuda.stream(ext):
                t = torch.as_tensor(_Alias(ptr, count), device="cuda")
                if dist.get_backend(group) == "gloo":  # host collective: stage through host memory, in stream order
                    h = t.cpu()
                    dist.all_reduce(h, op=rop, group=group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=rop, group=group)
            return 0
        except Exception as e:  # the C side turns a non-zero return into TC2LI_ERR_COMM
            print("all-reduce callback failed:", repr(e))
            return 1

    fn = ALLREDUCE_FN(cb)
    return BaShard(rank, world, C.cast(fn, C.c_void_p).value, None), fn


def ba_shard_select(edges, n_points, rank, world):
    """``tc2li_ba_shard_select`` -> (landmark_owned, edge_owned) masks of the rank."""
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    lo, eo = np.zeros(max(n_points, 1), np.uint8), np.zeros(max(len(edges), 1), np.uint8)
    _check(lib().tc2li_ba_shard_select(edges.ctypes.data, len(edges), n_points, rank, world, lo.ctypes.data, eo.ctypes.data))
    return lo[:n_points].astype(bool), eo[:len(edges)].astype(bool)


def local_lv_bundle_adjustment_sharded(shard, poses7, fixed, points3, edges, cam5, win_pose=None, clouds=None, Tcl7=None, weight=1.0,
                                       iterations=10, lambda_init=0.0, stop_flag=None, stream=0):
    """One window split over the ranks of ``shard`` (landmark partition + all-reduce of the shared-pose blocks); every rank
    passes the whole window and receives the whole result -> (poses7, points3, chi2, depth_positive, stats, lidar_stats)."""
    poses = np.ascontiguousarray(poses7, np.float64).copy()
    pts = np.ascontiguousarray(points3, np.float64).copy()
    fixed = np.ascontiguousarray(fixed, np.uint8)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    cam5 = np.ascontiguousarray(cam5, np.float64)
    chi2 = np.zeros(max(len(edges), 1))
    dpos = np.zeros(max(len(edges), 1), np.uint8)
    stats, lstats = BaStats(), LidarBaStats()
    stop_ptr = stop_flag.ctypes.data if stop_flag is not None else None
    w, keep = (None, None) if win_pose is None else _pack_lidar_window(win_pose, clouds, Tcl7, weight)
    _check(lib().tc2li_local_lv_bundle_adjustment_sharded(poses.ctypes.data, fixed.ctypes.data, len(poses), pts.ctypes.data, len(pts),
                                                          edges.ctypes.data, len(edges), cam5.ctypes.data, iterations, lambda_init, stop_ptr,
                                                          chi2.ctypes.data, dpos.ctypes.data, C.addressof(stats),
                                                          C.addressof(w) if w is not None else None, C.addressof(lstats),
                                                          C.addressof(shard), C.c_void_p(stream)))
    del keep
    return poses, pts, chi2[:len(edges)], dpos[:len(edges)], stats, lstats


class BaProblem(C.Structure):
    """tc2li_ba_problem"""
    _fields_ = [("poses7", C.c_void_p), ("fixed", C.c_void_p), ("points3", C.c_void_p), ("edges", C.c_void_p), ("n_poses", C.c_int32),
                ("n_points", C.c_int32), ("n_edges", C.c_int32), ("iterations", C.c_int32), ("lambda_init", C.c_double),
                ("stop_flag", C.c_void_p), ("edge_chi2", C.c_void_p), ("edge_depth_positive", C.c_void_p), ("stats", C.c_void_p),
                ("lidar", C.c_void_p), ("lidar_stats", C.c_void_p)]


class BaBatch:
    """A set of independent local-BA windows prepared once (arrays pinned down for the C side) and optimised together by
    ``tc2li_local_bundle_adjustment_batch``.  windows: dicts with poses, fixed, points, edges (BA_EDGE_DTYPE) and optionally
    win_pose, clouds, Tcl7, weight; iterations / lambda_init per window optional."""

    def __init__(self, windows, cam5):
        self.n = len(windows)
        self.cam5 = np.ascontiguousarray(cam5, np.float64)
        self.arr = (BaProblem * self.n)()
        self.init = []
        self.keep = []
        self.stats = (BaStats * self.n)()
        self.lstats = (LidarBaStats * self.n)()
        self.results = np.zeros(self.n, np.int32)
        for i, w in enumerate(windows):
            poses0 = np.ascontiguousarray(w["poses"], np.float64); pts0 = np.ascontiguousarray(w["points"], np.float64)
            poses, pts = poses0.copy(), pts0.copy()
            fixed = np.ascontiguousarray(w["fixed"], np.uint8)
            edges = np.ascontiguousarray(w["edges"], BA_EDGE_DTYPE)
            chi2, dpos = np.zeros(max(len(edges), 1)), np.zeros(max(len(edges), 1), np.uint8)
            lw = None
            if w.get("win_pose") is not None:
                lw = _pack_lidar_window(w["win_pose"], w["clouds"], w["Tcl7"], w.get("weight", 1.0))
            stop = w.get("stop_flag")   # (a uint8 array of one element the caller may set while the window is optimised: *pbStopFlag)
            self.init.append((poses0, pts0))
            self.keep.append((poses, pts, fixed, edges, chi2, dpos, lw, stop))
            self.arr[i] = BaProblem(poses.ctypes.data, fixed.ctypes.data, pts.ctypes.data, edges.ctypes.data, len(poses), len(pts), len(edges),
                                    int(w.get("iterations", 10)), float(w.get("lambda_init", 0.0)), stop.ctypes.data if stop is not None else None,
                                    chi2.ctypes.data, dpos.ctypes.data,
                                    C.addressof(self.stats) + i * C.sizeof(BaStats), C.addressof(lw[0]) if lw else None,
                                    C.addressof(self.lstats) + i * C.sizeof(LidarBaStats))

    def run(self, max_concurrency=8):
        """Resets every window to its initial estimate and optimises all of them -> number of successful windows."""
        for (p0, x0), k in zip(self.init, self.keep):
            k[0][...] = p0
            k[1][...] = x0
        return _check(lib().tc2li_local_bundle_adjustment_batch(C.addressof(self.arr), self.n, self.cam5.ctypes.data, max_concurrency,
                                                                self.results.ctypes.data))

    def run_group(self, group):
        """The same as ONE lock-step group on the library's context `group` (tc2li_local_bundle_adjustment_batch_group): for callers with
        several mapping workers, each on a group of its own."""
        for (p0, x0), k in zip(self.init, self.keep):
            k[0][...] = p0
            k[1][...] = x0
        return _check(lib().tc2li_local_bundle_adjustment_batch_group(C.addressof(self.arr), self.n, self.cam5.ctypes.data, int(group),
                                                                      self.results.ctypes.data))

    def reset(self):
        for (p0, x0), k in zip(self.init, self.keep):
            k[0][...] = p0
            k[1][...] = x0

    def result(self, i):
        k = self.keep[i]
        return k[0], k[1], k[4][:len(k[3])], k[5][:len(k[3])], self.stats[i], self.lstats[i]


class BaEngine:
    """``tc2li_ba_engine``: a running lock-step queue with up to ``max_windows`` local-BA windows in flight; windows join and leave it
    one by one.  submit(batch) -> ticket (the BaBatch's arrays must stay alive until wait(ticket))."""

    def __init__(self, cam5, max_windows=192):
        self.cam5 = np.ascontiguousarray(cam5, np.float64)
        self.h = C.c_void_p()
        _check(lib().tc2li_ba_engine_create(self.cam5.ctypes.data, int(max_windows), C.byref(self.h)))

    def submit(self, batch, first=0, count=None):
        """Resets the windows [first, first + count) of the BaBatch to their initial estimates and hands them to the engine."""
        count = batch.n - first if count is None else count
        for (p0, x0), k in zip(batch.init[first:first + count], batch.keep[first:first + count]):
            k[0][...] = p0
            k[1][...] = x0
        t = lib().tc2li_ba_engine_submit(self.h, C.addressof(batch.arr) + first * C.sizeof(BaProblem), count,
                                         batch.results.ctypes.data + 4 * first)
        if t < 0:
            _check(int(t))
        return t

    def wait(self, ticket):
        return _check(lib().tc2li_ba_engine_wait(self.h, ticket))

    def done(self, ticket):
        """True when wait(ticket) will not block."""
        return _check(lib().tc2li_ba_engine_poll(self.h, ticket)) == 1

    def close(self):
        if self.h:
            lib().tc2li_ba_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lidar_planes_host(poses7, win_pose, clouds, Tcl7, capacity=20000):
    """Host-only plane extraction of the LiDAR window -> (clusters [n, W, 10] = P00 P01 P02 P11 P12 P22 v N, coe [n])."""
    poses = np.ascontiguousarray(poses7, np.float64)
    w, keep = _pack_lidar_window(win_pose, clouds, Tcl7, 1.0)
    out, coe = np.zeros((capacity, w.n_keyframes, 10)), np.zeros(capacity)
    n = _check(lib().tc2li_host_lidar_planes(poses.ctypes.data, len(poses), C.addressof(w), out.ctypes.data, coe.ctypes.data, capacity))
    del keep
    return out[:n].copy(), coe[:n].copy()


def lidar_planes_device(poses7, win_pose, clouds, Tcl7, capacity=2048):
    """The same planes from the extraction kernels of the batched local BA (``tc2li_device_lidar_planes``) ->
    (clusters [n, W, 10], coe [n], info [4] = planes, declined, root voxels, planes found)."""
    poses = np.ascontiguousarray(poses7, np.float64)
    w, keep = _pack_lidar_window(win_pose, clouds, Tcl7, 1.0)
    out, coe, info = np.zeros((capacity, w.n_keyframes, 10)), np.zeros(capacity), np.zeros(4, np.int32)
    n = _check(lib().tc2li_device_lidar_planes(poses.ctypes.data, len(poses), C.addressof(w), out.ctypes.data, coe.ctypes.data, capacity, info.ctypes.data))
    del keep
    return out[:n].copy(), coe[:n].copy(), info


def local_lvi_bundle_adjustment(kf33, fixed, has_imu, calib24, points3, edges, link4, preintegrated, cam5, win_kf, clouds, Tcl7, Tbl7, weight,
                                iterations=10, lambda_init=1.0, stop_flag=None, stream=0):
    """``OptimizerWithLidar::LocalLVIBA``: local_inertial_bundle_adjustment plus the LiDAR edge over the keyframes ``win_kf``
    (rows of kf33) -> (kf33, points3, chi2, depth_positive, stats, lidar_stats)."""
    kf = np.ascontiguousarray(kf33, np.float64).copy()
    pts = np.ascontiguousarray(points3, np.float64).copy()
    fixed, has_imu = np.ascontiguousarray(fixed, np.uint8), np.ascontiguousarray(has_imu, np.uint8)
    calib24, cam5 = np.ascontiguousarray(calib24, np.float64), np.ascontiguousarray(cam5, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    link4 = np.ascontiguousarray(link4, np.float64).reshape(-1, 4)
    links = (InertialLink * max(len(link4), 1))()
    for l, row in enumerate(link4):
        links[l] = InertialLink(int(row[0]), int(row[1]), int(row[2] != 0), 0, float(row[3]), C.addressof(preintegrated[l].p))
    chi2 = np.zeros(max(len(edges), 1))
    dpos = np.zeros(max(len(edges), 1), np.uint8)
    stats, lstats = BaStats(), LidarBaStats()
    stop_ptr = stop_flag.ctypes.data if stop_flag is not None else None
    w, keep = _pack_lidar_window(win_kf, clouds, Tcl7, weight)
    tbl = np.ascontiguousarray(Tbl7, np.float32)
    f = lib().tc2li_local_lvi_bundle_adjustment
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                  C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(kf.ctypes.data, fixed.ctypes.data, has_imu.ctypes.data, len(kf), calib24.ctypes.data, pts.ctypes.data, len(pts), edges.ctypes.data,
             len(edges), C.addressof(links), len(link4), cam5.ctypes.data, iterations, lambda_init, stop_ptr, chi2.ctypes.data, dpos.ctypes.data,
             C.addressof(stats), C.addressof(w), tbl.ctypes.data, C.addressof(lstats), C.c_void_p(stream)))
    del keep
    return kf, pts, chi2[:len(edges)], dpos[:len(edges)], stats, lstats


class LviProblem(C.Structure):
    """tc2li_lvi_problem"""
    _fields_ = [("keyframes", C.c_void_p), ("fixed", C.c_void_p), ("has_imu", C.c_void_p), ("points3", C.c_void_p), ("edges", C.c_void_p), ("links", C.c_void_p),
                ("n_keyframes", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("n_links", C.c_int32), ("iterations", C.c_int32), ("pad_", C.c_int32),
                ("lambda_init", C.c_double), ("stop_flag", C.c_void_p), ("edge_chi2", C.c_void_p), ("edge_depth_positive", C.c_void_p), ("stats", C.c_void_p),
                ("lidar", C.c_void_p), ("Tbl", C.c_void_p), ("lidar_stats", C.c_void_p)]


class LviBatch:
    """A set of independent ``LocalLVIBA`` windows prepared once and optimised together by ``tc2li_local_lvi_bundle_adjustment_batch``.
    windows: dicts with kf33, fixed, has_imu, points, edges (BA_EDGE_DTYPE), link4, pre (list of Preintegrated) and optionally win_kf, clouds,
    Tcl7, Tbl7, weight; iterations / lambda_init per window optional (10, 1.0)."""

    def __init__(self, windows, calib24, cam5):
        self.n = len(windows)
        self.calib24, self.cam5 = np.ascontiguousarray(calib24, np.float64), np.ascontiguousarray(cam5, np.float64)
        self.arr = (LviProblem * self.n)()
        self.init, self.keep = [], []
        self.stats = (BaStats * self.n)()
        self.lstats = (LidarBaStats * self.n)()
        self.results = np.zeros(self.n, np.int32)
        for i, w in enumerate(windows):
            kf0, pts0 = np.ascontiguousarray(w["kf33"], np.float64), np.ascontiguousarray(w["points"], np.float64)
            kf, pts = kf0.copy(), pts0.copy()
            fixed, has_imu = np.ascontiguousarray(w["fixed"], np.uint8), np.ascontiguousarray(w["has_imu"], np.uint8)
            edges = np.ascontiguousarray(w["edges"], BA_EDGE_DTYPE)
            link4 = np.ascontiguousarray(w["link4"], np.float64).reshape(-1, 4)
            links = (InertialLink * max(len(link4), 1))()
            for l, row in enumerate(link4):
                links[l] = InertialLink(int(row[0]), int(row[1]), int(row[2] != 0), 0, float(row[3]), C.addressof(w["pre"][l].p))
            chi2, dpos = np.zeros(max(len(edges), 1)), np.zeros(max(len(edges), 1), np.uint8)
            lw, tbl = None, None
            if w.get("win_kf") is not None:
                lw = _pack_lidar_window(w["win_kf"], w["clouds"], w["Tcl7"], w.get("weight", 1.0))
                tbl = np.ascontiguousarray(w["Tbl7"], np.float32)
            self.init.append((kf0, pts0))
            self.keep.append((kf, pts, fixed, has_imu, edges, links, chi2, dpos, lw, tbl, w["pre"]))
            self.arr[i] = LviProblem(kf.ctypes.data, fixed.ctypes.data, has_imu.ctypes.data, pts.ctypes.data, edges.ctypes.data, C.addressof(links),
                                     len(kf), len(pts), len(edges), len(link4), int(w.get("iterations", 10)), 0, float(w.get("lambda_init", 1.0)), None,
                                     chi2.ctypes.data, dpos.ctypes.data, C.addressof(self.stats) + i * C.sizeof(BaStats),
                                     C.addressof(lw[0]) if lw else None, tbl.ctypes.data if lw else None,
                                     C.addressof(self.lstats) + i * C.sizeof(LidarBaStats))

    def run(self, max_concurrency=8):
        """Resets every window to its initial estimate and optimises all of them -> number of successful windows."""
        for (k0, x0), k in zip(self.init, self.keep):
            k[0][...] = k0
            k[1][...] = x0
        f = lib().tc2li_local_lvi_bundle_adjustment_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        return _check(f(C.addressof(self.arr), self.n, self.calib24.ctypes.data, self.cam5.ctypes.data, max_concurrency, self.results.ctypes.data))

    def run_group(self, group):
        """The same as ONE lock-step group on the library's context `group` (tc2li_local_lvi_bundle_adjustment_batch_group): for callers with several
        mapping workers, each on a group of its own."""
        for (k0, x0), k in zip(self.init, self.keep):
            k[0][...] = k0
            k[1][...] = x0
        f = lib().tc2li_local_lvi_bundle_adjustment_batch_group
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        return _check(f(C.addressof(self.arr), self.n, self.calib24.ctypes.data, self.cam5.ctypes.data, int(group), self.results.ctypes.data))

    def result(self, i):
        k = self.keep[i]
        return k[0], k[1], k[6][:len(k[4])], k[7][:len(k[4])], self.stats[i], self.lstats[i]


def lidar_window_evaluate(poses7, win_pose, clouds, Tcl7, derivatives=True):
    """The LiDAR edge alone -> (n_planes, residual, JacT [6W], Hessian [6W, 6W]) (ComputeError / ComputeJandHSE3)."""
    poses = np.ascontiguousarray(poses7, np.float64)
    w, keep = _pack_lidar_window(win_pose, clouds, Tcl7, 1.0)
    W = w.n_keyframes
    res = C.c_double(0)
    J, H = np.zeros(6 * W), np.zeros((6 * W, 6 * W))
    n = _check(lib().tc2li_lidar_window_evaluate(poses.ctypes.data, len(poses), C.addressof(w), C.addressof(res),
                                                 J.ctypes.data if derivatives else None, H.ctypes.data if derivatives else None, None))
    del keep
    return n, res.value, J, H


# ---- projection matching (ORBmatcher::SearchByProjection) ----------------------------------------------------------
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("u_right", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                        ("angle", "<f4"), ("valid", "<i2"), ("has_observations", "<i2"), ("descriptor", "u1", (32,))])
MAP_POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"), ("max_distance", "<f4"),
                            ("max_distance_raw", "<f4"), ("descriptor", "u1", (32,))])


class FrameView(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("descriptors", C.c_void_p), ("u_right", C.c_void_p), ("occupied", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


def search_by_projection(keys, desc, u_right, occupied, cols, rows, queries, mode, nn_ratio=0.9, check_orientation=False):
    """The matching loops of both tracking overloads -> (nmatches, match_of_query, query_of_keypoint)."""
    keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    ur = np.ascontiguousarray(u_right, np.float32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    fv = FrameView(keys.ctypes.data, desc.ctypes.data, ur.ctypes.data, None if occ is None else occ.ctypes.data, len(keys), 0.0,
                   float(cols), 0.0, float(rows))
    match = np.full(max(len(queries), 1), -1, np.int32)
    qok = np.full(max(len(keys), 1), -1, np.int32)
    n = _check(lib().tc2li_search_by_projection(C.byref(fv), queries.ctypes.data, len(queries), mode, nn_ratio, int(check_orientation),
                                                match.ctypes.data, qok.ctypes.data))
    return n, match[:len(queries)], qok[:len(keys)]


def project_last_frame(pose_cur7, pose_last7, cam4, b, bf, scales, cols, rows, has_point, outlier, Xw, last_keys, mp_desc, th, mono=False):
    pc, pl, cam4 = [np.ascontiguousarray(a, np.float32) for a in (pose_cur7, pose_last7, cam4)]
    scales = np.ascontiguousarray(scales, np.float32)
    hp, ol = np.ascontiguousarray(has_point, np.uint8), np.ascontiguousarray(outlier, np.uint8)
    Xw = np.ascontiguousarray(Xw, np.float32)
    last_keys = np.ascontiguousarray(last_keys, KEYPOINT_DTYPE)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    out = np.zeros(max(len(last_keys), 1), QUERY_DTYPE)
    _check(lib().tc2li_project_last_frame(pc.ctypes.data, pl.ctypes.data, cam4.ctypes.data, b, bf, scales.ctypes.data, len(scales), cols,
                                          rows, len(last_keys), hp.ctypes.data, ol.ctypes.data, Xw.ctypes.data, last_keys.ctypes.data,
                                          mp_desc.ctypes.data, th, int(mono), out.ctypes.data))
    return out[:len(last_keys)]


def project_local_map(pose7, cam4, bf, scales, log_scale, cols, rows, points, th, far_points=False, th_far=0.0, cos_limit=0.5):
    pose7, cam4, scales = [np.ascontiguousarray(a, np.float32) for a in (pose7, cam4, scales)]
    points = np.ascontiguousarray(points, MAP_POINT_DTYPE)
    out = np.zeros(max(len(points), 1), QUERY_DTYPE)
    _check(lib().tc2li_project_local_map(pose7.ctypes.data, cam4.ctypes.data, bf, scales.ctypes.data, len(scales), log_scale, cols, rows,
                                         len(points), points.ctypes.data, th, int(far_points), th_far, cos_limit, out.ctypes.data))
    return out[:len(points)]


# ---- local-map bookkeeping (Tracking::UpdateLocalKeyFrames / UpdateLocalPoints) ---------------------------------------------
class MapGraph(C.Structure):
    """tc2li_map_graph"""
    _fields_ = [("n_keyframes", C.c_int32), ("n_points", C.c_int32), ("kf_bad", C.c_void_p), ("covis_offsets", C.c_void_p), ("covis", C.c_void_p),
                ("child_offsets", C.c_void_p), ("children", C.c_void_p), ("parent", C.c_void_p), ("prev_kf", C.c_void_p),
                ("match_offsets", C.c_void_p), ("matches", C.c_void_p), ("point_bad", C.c_void_p), ("obs_offsets", C.c_void_p), ("obs_kf", C.c_void_p)]


class LocalMap:
    """Device-resident mirror of the keyframe graph (``tc2li_local_map``) and ``Tracking::UpdateLocalMap`` on it.  graph: dict of flat
    arrays kf_bad, covis_off, covis, child_off, children, parent, prev_kf, match_off, matches, point_bad, obs_off, obs_kf."""

    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().tc2li_local_map_create(C.byref(self.h)))
        self.n_keyframes = self.n_points = 0

    def set_graph(self, graph, stream=0):
        g = {k: np.ascontiguousarray(v, np.uint8 if k in ("kf_bad", "point_bad") else np.int32) for k, v in graph.items()}
        ptr = lambda k: g[k].ctypes.data if g[k].size else None
        mg = MapGraph(len(g["kf_bad"]), len(g["point_bad"]), ptr("kf_bad"), ptr("covis_off"), ptr("covis"), ptr("child_off"), ptr("children"),
                      ptr("parent"), ptr("prev_kf"), ptr("match_off"), ptr("matches"), ptr("point_bad"), ptr("obs_off"), ptr("obs_kf"))
        _check(lib().tc2li_local_map_set_graph(self.h, C.addressof(mg), C.c_void_p(stream)))
        self.n_keyframes, self.n_points = mg.n_keyframes, mg.n_points

    def update(self, frame_points, temporal_last_kf=-1, stream=0, keyframe_capacity=None, point_capacity=None):
        """-> (local keyframes, reference keyframe, local points, frame points cleared)"""
        fp = np.ascontiguousarray(frame_points, np.int32)
        kc = self.n_keyframes if keyframe_capacity is None else keyframe_capacity
        pc = self.n_points if point_capacity is None else point_capacity
        kfs, pts = np.zeros(max(kc, 1), np.int32), np.zeros(max(pc, 1), np.int32)
        cleared = np.zeros(max(len(fp), 1), np.uint8)
        n_k, n_p, ref = C.c_int32(0), C.c_int32(0), C.c_int32(-1)
        _check(lib().tc2li_local_map_update(self.h, fp.ctypes.data if len(fp) else None, len(fp), int(temporal_last_kf), kfs.ctypes.data, kc,
                                            C.addressof(n_k), C.addressof(ref), pts.ctypes.data, pc, C.addressof(n_p),
                                            cleared.ctypes.data if len(fp) else None, C.c_void_p(stream)))
        return kfs[:n_k.value].copy(), ref.value, pts[:n_p.value].copy(), cleared[:len(fp)].astype(bool)

    def close(self):
        if self.h:
            lib().tc2li_local_map_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
