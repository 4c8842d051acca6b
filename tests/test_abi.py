"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares, refuses to compute
without a GPU (no silent fallback), and its host-only stage (quadtree) equals the oracle."""
import ctypes as C
import os

import numpy as np
import pytest


def test_library_exports_header_symbols(pkg):
    L = pkg.lib()
    syms = pkg.exported_symbols()
    assert "tc2li_orb_extract" in syms and "tc2li_orb_extract_batch" in syms and len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert pkg.abi_version() >= 1


def test_no_cpu_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.Tc2liError) as e:
        pkg.OrbExtractor()
    assert e.value.code == -3  # TC2LI_ERR_NO_DEVICE
    assert "no HIP device" in str(e.value)


def _fresh_process(code):
    """Runs `code` in a fresh interpreter (the knobs below must come before the process's first HIP call / first pool)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="4")
    env.pop("TC2LI_HOST_THREAD_BUDGET", None)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nimport tc2li_loader\npkg = tc2li_loader.load()\n" % root + code],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_set_hardware_queues():
    """The one runtime knob that is an ABI call: it sets what the HIP runtime reads when it initialises, refuses nonsense, and refuses
    to pretend once the library has called into HIP."""
    out = _fresh_process("""
import ctypes
pkg.capi.set_hardware_queues(8)
libc = ctypes.CDLL(None)
libc.getenv.restype = ctypes.c_char_p
print(libc.getenv(b"GPU_MAX_HW_QUEUES").decode())   # the C environment is what counts
for bad in (0, -1, 33):
    try:
        pkg.capi.set_hardware_queues(bad)
        print("accepted", bad)
    except pkg.capi.Tc2liError:
        pass
pkg.device_count()                                  # the library's first HIP call
try:
    pkg.capi.set_hardware_queues(8)
    print("accepted late")
except pkg.capi.Tc2liError as e:
    print("late:", "first HIP call" in str(e))
""")
    assert out.split() == ["8", "late:", "True"], out


def test_host_thread_budget_and_shutdown():
    """Pool sizes follow the budget (8 ranks on 256 cores: every rank's pools + stage threads fit its 32 cores; a one-GPU box keeps the
    tuned sizes), the budget is refused once a pool exists, tc2li_shutdown joins the pools and the library keeps working afterwards."""
    out = _fresh_process("""
import numpy as np
pkg.capi.set_host_thread_budget(32)
print(pkg.capi.host_threads())
pkg.capi.set_host_thread_budget(256)
print(pkg.capi.host_threads())
import threading
n0 = threading.active_count()
def os_threads():
    import os
    return len(os.listdir('/proc/self/task'))
t0 = os_threads()
pkg.capi.set_host_thread_budget(32)
import ctypes as C
capi = pkg.capi
n = 8
def preintegrate():                                                   # a host-only entry that runs on the tracking pool
    pre, bias = (capi.PreintegratedPOD * n)(), (capi.ImuBias * n)()
    smp = np.zeros(12 * n, capi.IMU_SAMPLE_DTYPE)
    for name in smp.dtype.names:
        smp[name] = np.random.default_rng(1).normal(0, 0.1, smp[name].shape)
    tname = [k for k in smp.dtype.names if k.startswith("t")][0]
    smp[tname] = np.tile(np.arange(12) * 0.01, n)
    off = (np.arange(n + 1) * 12).astype(np.int32)
    t1, t2 = np.full(n, 0.005), np.full(n, 0.105)
    f = capi.lib().tc2li_imu_preintegrate_frames
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert f(n, C.addressof(pre), C.addressof(bias), 1e-3, 1e-2, 1e-5, 1e-4, smp.ctypes.data, off.ctypes.data, t1.ctypes.data, t2.ctypes.data) == n
    return bytes(pre)
a = preintegrate()                                                    # makes the tracking pool
t1 = os_threads()
try:
    pkg.capi.set_host_thread_budget(64)
    print("accepted with a live pool")
except pkg.capi.Tc2liError:
    pass
pkg.capi.shutdown()
t2 = os_threads()
pkg.capi.set_host_thread_budget(64)                                  # fine again: no pool exists
b = preintegrate()                                                    # and the pool is made again on demand
pkg.capi.shutdown()
print(t1 - t0, t2 - t0, a == b)
""")
    lines = out.strip().splitlines()
    small, big = eval(lines[0]), eval(lines[1])
    assert small["budget"] == 32 and big["budget"] == 256
    # 8 ranks on a 256-core node: extractor + tracking + LiDAR pools + 3 lock-step groups + the 5 stage threads of the caller -- at most eight
    # threads per CPU of the budget (round 5: the budget counts CPUs really granted, a pool thread waits for its stream half of its time)
    assert small["extractor_pool"] + small["tracking_pool"] + small["lidar_pool"] + 3 * small["ba_group_pool"] + 5 <= 8 * 32
    assert (big["extractor_pool"], big["tracking_pool"], big["lidar_pool"], big["ba_group_pool"]) == (32, 16, 16, 16)
    made, left, same = lines[2].split()
    assert int(made) == small["tracking_pool"] - 1 and int(left) == 0 and same == "True", lines[2]


def test_product_does_not_reference_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg_dir = os.path.join(root, "tc2li-slam_amd")
    for dp, _, files in os.walk(pkg_dir):
        if os.sep + "build" in dp or os.sep + "lib" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text, os.path.join(dp, f)


@pytest.mark.parametrize("seed,n,w,h,target", [(0, 5000, 1210, 343, 434), (1, 300, 315, 73, 122), (2, 40, 500, 200, 100),
                                               (3, 1, 400, 300, 50), (4, 0, 400, 300, 50), (5, 2500, 640, 640, 700),
                                               (6, 900, 980, 260, 1)])
def test_host_quadtree_equals_oracle(pkg, oracle, seed, n, w, h, target):
    rng = np.random.default_rng(seed)
    # distinct integer pixel positions in cv::FAST emission order is not required by the tree; use row-major order
    pos = rng.choice(w * h, size=n, replace=False) if n else np.zeros(0, np.int64)
    pos.sort()
    xyr = np.stack([pos % w, pos // w, rng.integers(7, 120, n)], 1).astype(np.float32).reshape(-1, 3)
    o = oracle.OrbOracle()
    want = o.distribute(xyr, 16, 16 + w, 16, 16 + h, target)
    got = pkg.distribute_quadtree_host(xyr, 16, 16 + w, 16, 16 + h, target)
    assert np.array_equal(got, want)
    if n:
        assert len(got) >= min(n, 1)


def test_host_quadtree_on_real_candidates(pkg, oracle, synthetic):
    left, _ = synthetic.stereo_pair(7, 800, 300)
    o = oracle.OrbOracle()
    o.extract(left)
    per_level = o.tables()[1]
    for lvl in range(8):
        c = o.candidates(lvl).copy()
        lh, lw = o.level(lvl).shape
        c[:, :2] -= 16
        want = o.distribute(c, 16, lw - 16, 16, lh - 16, int(per_level[lvl]))
        got = pkg.distribute_quadtree_host(c, 16, lw - 16, 16, lh - 16, int(per_level[lvl]))
        assert np.array_equal(got, want), lvl
