#!/usr/bin/env python3
"""bench.py -- frames/s of the TC2LI-SLAM per-frame loop on MI355X (BASELINE.json metric: ORB+LiDAR front end + local BA).

A step = one pass of the hot path over one batch of `--frames` synthetic KITTI-sized frames that are already resident in
HBM (F independent sequences advancing one frame each):
  camera / tracking thread : stereo ORB (2 x 1242x375) -> stereo matching -> TrackWithMotionModel (projection matching
                             against the last frame + pose-only optimisation)
  LiDAR thread             : preprocess -> voxel filter -> 5-NN + plane-fit feature extraction (one 64-beam scan per frame)
  local-mapping thread     : every `--kf-interval`-th frame inserts a keyframe -> LocalLVBundleAdjustment (visual edges
                             + the LiDAR plane edge), F / kf_interval windows per step
which is the reference's thread structure (tracking, LiDAR front end, local mapping).  N > 1: one process per GPU;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time -- sequences are
independent units, so the path shards with no data-path collective ("weak": every rank processes its own frames).

Prints ONE JSON line on rank 0; DESIGN.md section "Measurement" explains how roofline / cpu_baseline are derived.
"""
import argparse
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def level_dims(w, h, nlevels=8, scale=1.2):
    dims, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        dims.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
    return dims


def algorithmic_bytes(w, h, nkp, lidar_counts):
    """Compulsory HBM bytes per image (ORB kernels) / per scan (LiDAR kernels) of each kernel group."""
    px = [a * b for a, b in level_dims(w, h)]
    raw, pre, down, sel = lidar_counts
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),           # read levels 0..6, write levels 1..7
        "fast": sum(px) + 4 * 15000,                      # read every level once, write the candidate list
        "blur": 2 * sum(px),                              # read + write every level
        "orient_describe": nkp * (709 + 512 + 64 + 16),  # patch gathers + descriptor/angle/key out
        "lidar_preprocess": 2 * raw * 32 + pre * 48,      # count pass + scatter pass read the raw scan, one write
        "lidar_voxel_hash": 3 * pre * 48 + pre * 4,       # bbox, insert, fill passes over the points + member list
        "lidar_voxel_centroid": pre * (48 + 4) + down * 48,
        "lidar_knn_plane": down * (48 + 48 + 48 + 5 * 8 + 27 * 16),  # point in, world + normvec out, neighbours, 27-cell gather
        "lidar_knn_hard": down * 0.02 * (48 + 48 + 48 + 5 * 8 + 125 * 16),  # the ~2 % of the queries that need the wider cubes
        "lidar_select": down * 1 + sel * 4 * 48,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames (stereo pair + scan) per step per GPU")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic frames rendered (tiled to --frames)")
    ap.add_argument("--kf-interval", type=int, default=4, help="a keyframe (one local BA window) every k-th frame")
    ap.add_argument("--ba-concurrency", type=int, default=8, help="local-BA windows in flight (streams) per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--front-end-only", action="store_true", help="configs[1]: leave the local BA out of the step")
    ap.add_argument("--split-ba", action="store_true", help="after the timed loop: ONE LV-BA window split over all ranks (landmark partition + "
                    "RCCL all-reduce of the shared-pose blocks, tc2li_local_lv_bundle_adjustment_sharded) next to the same window on one GPU; "
                    "reported as \"sharded_window\", not part of `value`")
    ap.add_argument("--stages", default="orb,track,lidar,ba", help="diagnostics: run only these stage threads in the timed loop")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    ge.build_native()
    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic, dist_util
    rank, local_rank, world = dist_util.rank_info()

    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = dist_util.init("nccl", rank, world, device=torch.device("cuda", local_rank))  # None when world == 1

    W, H = synthetic.WIDTH, synthetic.HEIGHT
    F, U = args.frames, min(args.unique, args.frames)
    bf = np.float32(synthetic.BF)
    b = np.float32(bf / np.float32(synthetic.FX))
    cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, bf]).astype(np.float64)

    # ---- synthetic workload: U distinct frames (scene seed per rank), tiled to F --------------------------------
    scene = synthetic.Scene(1000 * rank)
    uniq_img = np.empty((U, 2, H, W), np.uint8)
    scans = []
    for f in range(U):
        sc = synthetic.Scene(1000 * rank + f)
        uniq_img[f, 0], _ = sc.render(0.0, W, H, noise_seed=1)
        uniq_img[f, 1], _ = sc.render(synthetic.BASELINE, W, H, noise_seed=2)
        scans.append(synthetic.lidar_scan(scene, f + 1))
    tile = [f % U for f in range(F)]
    frames = uniq_img[tile]
    n_img = 2 * F
    dev_img = torch.from_numpy(frames.reshape(n_img, H, W)).cuda()
    raw = np.concatenate([scans[t] for t in tile])
    raw_offs = np.concatenate([[0], np.cumsum([len(scans[t]) for t in tile])]).astype(np.int32)
    dev_raw = torch.from_numpy(raw.view(np.uint8)).cuda()
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(t + 1)[:2]) for t in tile])
    stream = torch.cuda.current_stream().cuda_stream

    ext = pkg.OrbExtractor(max_width=W, max_height=H, max_images=n_img)
    ext2 = pkg.OrbExtractor(max_width=W, max_height=H, max_images=n_img)  # second feature buffer: extraction of batch k+1 overlaps tracking of batch k
    lidar = pkg.LidarFrontEnd(max_points_per_scan=int(max(len(s) for s in scans)), max_scans=F)
    # resident map: the world-frame down-sampled scan of frame 0 (what ikdtree.Build gets, LidarFrontEnd.cpp:918-931)
    boot = pkg.LidarMap()
    scan0 = synthetic.lidar_scan(scene, 0)
    down0 = lidar.voxel_filter(lidar.process(scan0))
    boot.Build(down0[:8])
    world0 = lidar.feature_extraction(boot, down0, pkg.pack_lidar_state(*synthetic.lidar_state(0)[:2]))["world"]
    lmap = pkg.LidarMap()
    lmap.Build(world0)
    maps = [lmap] * F

    # ---- tracking inputs: the "last frame" of every sequence = the frame's own stereo points seen from the previous pose
    orb_out = ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream)
    st_out = pkg.stereo_match_batch(ext, F, float(bf), float(b), stream=stream)
    fx, fy, cx, cy = [np.float32(v) for v in cam5[:4]]
    uniq_last = []
    for u in range(U):
        rng = np.random.default_rng(7000 + 1000 * rank + u)
        f = tile.index(u)
        n = int(orb_out[2][2 * f])
        kl, dl, z = orb_out[0][2 * f, :n].copy(), orb_out[1][2 * f, :n].copy(), st_out[1][f, :n].copy()
        order = rng.permutation(n)
        lk = kl[order]
        lk["angle"] = (lk["angle"] + rng.normal(0, 3, n).astype(np.float32)) % np.float32(360)
        zz = np.where(z[order] > 0, z[order], 1).astype(np.float32)
        Xw = np.stack([(lk["x"] - cx) * zz / fx, (lk["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
        md = dl[order]
        flips = rng.integers(0, 256, (n, 12))
        for k in range(12):  # descriptor drift between consecutive frames
            sel_rows = rng.random(n) < 0.7
            md[sel_rows, flips[sel_rows, k] // 8] ^= (np.uint8(1) << (flips[sel_rows, k] % 8).astype(np.uint8))
        uniq_last.append(dict(has_point=(z[order] > 0).astype(np.uint8), outlier=(rng.random(n) < 0.03).astype(np.uint8), Xw=Xw, keys=lk,
                              descriptors=md, pose7=np.array([0, 0, 0, 1, 0, 0, 0], np.float32)))
    last_frames = pkg.capi.pack_last_frames([uniq_last[t] for t in tile])
    ang = 0.002
    pose_pred = np.tile(np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, -0.08], np.float32), (F, 1))
    trk_out = None

    # ---- local mapping: one LV-BA window per keyframe (12 + 20 keyframes, ~3000 points, 6 clouds)
    n_ba = 0 if args.front_end_only else max(1, F // args.kf_interval)
    ba_batch = None
    ba_windows = []
    if n_ba:
        for k in range(min(4, n_ba)):
            w = synthetic.ba_window(100 * rank + k, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
            last = len(w["poses"]) - 1
            win = list(range(last, last - 6, -1))
            ba_windows.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), edges6=w["edges"],
                                   win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0,
                                   cam=w["cam"]))
        ba_batch = pkg.capi.BaBatch([ba_windows[k % len(ba_windows)] for k in range(n_ba)], ba_windows[0]["cam"])

    # Camera path, LiDAR path and local mapping are independent threads in the reference (src/examples/camera_lidar.cc:84,
    # System.cc: tracking / local mapping threads, queues in between); here each stage runs its steps on its own host thread and
    # HIP stream: ORB extraction -> (stereo matching + TrackWithMotionModel) as a two-deep pipeline over two feature buffers,
    # the LiDAR front end, and local mapping.  A run of K steps = every stage has processed K batches.
    import queue
    import threading
    lidar_stream = torch.cuda.Stream()
    track_stream = torch.cuda.Stream()
    exts = [ext, ext2]
    orb_outs = [orb_out, tuple(np.zeros_like(a) for a in orb_out)]
    st_outs = [st_out, None]
    trk_outs = [None, None]
    lidar_counts = None
    thread_ms = {}
    orb_times, lidar_times = [], []  # HIP-event stage durations of every call (the timed region is what run_steps did last)

    def run_steps(n_steps):
        nonlocal lidar_counts
        free = queue.Queue()
        ready = queue.Queue()
        free.put(0); free.put(1)
        errors = []

        def guard(fn):
            def wrapped():
                try:
                    fn()
                except BaseException as e:  # noqa: BLE001
                    errors.append(e)
                    ready.put(None)
            return wrapped

        def orb_thread():
            for _ in range(n_steps):
                k = free.get()
                orb_outs[k] = exts[k].extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_outs[k])
                orb_times.append(exts[k].last_timings().astype(float))
                ready.put(k)

        def track_thread():
            for _ in range(n_steps):
                k = ready.get()
                if k is None:
                    return
                st_outs[k] = pkg.stereo_match_batch(exts[k], F, float(bf), float(b), stream=track_stream.cuda_stream, out=st_outs[k])
                trk_outs[k] = pkg.capi.track_motion_model_batch(exts[k], F, orb_outs[k][0], st_outs[k][0], last_frames, pose_pred, cam5, float(b), 7.0,
                                                                stream=track_stream.cuda_stream, out=trk_outs[k])
                free.put(k)

        def lidar_thread():
            nonlocal lidar_counts
            for _ in range(n_steps):
                lidar_counts = lidar.frontend_batch(dev_raw.data_ptr(), raw_offs, maps, states, stream=lidar_stream.cuda_stream, want_points=False)[0]
                lidar_times.append(lidar.last_timings().astype(float))

        def ba_thread():
            for _ in range(n_steps):
                if ba_batch.run(args.ba_concurrency) != n_ba:
                    raise RuntimeError("a local BA window failed")

        fns = [orb_thread, track_thread, lidar_thread] + ([ba_thread] if ba_batch else [])
        want = set(args.stages.split(","))
        if "track" in want:
            want.add("orb")  # tracking consumes what the extraction produces
        fns = [f for f in fns if f.__name__.split("_")[0] in want]
        if "track" not in want and "orb" in want:  # nobody returns the feature buffers: the extraction thread recycles them itself
            for _ in range(n_steps):
                free.put(0)

        def timed(fn):
            def wrapped():
                torch.cuda.set_device(local_rank)  # HIP's current device is per thread; a new thread starts on device 0
                t = time.perf_counter()
                fn()
                thread_ms[fn.__name__] = 1e3 * (time.perf_counter() - t) / max(n_steps, 1)
            return wrapped
        fns = [timed(fn) for fn in fns]
        threads = [threading.Thread(target=guard(fn)) for fn in fns]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def barrier():
        dist_util.barrier(dist, torch.cuda.synchronize)

    if args.warmup:
        run_steps(args.warmup)
    barrier()
    orb_times.clear(); lidar_times.clear()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    if set(args.stages.split(",")) != {"orb", "track", "lidar", "ba"}:  # diagnostics: which stages slow each other down
        if rank == 0:
            print(json.dumps({"diagnostic_stages": args.stages, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / args.steps, 3),
                              "stage_thread_ms_per_step_concurrent": {k: round(v, 3) for k, v in thread_ms.items()}}))
        return
    loop_orb_ms, loop_lidar_ms = np.mean(orb_times, 0), np.mean(lidar_times, 0)
    loop_chunks = ext.last_chunks()
    elapsed = dist_util.max_elapsed(dist, time.perf_counter() - t0, device="cuda")
    stage_ms = ext.last_timings().astype(float)
    orb_out, st_out, trk_out = orb_outs[0], st_outs[0], trk_outs[0]
    if st_out is None:
        st_out = pkg.stereo_match_batch(ext, F, float(bf), float(b), stream=stream)
    nkp = float(np.mean(orb_out[2]))
    n_match = float(np.mean((st_out[1] > 0).sum(1)))
    lid_mean = [int(np.mean(np.diff(raw_offs)))] + [int(v) for v in np.mean(lidar_counts, 1)]

    # stage wall times of one more step, every stage alone (host clock, each stage synchronises at its end)
    wall = {}
    t_a = time.perf_counter()
    ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_out)
    wall["orb_extract_batch"] = time.perf_counter() - t_a; t_a = time.perf_counter()
    pkg.stereo_match_batch(ext, F, float(bf), float(b), stream=stream, out=st_out)
    wall["stereo_match_batch"] = time.perf_counter() - t_a; t_a = time.perf_counter()
    pkg.capi.track_motion_model_batch(ext, F, orb_out[0], st_out[0], last_frames, pose_pred, cam5, float(b), 7.0, stream=stream, out=trk_out)
    wall["track_motion_model_batch"] = time.perf_counter() - t_a; t_a = time.perf_counter()
    lidar.frontend_batch(dev_raw.data_ptr(), raw_offs, maps, states, stream=stream, want_points=False)
    wall["lidar_frontend_batch"] = time.perf_counter() - t_a; t_a = time.perf_counter()
    lidar_ms = lidar.last_timings().astype(float)
    if ba_batch:
        ba_batch.run(args.ba_concurrency)
        wall["local_lv_ba_batch(%d windows)" % n_ba] = time.perf_counter() - t_a; t_a = time.perf_counter()
        ba_batch.run(1)
        wall["local_lv_ba_batch, 1 window at a time"] = time.perf_counter() - t_a

    # ---- roofline of the dominant kernel: HIP-event durations of every launch of the timed region (the stages run
    # concurrently there, as in the rocprofv3 trace of the same command), on the stream each kernel is launched on; the same
    # kernels measured alone afterwards are reported next to them ----
    ext.set_profiling(True)
    prof = []
    for _ in range(3):
        ext.extract_batch_dev(dev_img.data_ptr(), n_img, W, H, W, W * H, stream=stream, out=orb_out)
        prof.append(ext.last_timings().astype(float))
    ext.set_profiling(False)
    prof = np.mean(prof, 0)
    lprof = []
    for _ in range(3):
        lidar.frontend_batch(dev_raw.data_ptr(), raw_offs, maps, states, stream=stream, want_points=False)
        lprof.append(lidar.last_timings().astype(float))
    lprof = np.mean(lprof, 0)

    def stage_table(o, l):
        return {"pyramid": o[0], "fast": o[1], "blur": o[3], "orient_describe": o[4], "lidar_preprocess": l[0], "lidar_voxel_hash": l[1],
                "lidar_voxel_centroid": l[2], "lidar_knn_plane": l[6], "lidar_knn_hard": l[7], "lidar_select": l[4]}
    isolated_ms = stage_table(prof, lprof)
    alg = algorithmic_bytes(W, H, nkp, lid_mean)
    kern_ms = stage_table(loop_orb_ms, loop_lidar_ms)
    units = {k: (F if k.startswith("lidar") else n_img) for k in kern_ms}
    # the ORB call pipelines chunks of images: each of its device stages is launched once per chunk, the event durations are the
    # chunks' sums (the profiling passes above run unchunked, so `measured alone` is one launch per stage)
    ch = max(1, loop_chunks)
    launches = {"pyramid": 7 * ch, "fast": ch, "blur": ch, "orient_describe": ch, "lidar_preprocess": 3, "lidar_voxel_hash": 8, "lidar_voxel_centroid": 1,
                "lidar_knn_plane": 1, "lidar_knn_hard": 1, "lidar_select": 3}
    names = {"fast": "k_fast_cells", "blur": "k_blur7_strips", "pyramid": "k_resize_linear", "orient_describe": "k_orient_describe",
             "lidar_preprocess": "k_pre_count+k_seg_scan+k_pre_scatter", "lidar_voxel_hash": "k_voxel_bbox..k_voxel_fill",
             "lidar_voxel_centroid": "k_voxel_centroid", "lidar_knn_plane": "k_knn_plane", "lidar_knn_hard": "k_knn_hard",
             "lidar_select": "k_sel_count+k_seg_scan+k_sel_scatter"}
    single = [k for k in kern_ms if k in ("pyramid", "fast", "blur", "orient_describe") or launches[k] == 1]  # groups made of one kernel (x launches)
    dom = max(single, key=lambda k: kern_ms[k])  # the kernel with the largest device time per step
    bytes_per_launch = alg[dom] * units[dom] / launches[dom]
    achieved = bytes_per_launch / (kern_ms[dom] / launches[dom] * 1e-3) / 1e9
    # HBM traffic of that kernel from the PMC passes of tools/profile_round.sh (FETCH_SIZE x2 + WRITE_SIZE per launch, see
    # tools/summarize_profile.py); the counters cannot be read inside this process, so the committed summary is used
    traffic = None
    for f in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        pmc = json.load(open(f))
        # template instantiations of one kernel (FAST: the 48 x 48 and the full-size window variant, launched back to back per chunk)
        # count as one launch of the stage
        hit = [v for k, v in pmc.items() if k.split("::")[-1].split("<")[0] == names[dom] and v.get("hbm_bytes_per_launch")]
        if hit:
            traffic = {"bytes_per_launch": int(sum(v["hbm_bytes_per_launch"] for v in hit)), "source": os.path.basename(f),
                       "note": "FETCH_SIZE x 2 + WRITE_SIZE per launch, from the rocprofv3 --pmc passes of this same command"}
            break
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic, "avg_launch_ms": round(kern_ms[dom] / launches[dom], 5),
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "launches_per_step": launches[dom],
                "all_kernels_ms": {k: round(v, 4) for k, v in kern_ms.items()},
                "all_kernels_ms_measured_alone": {k: round(v, 4) for k, v in isolated_ms.items()},
                "all_kernels_GBps": {k: round(alg[k] * units[k] / (kern_ms[k] * 1e-3) / 1e9, 2) for k in kern_ms if kern_ms[k] > 0}}

    # ---- CPU baseline: the oracle (a port) with the reference's threading -----------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed at N = 1 only
        from oracle import pyoracle
        pyoracle.build()
        L = pyoracle.lib()
        L.oracle_loop_frame.argtypes = ([C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int] + [C.c_void_p] * 5 +
                                        [C.c_float, C.c_int] + [C.c_void_p] * 8)
        ol, orr = pyoracle.OrbOracle(), pyoracle.OrbOracle()
        tree = pyoracle.KdTree(world0)
        nsel, nmat = C.c_int(0), C.c_int(0)
        pose_out = np.zeros(7)
        lasts6 = [pyoracle._kps_to_floats(u["keys"]) for u in uniq_last]
        ba_pool = ThreadPoolExecutor(max_workers=1)  # the local-mapping thread
        ba_futs = []

        def cpu_ba(k):
            w = ba_windows[k % len(ba_windows)]
            return pyoracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges6"], w["cam"], w["win_pose"], w["clouds"], w["Tcl7"], 1.0)[4]

        done, tcpu0 = 0, time.perf_counter()
        while done < 4 or (time.perf_counter() - tcpu0 < args.cpu_seconds and done < 600):
            t = tile[done % F]
            u = uniq_last[t]
            L.oracle_loop_frame(ol._h, orr._h, uniq_img[t, 0].ctypes.data, uniq_img[t, 1].ctypes.data, W, H, float(bf), float(b),
                                scans[t].ctypes.data, len(scans[t]), tree._h, states[done % F].ctypes.data, pose_pred[0].ctypes.data,
                                u["pose7"].ctypes.data, cam5.ctypes.data, 7.0, len(u["keys"]), u["has_point"].ctypes.data,
                                u["outlier"].ctypes.data, u["Xw"].ctypes.data, lasts6[t].ctypes.data, u["descriptors"].ctypes.data,
                                pose_out.ctypes.data, C.byref(nsel), C.byref(nmat))
            done += 1
            if n_ba and done % args.kf_interval == 0:
                ba_futs.append(ba_pool.submit(cpu_ba, done // args.kf_interval))
        for fut in ba_futs:
            fut.result()  # the loop is not finished before local mapping has caught up
        tcpu = time.perf_counter() - tcpu0
        cpu = {"value": round(done / tcpu, 3), "unit": "frames/s", "cores": 4 if n_ba else 3, "kind": "port",
               "sample": "%d synthetic frames of the same workload in %.1f s (%d local LV-BA windows); threads as in the reference: left/right "
                         "ORB on 2 threads, then stereo match + TrackWithMotionModel on the tracking thread; LiDAR front end on a 3rd "
                         "thread; local mapping (LV-BA, single-threaded g2o semantics) on a 4th" % (done, tcpu, len(ba_futs)),
               "host_cpus": os.cpu_count()}

    # ---- optional: one window over all ranks (BASELINE configs[4]); every rank passes the same window ----
    sharded_window = None
    if args.split_ba:
        w = synthetic.ba_window(4242, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
        last = len(w["poses"]) - 1
        win = list(range(last, last - 6, -1))
        clouds = synthetic.ba_window_clouds(w, win, n_points=3000)
        e = pkg.pack_ba_edges(w["edges"])
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(pkg.capi.RcclComm.unique_id()), dtype=torch.uint8))
        if dist is not None:
            dist.broadcast(uid, 0)
        comm = pkg.capi.RcclComm(uid.cpu().numpy().tobytes(), rank, world)
        shard = comm.shard()
        t_sh, t_one = [], []
        for _ in range(4):
            barrier()
            t_a = time.perf_counter()
            got = pkg.capi.local_lv_bundle_adjustment_sharded(shard, w["poses"], w["fixed"], w["points"], e, w["cam"], win_pose=win, clouds=clouds,
                                                              Tcl7=synthetic.TCL7, weight=1.0)
            t_sh.append(time.perf_counter() - t_a)
            t_a = time.perf_counter()
            one = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0)
            t_one.append(time.perf_counter() - t_a)
        comm.close()
        sharded_window = {"ranks": world, "ms_per_window": round(1e3 * min(t_sh[1:]), 3), "single_gpu_ms_per_window": round(1e3 * min(t_one[1:]), 3),
                          "iterations/trials": [int(got[4].iterations), int(got[4].trials)], "allreduces_per_window": 2 * int(got[4].trials) + int(got[4].iterations) + 3,
                          "max_pose_difference_vs_single_gpu": float(np.abs(got[0] - one[0]).max()),
                          "note": "landmarks l % ranks; per LM trial one sum of [S | b_schur | b_p] and one of [scale, chi2, stop]"}

    if rank == 0:
        total_frames = F * args.steps * world
        line = {
            "metric": "frames/sec (ORB+LiDAR front-end + local BA) on KITTI-00, 1/2/4/8 GPU; ATE vs ref",
            "value": round(dist_util.job_throughput(F, args.steps, world, elapsed), 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (ORB, matching), f32 (LiDAR), f64 (optimisation)", "data": "synthetic",
            "config": {"stage_threads": "ORB extraction | stereo matching + TrackWithMotionModel | LiDAR front end | local mapping, each on its own host "
                                        "thread and HIP stream as in the reference (tracking / LiDAR / local-mapping threads); a step = every stage "
                                        "has processed one batch",
                       "workload": ("configs[1]: KITTI-00 camera-LiDAR front end on 1xMI355X per rank" if args.front_end_only else
                                    "configs[1]+[2]: KITTI camera-LiDAR loop on 1xMI355X per rank: front end + HIP local LV-BA every %d-th frame"
                                    % args.kf_interval) +
                                   " -- stereo ORB (2 x 1242x375, 2000 features, 8 levels, FAST 20/7), stereo matching, "
                                   "TrackWithMotionModel (projection matching + pose optimisation), LiDAR preprocess / voxel 0.5 m / 5-NN "
                                   "plane features (64-beam scan, ~130k returns)" +
                                   ("" if args.front_end_only else ", LocalLVBundleAdjustment (12 free + 20 fixed keyframes, ~2500 points, "
                                                                   "~26k stereo edges, LiDAR plane edge over 6 keyframes x 3000 points)"),
                       "frames_per_step_per_gpu": F, "images_per_step_per_gpu": n_img, "ba_windows_per_step_per_gpu": n_ba,
                       "keypoints_per_image": round(nkp, 1), "stereo_matches_per_frame": round(n_match, 1),
                       "tracked_matches/inliers_per_frame": [round(float(np.mean(trk_out[2])), 1), round(float(np.mean(trk_out[3])), 1)],
                       "scan_points_raw/preprocessed/downsampled/selected": lid_mean, "map_points": int(lmap.size()),
                       "ba": None if not ba_batch else {"iterations": int(ba_batch.stats[0].iterations), "trials": int(ba_batch.stats[0].trials),
                                                        "planes": int(ba_batch.lstats[0].n_planes), "edges": int(len(ba_windows[0]["edges"]))}},
            "roofline": roofline, "cpu_baseline": cpu, **({"sharded_window": sharded_window} if sharded_window else {}),
            "stage_thread_ms_per_step_concurrent": {k: round(v, 3) for k, v in thread_ms.items()}, "stage_wall_ms_per_step": {k: round(1e3 * v, 3) for k, v in wall.items()},
            "orb_stage_ms_last_step": {"pyramid": round(stage_ms[0], 4), "fast": round(stage_ms[1], 4),
                                       "compact": round(stage_ms[2], 4), "blur": round(stage_ms[3], 4),
                                       "orient_describe": round(stage_ms[4], 4), "host_quadtree": round(stage_ms[5], 4),
                                       "host_until_quadtree": round(stage_ms[6], 4), "call_total": round(stage_ms[7], 4)},
            "lidar_stage_ms": {"preprocess": round(lidar_ms[0], 4), "voxel_hash": round(lidar_ms[1], 4), "voxel_centroid": round(lidar_ms[2], 4),
                               "knn_plane": round(lidar_ms[3], 4), "select": round(lidar_ms[4], 4), "total": round(lidar_ms[5], 4)},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
