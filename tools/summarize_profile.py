"""Condenses a tools/profile_round.sh run into the files committed under profiles/: kernel stats (csv) and per-kernel HBM
traffic from the FETCH_SIZE / WRITE_SIZE passes (json).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(128-B requests tallied at 64 B); both counters are reported in KiB by rocprofv3."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], "profiles/%s_bench_kernel_stats.csv" % tag)
stats_i = glob.glob(os.path.join(out, "trace_i", "**", "*kernel_stats.csv"), recursive=True)
if stats_i:
    shutil.copy(stats_i[0], "profiles/%s_inertial_kernel_stats.csv" % tag)
for sub, name in (("trace_64", "seq64_kernel_stats.csv"), ("trace_orb", "alone_orb_kernel_stats.csv")):
    st = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], "profiles/%s_%s" % (tag, name))
line = [l for l in open(os.path.join(out, "bench_trace.log")) if l.startswith("{")]
if line:
    open("profiles/%s_bench_line_under_rocprof.json" % tag, "w").write(line[-1])


def counter_per_kernel(sub, name):
    acc = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


sys.path.insert(0, os.getcwd())
import bench  # noqa: E402  (source_hash: the traffic figures are only quoted for the sources they were measured on)


def traffic(fetch_dir, write_dir):
    fetch, write = counter_per_kernel(fetch_dir, "FETCH_SIZE"), counter_per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        res[k] = {"launches": max(f[0], w[0]),
                  "fetch_bytes_per_launch": (2.0 * 1024.0 * f[1] / f[0]) if f[0] else None,   # gfx950 correction: x2
                  "write_bytes_per_launch": (1024.0 * w[1] / w[0]) if w[0] else None}
        fb, wb = res[k]["fetch_bytes_per_launch"], res[k]["write_bytes_per_launch"]
        res[k]["hbm_bytes_per_launch"] = (fb or 0.0) + (wb or 0.0)
    return res


res = traffic("pmc_fetch", "pmc_write")
res["_source_hash"] = bench.source_hash()
json.dump(res, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1, sort_keys=True)
del res["_source_hash"]
res_i = traffic("pmc_fetch_i", "pmc_write_i")  # the inertial loop's own passes (configs[3])
if res_i:
    res_i["_source_hash"] = bench.source_hash()
    json.dump(res_i, open("profiles/%s_pmc_traffic_inertial.json" % tag, "w"), indent=1, sort_keys=True)
top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
for k, v in top:
    print("%-40s launches %5d  fetch %12.0f B  write %12.0f B per launch" % (k[:40], v["launches"], v["fetch_bytes_per_launch"] or 0, v["write_bytes_per_launch"] or 0))


# ---- matrix-unit occupancy (the Schur GEMM is the only MFMA user): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs that
# ran the kernel; SQ_BUSY_CYCLES is the time the SQ had any wave of the dispatch, per shader engine / XCD instance as rocprofv3 sums it
mf = counter_per_kernel("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
busy = counter_per_kernel("pmc_mfma", "SQ_BUSY_CYCLES")
wave = counter_per_kernel("pmc_mfma", "SQ_WAVE_CYCLES")
def kernel_durations(sub):
    """{kernel: [launches, total ns]} from the kernel trace of a --pmc pass (the same dispatches the counters belong to)"""
    acc = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc


mdur = kernel_durations("pmc_mfma")
mres = {}
for k, (n, v) in mf.items():
    if v <= 0:
        continue
    b, w = busy.get(k, [0, 0.0]), wave.get(k, [0, 0.0])
    mres[k] = {"launches": n, "mfma_busy_cycles_per_launch": v / n, "sq_busy_cycles_per_launch": b[1] / b[0] if b[0] else None,
               "sq_wave_cycles_per_launch": w[1] / w[0] if w[0] else None}
    if k in mdur and mdur[k][0]:
        # the pass's own launch duration next to the counters -- one v_mfma_f64_16x16x4_f64 (2048 FLOP) keeps the pipe busy for 64
        # counted cycles (k_diag_mfma_f64: 2^27 instructions per launch, 2^33 busy cycles), so busy / 64 x 2048 / duration is the executed rate
        dur_ns = mdur[k][1] / mdur[k][0]
        mres[k]["avg_duration_ns_this_pass"] = dur_ns
        mres[k]["mfma_instructions_per_launch"] = v / n / 64.0
        mres[k]["executed_tflops_from_counters"] = v / n / 64.0 * 2048.0 / dur_ns / 1e3
if mres:
    json.dump(mres, open("profiles/%s_pmc_mfma.json" % tag, "w"), indent=1, sort_keys=True)
    for k, v in mres.items():
        print("MFMA %-40s launches %5d  mfma busy %12.0f  sq busy %12.0f  wave cycles %12.0f per launch" %
              (k[:40], v["launches"], v["mfma_busy_cycles_per_launch"], v["sq_busy_cycles_per_launch"] or 0, v["sq_wave_cycles_per_launch"] or 0))


# ---- instruction mix / LDS behaviour per kernel (SQ counters summed over the dispatch's waves), with the kernel's average duration
# from the stats CSV beside it: VALU instructions per nanosecond against the chip's issue rate says whether a kernel is ALU-bound
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_ANY"]
mix = {}
for sub, group in (("pmc_insts", names[:4]), ("pmc_lds", names[4:])):
    for c in group:
        for k, (n, v) in counter_per_kernel(sub, c).items():
            mix.setdefault(k, {})[c + "_per_launch"] = v / n if n else None
            mix[k]["launches"] = n
dur = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        dur[r["Name"].split("(")[0]] = float(r["AverageNs"])
for k, v in mix.items():
    if k in dur:
        v["avg_duration_ns_kernel_stats"] = dur[k]
        if v.get("SQ_INSTS_VALU_per_launch"):
            # 256 CUs x 4 SIMD-32s: a wave64 VALU instruction takes two cycles of its SIMD (MI355X_MICROARCH.md, wave scheduling), 2.4 GHz:
            # 1228.8 wave-instructions per ns.  (Rounds 1-2 divided by 2457.6 -- one instruction per cycle -- and read the ORB kernels as
            # "18 % VALU"; they sit at about half of the issue rate when alone, and their time scales with their instruction count.)
            v["valu_issue_frac"] = v["SQ_INSTS_VALU_per_launch"] / dur[k] / 1228.8
if mix:
    json.dump(mix, open("profiles/%s_pmc_instruction_mix.json" % tag, "w"), indent=1, sort_keys=True)
    for k, v in sorted(mix.items(), key=lambda kv: -(kv[1].get("SQ_INSTS_VALU_per_launch") or 0) * kv[1].get("launches", 0))[:10]:
        print("MIX %-36s valu %12.0f lds %11.0f conflicts %11.0f per launch, valu issue frac %s" %
              (k[:36], v.get("SQ_INSTS_VALU_per_launch") or 0, v.get("SQ_INSTS_LDS_per_launch") or 0, v.get("SQ_LDS_BANK_CONFLICT_per_launch") or 0,
               ("%.3f" % v["valu_issue_frac"]) if "valu_issue_frac" in v else "-"))


# ---- the line's roofline against the CSV of the same run: the dominant kernel's average duration as bench.py measured it (start / stop
# events of every dispatch) next to rocprofv3's, and the roofline fraction recomputed from the CSV duration
if line and stats:
    try:
        d = json.loads(line[-1])
        r = d.get("roofline") or {}
        name = r.get("kernel")
        rows = [x for x in csv.DictReader(open(stats[0])) if x["Name"].split("(")[0].replace("void ", "").replace("tc2li::", "").split("<")[0] == name]
        if name and rows:
            calls = sum(int(x["Calls"]) for x in rows)
            tot = sum(float(x["TotalDurationNs"]) for x in rows)
            csv_ms = tot / calls / 1e6
            per_launch = r.get("algorithmic_bytes_per_launch") or r.get("algorithmic_flops_per_launch")
            chk = {"kernel": name, "bench_avg_launch_ms": r.get("avg_launch_ms"), "rocprof_csv_avg_launch_ms": round(csv_ms, 6), "rocprof_calls": calls,
                   "bench_over_csv": round(r.get("avg_launch_ms", 0) / csv_ms, 4), "bench_frac": r.get("frac")}
            if per_launch and r.get("peak"):
                scale = 1e9 if r.get("unit") == "GB/s" else 1e12
                chk["frac_from_csv_duration"] = round(per_launch / (csv_ms * 1e-3) / scale / r["peak"], 5)
            json.dump(chk, open("profiles/%s_roofline_check.json" % tag, "w"), indent=1)
            print("ROOFLINE CHECK", chk)
    except Exception as e:  # noqa: BLE001
        print("roofline check skipped:", e)
