"""The committed profile summaries of the current round are what bench.py's roofline entries are checked against: every kernel that
takes a visible share of the loop's device time there must have a price in bench.py's algorithmic_work (CPU-only, reads text files)."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"


def test_round_profiles_are_committed_and_stamped():
    for name in ("bench_kernel_stats.csv", "bench_line_under_rocprof.json", "pmc_traffic.json", "pmc_instruction_mix.json", "roofline_check.json"):
        assert os.path.isfile(os.path.join(ROOT, "profiles", "%s_%s" % (ROUND, name))), name
    traffic = json.load(open(os.path.join(ROOT, "profiles", ROUND + "_pmc_traffic.json")))
    assert re.fullmatch(r"[0-9a-f]{16}", traffic["_source_hash"])  # bench.py quotes `traffic` only for sources with this hash
    line = json.load(open(os.path.join(ROOT, "profiles", ROUND + "_bench_line_under_rocprof.json")))
    assert {"roofline", "value", "ms_per_step"} <= set(line)


def test_kernels_with_a_visible_share_are_priced():
    src = open(os.path.join(ROOT, "bench.py")).read()
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", ROUND + "_bench_kernel_stats.csv"))))
    unpriced = []
    for r in rows:
        m = re.search(r"tc2li::(k_[a-z0-9_]+)", r["Name"])
        if not m or float(r["Percentage"]) < 1.0 or m.group(1).startswith("k_diag_"):  # the peak micro-benchmarks are not part of the loop
            continue
        if '"%s"' % m.group(1) not in src:
            unpriced.append((m.group(1), r["Percentage"]))
    assert not unpriced, unpriced
