#!/usr/bin/env python3
"""Copies the outputs of `tools/profile.sh <tag>` (+ gpurun_out/gpu_pytest.log) from gpurun_out/ into profiles/ under the
round's names and rewrites profiles/traffic.json for the kernel sources they were measured on.
   python tools/install_evidence.py r02final r02_final"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles")
b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
u = json.loads(open(src + "/bench_under_rocprof.json").read().strip().splitlines()[-1])
assert bench.kernel_source_hash() == b["config"]["kernel_source_hash"], "the evidence was measured on other kernel sources"
for a, c in (("bench.json", "_bench_3gb.json"), ("bench.log", "_bench_3gb.log"), ("pmc/pmc_FETCH_SIZE.csv", "_bench_pmc_FETCH_SIZE.csv"),
             ("pmc/pmc_WRITE_SIZE.csv", "_bench_pmc_WRITE_SIZE.csv"), ("trace/bench_kernel_stats.csv", "_bench_kernel_stats.csv"),
             ("bench_under_rocprof.json", "_bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, name + c))
log = os.path.join(ROOT, "gpurun_out", "gpu_pytest.log")
if os.path.exists(log):
    shutil.copy(log, os.path.join(dst, name.replace("_final", "") + "_gpu_pytest.log"))
r, t = b["roofline"], b["roofline"]["traffic_info"]
old = json.load(open(dst + "/traffic.json"))
kind = "seq" if b["config"].get("index", "").startswith("sequence-only") else "full"
json.dump({b["config"]["workload"] + ":" + kind: r["traffic"], "kernel_source_hash": b["config"]["kernel_source_hash"],
           "_note": "HBM bytes per launch of the dominant kernel (" + r["kernel"] + "), measured by bench.py's own rocprofv3 PMC passes: 2 x FETCH_SIZE*1024 + WRITE_SIZE*1024; "
                    "gfx950 correction x2 on FETCH_SIZE (MI355X_MICROARCH section HBM; calibration profiles/r01_fetch_size_calibration.csv). Key = workload:index kind. "
                    "bench.py uses this file only when rocprofv3 is unavailable AND kernel_source_hash equals the hash of the current kernel sources.",
           "_raw_fetch_bytes": t["raw_fetch_bytes"], "_raw_write_bytes": t["raw_write_bytes"], "_kmers_per_launch": r["kmers_per_launch"],
           "_bytes_per_kmer_corrected": r["bytes_per_kmer"], "_lines_per_kmer": r["lines_per_kmer"], "_index_gb": b["config"]["index_gb"]},
          open(dst + "/traffic.json", "w"), indent=1)
c, cb = b["config"], b["cpu_baseline"]
print("value %.2f G k-mers/s  ms/step %.2f  kernel_ms %.2f  value_8d %.2f G (%.1f ms)  pageable %.2f G  frac %.3f  traffic %.2f GB  %.2f B/k-mer  %.4f lines/k-mer  overfetch %.2f  frac_of_gather_ceiling %.3f  index %.1f GB"
      % (b["value"] / 1e9, b["ms_per_step"], r["kernel_ms"], b["value_8d"] / 1e9, min(c["h2d_inclusive"]["pinned_source_s"], c["h2d_inclusive"]["pageable_source_s"]) * 1e3,
         c["kmers_per_s_h2d_inclusive_pageable"] / 1e9, r["frac"], r["traffic"] / 1e9, r["bytes_per_kmer"], r["lines_per_kmer"],
         r["overfetch_vs_useful"], r.get("frac_of_gather_ceiling") or 0.0, c["index_gb"]))
print("cpu_baseline %.1f M contig-scheduled / %.1f M tiled; under rocprof: %.2f G, kernel_ms %.2f; raw FETCH %.2fe6 KB"
      % (cb["value"] / 1e6, cb["value_position_tiled"] / 1e6, u["value"] / 1e9, u["roofline"]["kernel_ms"], t["raw_fetch_bytes"] / 1024 / 1e6))
for l in open(src + "/trace/bench_kernel_stats.csv"):
    if "mfx_hist_kernel" in l:
        print(l.strip()[:200])
