#!/usr/bin/env python3
"""Copies what `tools/run_evidence.sh <tag> ...` left in gpurun_out/ into profiles/ (same names) and rewrites profiles/traffic.json for the
kernel sources the bench line was measured on (bench.py uses that file only when rocprofv3 is unavailable or at N > 1, and only for the
same kernel_source_hash).   python tools/install_round.py r06"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
names = ["_bench_final.json", "_bench_final.log", "_final_bench_kernel_stats.csv", "_gpu_suite.txt", "_smoke.txt", "_repeats_rates.txt", "_repeats_trace.txt", "_valu.txt",
         "_bench_rehearse_2ranks.json", "_isa_hot_loop.txt"]
for n in names:
    p = os.path.join(src, tag + n)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, tag + n))
        print("installed", tag + n)
pm = os.path.join(src, tag + "_pmc_final")
if os.path.isdir(pm):
    shutil.rmtree(os.path.join(dst, tag + "_pmc_final"), ignore_errors=True)
    shutil.copytree(pm, os.path.join(dst, tag + "_pmc_final"))
    print("installed", tag + "_pmc_final/", sorted(os.listdir(pm)))
b = json.loads(open(os.path.join(src, tag + "_bench_final.json")).read().strip().splitlines()[-1])
assert bench.kernel_source_hash() == b["config"]["kernel_source_hash"], "the bench line was measured on other kernel sources (%s, now %s)" % (b["config"]["kernel_source_hash"], bench.kernel_source_hash())
r, t = b["roofline"], b["roofline"]["traffic_info"]
kind = "seq" if b["config"].get("index", "").startswith("sequence-only") else "full"
json.dump({b["config"]["workload"] + ":" + kind: r["traffic"], "kernel_source_hash": b["config"]["kernel_source_hash"],
           "_note": "HBM bytes per launch of the dominant kernel (" + r["kernel"] + "), measured by bench.py's own rocprofv3 PMC passes: 2 x FETCH_SIZE*1024 + WRITE_SIZE*1024; "
                    "gfx950 correction x2 on FETCH_SIZE (MI355X_MICROARCH section HBM; calibration profiles/r01_fetch_size_calibration.csv). Key = workload:index kind. "
                    "bench.py uses this file only when rocprofv3 is unavailable (or at N > 1: bytes per k-mer x the rank's k-mers) AND kernel_source_hash equals the hash of the current kernel sources.",
           "_raw_fetch_bytes": t["raw_fetch_bytes"], "_raw_write_bytes": t["raw_write_bytes"], "_kmers_per_launch": r["kmers_per_launch"],
           "_bytes_per_kmer_corrected": r["bytes_per_kmer"], "_lines_per_kmer": r["lines_per_kmer"], "_index_gb": b["config"]["index_gb"]},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
c = b["config"]
print("value %.2f G  ms/step %.3f  kernel_ms %.3f  frac %.3f  lines/k-mer %.4f  VALU/k-mer %.1f  gather frac %.3f" %
      (b["value"] / 1e9, b["ms_per_step"], r["kernel_ms"], r["frac"], r["lines_per_kmer"], r["issue"]["valu_insts_per_kmer"], r.get("frac_of_gather_ceiling") or 0))
print("value_8d %.2f G (best %.2f, first call %.3f s)  model %s" % (b["value_8d"] / 1e9, b["value_8d_best"] / 1e9, b["value_8d_first_call_s"],
      {k: round(v / 1e9, 1) for k, v in (b.get("value_8d_model") or {}).items()}))
print("value_full_index %.2f G  value_k31 %.2f G (frac %.3f, VALU %.1f)  value_repeats %.2f G" %
      (b["value_full_index"] / 1e9, b["value_k31"] / 1e9, b["k31"]["roofline"]["frac"], b["k31"]["roofline"]["issue"]["valu_insts_per_kmer"], b["value_repeats"] / 1e9))
for L, row in b["repeats"]["levels"].items():
    print("  repeats %s %%: %.2f G  %.3f of the i.i.d. kernel rate  side share %.4f  listed %.4f" % (L, row["value"] / 1e9, row["of_the_iid_kernel_rate"], row.get("side_table_share", 0), row.get("listed_share", 0)))
print("  repeats roofline frac %.3f, parity %r" % (b["repeats"]["roofline"]["frac"], (b["repeats"].get("oracle_parity") or {}).get("gpu_parity_on_sample")))
e = b["e2e"]
print("e2e wall %.2f s (b2b mean %.2f), index_build %.2f; placed wall %.2f, build %.2f; from text: %s" % (e["wall_s"], e["wall_back_to_back_mean_s"], e["index_build_s"],
      e["placed"]["wall_s"], e["placed"]["index_build_s"], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in (e.get("from_text") or {}).items() if k != "note"}))
cb = b["cpu_baseline"]
print("cpu_baseline %.2f M contig-scheduled / %.2f M tiled on %d cores, parity %r" % (cb["value"] / 1e6, cb["value_position_tiled"] / 1e6, cb["cores"], cb["gpu_parity_on_sample"]))
ks = os.path.join(dst, tag + "_final_bench_kernel_stats.csv")
if os.path.exists(ks):
    for l in open(ks):
        if "mfx_hist" in l or "mfx_sum" in l:
            print(l.strip()[:220])
