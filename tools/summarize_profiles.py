#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/gpu_round.sh (gpurun_out/) into the committed, judged
summaries under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and
<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE PMC passes, corrected as MI355X_MICROARCH.md
§HBM prescribes: FETCH_SIZE x2 on gfx950, both counters are in KB)."""
import collections
import csv
import json
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = "gpurun_out"
shutil.copy(f"{src}/prof_stats/r_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
import os
for a, b in [("bench_bf16.json", "bench_bf16.json"), ("bench_f16.json", "bench_f16.json"),
             ("bench_uint8.json", "bench_uint8.json"),
             ("prof_stats_cfg2/r_kernel_stats.csv", "kernel_stats_cfg2_multilingual_bf16.csv"),
             ("conv16_fused_pair.txt", "conv16_fused_pair.txt"),
             ("conv_microbench.txt", "conv_microbench.txt"), ("resblock_chain.txt", "resblock_chain.txt"),
             ("stream_v1.json", "stream_v1.json"), ("stream_vits2_vocos.json", "stream_vits2_vocos.json"),
             ("stream_v1_old_path.json",) * 2, ("b1_summary.txt", "b1_trace_summary.txt"),
             ("bench_vocos.json", "bench_vocos.json"), ("bench_vits2_vocos.json", "bench_vits2_vocos.json"),
             ("bench_cfg2_multilingual_bf16.json",) * 2, ("bench_cfg2_multilingual_f32.json",) * 2,
             ("bench_cfg3_aishell3.json",) * 2, ("bench_cfg4_stress48k_f16.json",) * 2,
             ("bench_gpus2_refused.err",) * 2, ("bench_2rank_dryrun.json",) * 2,
             ("pytest_gpu.log",) * 2]:
    if os.path.exists(f"{src}/{a}"):
        shutil.copy(f"{src}/{a}", f"profiles/{tag}_{b}")


def agg(path, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


f = agg(f"{src}/pmc_fetch/r_counter_collection.csv", "FETCH_SIZE")
w = agg(f"{src}/pmc_write/r_counter_collection.csv", "WRITE_SIZE")
# durations from the un-countered stats run
stats = {r["Name"]: r for r in csv.DictReader(open(f"{src}/prof_stats/r_kernel_stats.csv"))}
out = {"note": "per kernel name; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch "
               "(gfx950: FETCH_SIZE reports half of a coalesced stream; narrower accesses are "
               "uncalibrated, so read this as an upper bound on reads)",
       "kernels": {}}
dom = {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0}
for k in sorted(f, key=lambda k: -sum(f[k])):
    n = len(f[k])
    ent = {"launches_in_pmc_run": n, "fetch_size_kb_per_launch": sum(f[k]) / n,
           "write_size_kb_per_launch": sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))),
           }
    ent["hbm_bytes_per_launch"] = (2 * ent["fetch_size_kb_per_launch"] +
                                   ent["write_size_kb_per_launch"]) * 1024
    if k in stats:
        ent["avg_duration_ns"] = float(stats[k]["AverageNs"])
        ent["calls_in_stats_run"] = int(stats[k]["Calls"])
    out["kernels"][k] = ent
    # dominant kernel class = the MRF ResBlock launches: the chain / pair kernels and the conv_mfma
    # instantiations whose MRF template flag is set (only run_hifigan's ResBlock launches set it)
    mm = re.search(r"conv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, (true|false)", k)
    is_dom = (mm and mm.group(1) == "true") or "resblock_pair32_kernel" in k or \
        "resblock_chain32_kernel" in k
    if is_dom:
        dom["launches"] += n
        dom["fetch_kb"] += sum(f[k])
        dom["write_kb"] += sum(w.get(k, [0]))
        if k in stats:
            dom["stat_calls"] = dom.get("stat_calls", 0) + int(stats[k]["Calls"])
            dom["stat_ns"] = dom.get("stat_ns", 0.0) + float(stats[k]["TotalDurationNs"])
out["dominant_conv_mfma"] = {
    "launches": dom["launches"],
    "hbm_bytes_per_launch": (2 * dom["fetch_kb"] + dom["write_kb"]) * 1024 / max(1, dom["launches"]),
    "raw_fetch_bytes_per_launch": dom["fetch_kb"] * 1024 / max(1, dom["launches"]),
    "write_bytes_per_launch": dom["write_kb"] * 1024 / max(1, dom["launches"]),
    # rocprofv3 --stats view of the same class (compare with bench.py roofline.avg_launch_ms)
    "rocprof_avg_duration_ms": dom.get("stat_ns", 0.0) / max(1, dom.get("stat_calls", 0)) / 1e6,
    "rocprof_calls": dom.get("stat_calls", 0),
}
json.dump(out, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out["dominant_conv_mfma"], indent=1))
