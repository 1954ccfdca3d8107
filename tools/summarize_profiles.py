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

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = "gpurun_out"
import os
for a, b in [("prof_stats_baker/r_kernel_stats.csv", "kernel_stats.csv"), ("bench.json",) * 2,
             ("prof_stats_baker_serial/r_kernel_stats.csv", "kernel_stats_serial.csv"), ("bench_serial.json",) * 2,
             ("bench_bf16.json",) * 2, ("bench_f16.json",) * 2, ("bench_uint8.json",) * 2,
             ("bench_ungrouped.json",) * 2,
             ("prof_stats_cfg2/r_kernel_stats.csv", "kernel_stats_cfg2_multilingual_bf16.csv"),
             ("prof_stats_bf16/r_kernel_stats.csv", "kernel_stats_bf16.csv"),
             ("prof_stats_stress48k/r_kernel_stats.csv", "kernel_stats_cfg4_stress48k_f16.csv"),
             ("prof_stats_vocos/r_kernel_stats.csv", "kernel_stats_vocos.csv"),
             ("prof_stats_vits2vocos/r_kernel_stats.csv", "kernel_stats_vits2_vocos_v1.csv"),
             ("prof_stats_uint8/r_kernel_stats.csv", "kernel_stats_uint8.csv"),
             ("pytest_gpu_margins.txt",) * 2, ("stream_v1_graph.json",) * 2,
             ("conv16_fused_pair.txt",) * 2, ("conv_microbench.txt",) * 2, ("pw_gemm_microbench.txt",) * 2,
             ("stream_v1.json",) * 2, ("stream_vits2_vocos.json",) * 2, ("mas.json", "mas_bench.json"),
             ("bench_vocos.json",) * 2, ("bench_vits2_vocos.json",) * 2,
             ("bench_cfg2_multilingual_bf16.json",) * 2, ("bench_cfg2_multilingual_f32.json",) * 2,
             ("bench_cfg3_aishell3.json",) * 2, ("bench_cfg3_aishell3_padded.json",) * 2,
             ("bench_cfg4_stress48k_f16.json",) * 2,
             ("bench_gpus2_refused.err",) * 2, ("bench_2rank_dryrun.json",) * 2, ("pytest_gpu.log",) * 2,
             ("b1_summary.txt", "b1_anatomy.txt"), ("stream_v1_bf16.json",) * 2,
             # SQ counters per kernel symbol (tools/gpu_sq_counters.sh + tools/sq_table.py): f32 MRF class, 16-bit classes
             ("sq_table_mrf.txt", "sq_counters_mrf.txt"), ("sq_table_mrf16_cfg2.txt", "sq_counters_mrf16_cfg2.txt"),
             ("sq_table_mrf16_stress48k.txt", "sq_counters_mrf16_stress48k.txt")]:
    # gpurun_out/ accumulates over rounds (every call merges into it): only what THIS pass wrote is copied -- files not
    # older than the pass's library-digest stamp
    fresh_after = os.path.getmtime(f"{src}/lib_digest.txt") - 5 if os.path.exists(f"{src}/lib_digest.txt") else 0
    if os.path.exists(f"{src}/{a}") and os.path.getmtime(f"{src}/{a}") >= fresh_after:
        shutil.copy(f"{src}/{a}", f"profiles/{tag}_{b}")


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import agg, is_mrf, is_pw, is_u8, is_mrf16 as _mrf16  # noqa: E402  (shared with bench.py --live-traffic)


out = {"note": "per kernel name; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch "
               "(gfx950: FETCH_SIZE reports half of a coalesced stream; narrower accesses are "
               "uncalibrated, so read this as an upper bound on reads).  Class keys: dominant_conv_mfma = the f32 "
               "headline's MRF class, mrf16_<config> = the 16-bit MRF class of bench.py --config <config> "
               "(baker = --decoder-dtype bf16)"}
# key in the JSON -> (sub-directory tag of the three rocprofv3 passes, kernel-class predicate,
#                     launches of the class counted as: kernels (None) or this kernel-name substring only)
CLASSES = (("dominant_conv_mfma", "baker", is_mrf, None), ("mrf16_baker", "bf16", _mrf16, None),
           ("mrf16_stress48k", "stress48k", _mrf16, None), ("mrf16_multilingual", "cfg2", _mrf16, None),
           ("pw_vocos", "vocos", is_pw, None), ("pw_vits2_vocos_v1", "vits2vocos", is_pw, None),
           # uint8: bench.py counts one launch per Conv node = its qconv_i8_kernel (the quantise / range kernels'
           # bytes are charged to that node)
           ("mrf_uint8", "uint8", is_u8, "qconv_i8_kernel"))
# digest of the library sources the passes ran on (tools/gpu_round.sh writes it on the GPU box)
lib_digest = open(f"{src}/lib_digest.txt").read().strip() if os.path.exists(f"{src}/lib_digest.txt") else None
out["lib_digest"] = lib_digest
for key, sub, in_class, count_only in CLASSES:
    f = agg(f"{src}/pmc_fetch_{sub}/r_counter_collection.csv", "FETCH_SIZE")
    w = agg(f"{src}/pmc_write_{sub}/r_counter_collection.csv", "WRITE_SIZE")
    sp = f"{src}/prof_stats_{sub}/r_kernel_stats.csv"
    stats = {r["Name"]: r for r in csv.DictReader(open(sp))} if os.path.exists(sp) else {}
    dom = {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0}
    kernels = {}
    for k in sorted(f, key=lambda k: -sum(f[k])):
        n = len(f[k])
        ent = {"launches_in_pmc_run": n, "fetch_size_kb_per_launch": sum(f[k]) / n,
               "write_size_kb_per_launch": sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))}
        ent["hbm_bytes_per_launch"] = (2 * ent["fetch_size_kb_per_launch"] + ent["write_size_kb_per_launch"]) * 1024
        if k in stats:
            ent["avg_duration_ns"] = float(stats[k]["AverageNs"])
            ent["calls_in_stats_run"] = int(stats[k]["Calls"])
        if len(kernels) < 12:
            kernels[k] = ent
        if in_class(k):
            dom["launches"] += n if (count_only is None or count_only in k) else 0
            dom["fetch_kb"] += sum(f[k])
            dom["write_kb"] += sum(w.get(k, [0]))
            if k in stats:
                dom["stat_calls"] = dom.get("stat_calls", 0) + (int(stats[k]["Calls"]) if (count_only is None or count_only in k) else 0)
                dom["stat_ns"] = dom.get("stat_ns", 0.0) + float(stats[k]["TotalDurationNs"])
    out[key] = {
        "launches": dom["launches"],
        "hbm_bytes_per_launch": (2 * dom["fetch_kb"] + dom["write_kb"]) * 1024 / max(1, dom["launches"]),
        "raw_fetch_bytes_per_launch": dom["fetch_kb"] * 1024 / max(1, dom["launches"]),
        "write_bytes_per_launch": dom["write_kb"] * 1024 / max(1, dom["launches"]),
        # rocprofv3 --stats view of the same class (compare with bench.py roofline.avg_launch_ms)
        "rocprof_avg_duration_ms": dom.get("stat_ns", 0.0) / max(1, dom.get("stat_calls", 0)) / 1e6,
        "rocprof_calls": dom.get("stat_calls", 0),
        "lib_digest": lib_digest,
        "kernels": kernels,
    }
    if key == "dominant_conv_mfma":
        # the f32 decoder's default schedule runs a stage's chains on three streams: the per-kernel durations of THAT trace
        # overlap.  The --decoder-serial trace (one stream) gives the class's kernels one at a time:
        sps = f"{src}/prof_stats_baker_serial/r_kernel_stats.csv"
        if os.path.exists(sps):
            rows = [r for r in csv.DictReader(open(sps)) if in_class(r["Name"])]
            calls = sum(int(r["Calls"]) for r in rows)
            out[key]["rocprof_avg_duration_ms_serial"] = sum(float(r["TotalDurationNs"]) for r in rows) / max(1, calls) / 1e6
            out[key]["rocprof_calls_serial"] = calls
            out[key]["note"] = ("rocprof_avg_duration_ms: the default (three-stream) schedule -- overlapping kernels, sums "
                                "to more than the class's wall time; rocprof_avg_duration_ms_serial: --decoder-serial, "
                                "compare with bench.py roofline.isolated_serial.avg_launch_ms")
    if not f:
        del out[key]  # this pass was not run
        continue
    print(key, json.dumps({k: v for k, v in out[key].items() if k != "kernels"}))
json.dump(out, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
