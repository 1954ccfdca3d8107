#!/usr/bin/env python3
"""One table from the three counter passes of tools/gpu_sq_counters.sh (gpurun_out/pmc_sq_{a,b,c}_<tag>): per kernel SYMBOL
(dispatches of every grid size summed) the matrix-pipe busy share, the wave-cycle split and the instruction mix per MFMA.
Normalisation as in profiles/r05_sq_counters_mrf.txt: SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs,
GRBM_GUI_ACTIVE over the 8 XCDs -> busy share = MFMA_BUSY / (1024 x GUI_ACTIVE / 8).
usage: sq_table.py <tag> [regex on the kernel name]"""
import collections
import csv
import glob
import re
import sys

tag = sys.argv[1]
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
tot = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for ps in "abc":
    for f in glob.glob(f"gpurun_out/pmc_sq_{ps}_{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void wetts::", "").replace("wetts::", "")
            if rx and not rx.search(name):
                continue
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            if ps == "a":
                ndisp[name].add(row.get("Dispatch_Id", len(ndisp[name])))
print(f"{'kernel':58s} {'disp':>5s} {'MFMA busy':>9s}  {'VALU':>5s} {'SALU':>5s} {'LDS':>5s} {'VMEM':>5s} per MFMA   "
      "wave cycles: wait_any / wait_inst / active, LDS stall")
for name, m in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui / 8) if gui else float("nan")
    mf = m.get("SQ_INSTS_MFMA", 0) or float("nan")
    wc = m.get("SQ_WAVE_CYCLES", 0) or float("nan")
    print(f"{name[:58]:58s} {len(ndisp[name]):5d} {busy:9.3f}  {m.get('SQ_INSTS_VALU', 0) / mf:5.1f} "
          f"{m.get('SQ_INSTS_SALU', 0) / mf:5.1f} {m.get('SQ_INSTS_LDS', 0) / mf:5.2f} "
          f"{(m.get('SQ_INSTS_VMEM_RD', 0) + m.get('SQ_INSTS_VMEM_WR', 0)) / mf:5.2f}            "
          f"{m.get('SQ_WAIT_ANY', 0) / wc:.3f} / {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} / "
          f"{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}, {m.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}")
