#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (counter_collection.csv): mean counter value per kernel
name, plus ratios useful for the MFMA-bound conv (see MI355X_MICROARCH.md §rocprofv3 PMC slots)."""
import csv
import collections
import glob
import sys

root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "?")
        if len(sys.argv) >= 3 and sys.argv[2] == "mrf":  # the f32 MRF class: tagged convs, grouped launches, chain kernels
            import re
            mm = re.search(r"conv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, (true|false)", name)
            if not ((mm and mm.group(1) == "true") or "conv_mfma_group_kernel" in name or "resblock_chain32_kernel" in name):
                continue
        elif len(sys.argv) >= 3 and sys.argv[2].startswith("re:"):  # any class: a regular expression on the kernel name
            import re
            if not re.search(sys.argv[2][3:], name):
                continue
        elif "conv_mfma" not in name and len(sys.argv) < 3:
            continue
        key = (name[:70], row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, ctr in sorted(agg.items()):
    print(key)
    m = {k: sum(v) / len(v) for k, v in ctr.items()}
    for k, v in sorted(m.items()):
        print(f"    {k:32s} {v:16.1f}  (n={len(ctr[k])})")
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in m:
                print(f"    {k}/WAVE_CYCLES = {m[k] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print(f"    MFMA_BUSY/BUSY_CYCLES = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print(f"    MFMA_BUSY/GRBM_GUI_ACTIVE = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['GRBM_GUI_ACTIVE']:.3f}")
