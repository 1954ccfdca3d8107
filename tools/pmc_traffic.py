#!/usr/bin/env python3
"""HBM traffic of a kernel CLASS from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes (both counters are in KB; FETCH_SIZE reports half of a coalesced stream on gfx950: x2).
Shared by tools/summarize_profiles.py (the committed profiles/rNN_hbm_traffic.json) and by bench.py --live-traffic, which
runs the two counter passes of its OWN command as child processes so that `roofline.traffic` in the line the driver
records belongs to the library that just ran.  Measurement tooling: nothing in wetts_amd/ imports it."""
import collections
import csv
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile


def is_mrf(k):
    """The MRF ResBlock class by kernel symbol: f32 -- conv_mfma_kernel instantiations with the MRF flag, the
    grouped launches, the chain / pair kernels; 16 bit -- the fused pair kernels and the MRF-tagged single convs."""
    mm = re.search(r"conv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, (true|false)", k)
    if mm and mm.group(1) == "true":
        return True
    if "conv_mfma_group_kernel" in k or "resblock_pair32_kernel" in k or "resblock_chain32_kernel" in k:
        return True
    if "resblock_pair16_kernel" in k or "conv16_mb2_kernel" in k or "resblock1_chain16_kernel" in k:
        return True
    mm = re.search(r"conv_bf16_kernel<\d+, \d+, \d+, \d+, (true|false), \d+, (true|false)", k)
    return bool(mm and mm.group(2) == "true")


def is_pw(k):  # the tagged launches: the ConvNeXt GEMMs bench.py times for the Vocos models
    return "pw_gemm_kernel" in k and ", true>" in k


def is_u8(k):
    return "qconv_i8_kernel" in k or "qquantize" in k or "qminmax" in k or "qrange" in k


def is_mrf16(k):
    return is_mrf(k) or "rb2_stage16_kernel" in k


# class key of profiles/rNN_hbm_traffic.json -> (kernel-class predicate, launches counted as: every kernel of the class
# (None) or this kernel-name substring only)
CLASS_OF_KEY = {"dominant_conv_mfma": (is_mrf, None), "mrf16": (is_mrf16, None), "pw": (is_pw, None),
                # uint8: bench.py counts one launch per Conv node = its qconv_i8_kernel (the quantise / range kernels'
                # bytes are charged to that node)
                "mrf_uint8": (is_u8, "qconv_i8_kernel")}


def agg(path, ctr):
    """{kernel name: [counter value per dispatch]} of one counter_collection.csv."""
    d = collections.defaultdict(list)
    if not path or not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def class_traffic(fetch, write, in_class, count_only=None):
    """(HBM bytes per launch of the class, launches, raw fetch bytes per launch, write bytes per launch) from the two
    per-kernel dictionaries of agg()."""
    launches, fkb, wkb = 0, 0.0, 0.0
    for k, v in fetch.items():
        if in_class(k):
            launches += len(v) if (count_only is None or count_only in k) else 0
            fkb += sum(v)
            wkb += sum(write.get(k, [0]))
    n = max(1, launches)
    return (2 * fkb + wkb) * 1024 / n, launches, fkb * 1024 / n, wkb * 1024 / n


def find_csv(root):
    f = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    return f[0] if f else None


def live(argv, key, budget_s=200.0, log=None):
    """Runs `python argv...` twice under rocprofv3 (--pmc FETCH_SIZE, then WRITE_SIZE; separate passes, kernel trace only
    beside --pmc, as the guide prescribes) and returns (bytes per launch, launches) of class `key`, or (None, reason).
    Never raises; bounded by `budget_s` seconds in total."""
    import time
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    if key not in CLASS_OF_KEY:
        return None, f"no class {key!r}"
    in_class, count_only = CLASS_OF_KEY[key]
    t0 = time.time()
    got = {}
    tmp = tempfile.mkdtemp(prefix="wetts_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.time() - t0)
            if left < 20:
                return None, "time budget spent"
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "r", "--",
                   sys.executable] + list(argv)
            env = dict(os.environ, TMPDIR="/tmp")
            # own session: a pass that overruns is killed as a GROUP (rocprofv3 and the python under it), so that no
            # grandchild keeps the GPU busy behind the bench's back
            import signal
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                p.wait(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
                return None, f"{ctr} pass timed out"
            if log:
                log(f"[live traffic] {ctr} pass: exit {p.returncode}, {time.time() - t0:.0f} s")
            csvf = find_csv(out)
            if p.returncode != 0 or not csvf:
                return None, f"{ctr} pass failed (exit {p.returncode})"
            got[ctr] = agg(csvf, ctr)
        b, n, _, _ = class_traffic(got["FETCH_SIZE"], got["WRITE_SIZE"], in_class, count_only)
        if n == 0:
            return None, "no launch of the class in the counter pass"
        return b, n
    except Exception as e:  # measurement aid: never the reason a bench line is lost
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
