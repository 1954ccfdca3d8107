"""CPU tier: bench.py's N = 2 control flow end to end (its own rank spawn, gloo process group, weight broadcast and
its verification, batching plan, timed loop, max / sum reductions, per-rank times, the JSON line) against the
device-free stub backend of tests/bench_stub.py -- argument, sharding and reduce bugs surface without a GPU."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, args=(), gpus=2, batch=8):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(WETTS_BENCH_TEST_BACKEND="tests.bench_stub:StubBackend", WETTS_DIST_BACKEND="gloo",
               PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--config", "aishell3", "--model", "tiny", "--presteps-s", "0.02", "--no-cpu-baseline"] + \
        (["--batch", str(batch)] if batch else []) + list(args)
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)


def _diag(path):
    return [json.loads(l) for l in open(path).read().splitlines() if l.strip()]


def test_bench_two_ranks_control_flow_and_reductions(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    from tests import bench_stub
    from wetts_amd import batching
    diag = tmp_path / "scale_diag.json"
    p = _run({"WETTS_SCALE_DIAG": str(diag)})
    assert p.returncode == 0, p.stderr[-3000:]
    # gpurun_out/scale_diag.json (here: the tmp path): every rank's start-up line and rank 0's result, one JSON line each
    ev = _diag(diag)
    assert sorted(e["rank"] for e in ev if e["event"] == "init_process_group") == [0, 1]
    assert all("device" in e and e["world"] == 2 for e in ev if e["event"] == "init_process_group")
    res = [e for e in ev if e["event"] == "result"]
    assert len(res) == 1 and res[0]["rank"] == 0 and res[0]["n_gpus"] == 2 and len(res[0]["rank_ms"]) == 2
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly ONE JSON line, from rank 0
    # ... and nothing else on stdout: the banners Gloo prints from C++ ("[Gloo] Rank 0 is connected ...") go to stderr
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, p.stdout
    d = json.loads(lines[0])
    assert d["backend_label"].startswith("STUB")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["blob_checksum_ok"] is True
    assert len(d["rank_ms"]) == 2 and all(t > 0 for t in d["rank_ms"]) and d["imbalance"] >= 0
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "samples/s"
    assert "cpu_baseline" not in d
    # work = SUM over ranks: the same global utterance list the ranks drew, through the stub's duration rule
    x, lens, sid = bench.make_inputs("tiny", 256, 218, 16, 128, True)
    want = bench_stub.expected_frames(lens.tolist(), 0.92)
    assert abs(d["config"]["valid_frames_per_step"] - want) < 1e-6
    assert d["config"]["global_batch"] == 16
    # time = MAX over ranks, value = all ranks' samples / that time
    assert d["ms_per_step"] >= max(d["rank_ms"]) - 1e-6
    hop = d["config"]["hop"]
    assert abs(d["value"] - want * hop / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the plan in the line is the plan of rank 0's shard
    assert d["config"]["decode"] == "ragged"  # the aishell3 preset's default
    pl = batching.plan(lens.tolist(), 2, max_pad_frac=0.08, ragged=True)
    assert d["config"]["sub_batch_plan"]["sizes_rank0"] == [len(b) for b in pl.buckets[0]]
    assert d["config"]["padded_sub_batches_per_step"] == len(pl.buckets[0])
    # ragged decode: only the masked stages pay for padding (batching.RAGGED_PAD_WEIGHT of a slot), and it is
    # that weighted share the plan bounds
    f = d["config"]["sub_batch_plan"]["phoneme_pad_frac"]
    w = batching.RAGGED_PAD_WEIGHT
    assert w * f / ((1 - f) + w * f) <= 0.08 + 1e-9
    assert d["roofline"]["launches"] > 0 and d["roofline"]["frac"] > 0


def _fracs(node, path=""):
    """Every value stored under a key named `frac`, anywhere in the line."""
    if isinstance(node, dict):
        for k, v in node.items():
            if k == "frac":
                yield path + "/frac", v
            else:
                yield from _fracs(v, path + "/" + k)


def test_roofline_fractions_are_fractions_of_what_the_launches_move():
    """Round 4 printed roofline.frac 0.98 / isolated.frac 1.08 for the fused 16-bit class: SURVEY 8(d)'s per-conv bytes
    divided by the time of kernels that fuse a whole ResBlock stage.  The stub models exactly that class (per-conv
    accounting 1.2x the HBM peak, launched bytes a quarter of it): every `frac` of the line must be a fraction of the
    bytes the launches move (<= 1), the per-conv rate must survive only as the labelled side view."""
    for extra in (["--config", "multilingual", "--model", "v3"], ["--decoder-dtype", "bf16"], ["--decoder-dtype", "uint8"], []):
        env = dict(os.environ)
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        env.update(WETTS_BENCH_TEST_BACKEND="tests.bench_stub:StubBackend", PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--phonemes", "16",
               "--presteps-s", "0.02", "--no-cpu-baseline"] + extra
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        r = d["roofline"]
        fr = dict(_fracs(r))
        assert "/frac" in fr and fr["/frac"] > 0
        if r["bound"] == "mfma":
            assert abs(fr["/frac"] - 0.5) < 1e-6
        if r["bound"] == "hbm" and "uint8" not in extra:
            assert abs(fr["/frac"] - 0.3) < 1e-6 and abs(r["isolated"]["frac"] - 0.3) < 1e-6, (extra, fr)
            assert "traffic_head" in r and "traffic_current" in r
        if "perconv_view" in r:  # the accounting rate is there, labelled, and is not called a fraction
            assert abs(r["perconv_view"]["ratio_to_hbm_peak"] - 1.2) < 1e-6 and "frac" not in r["perconv_view"]
        bad = {k: v for k, v in fr.items() if v is not None and not (0 <= v <= 1.0)}
        assert not bad, (extra, bad)


def test_bench_eight_ranks_aishell3_preset_plan_and_imbalance():
    """BASELINE.json configs[3] as the driver will launch it on an 8-GPU node -- `bench.py --gpus 8 --config aishell3`:
    512 ragged utterances (64 per rank), LPT deal, per-rank ragged plans -- through the same spawn / gloo / broadcast /
    reduce path, on the stub backend: all 8 ranks seen, every utterance counted once, the padded-slot imbalance of the
    plan and the per-rank times in the line."""
    sys.path.insert(0, ROOT)
    import bench
    from tests import bench_stub
    from wetts_amd import batching
    p = _run(gpus=8, batch=0)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["blob_checksum_ok"] is True and len(d["rank_ms"]) == 8
    assert d["config"]["global_batch"] == 512 and d["scaling"] == "weak"
    x, lens, sid = bench.make_inputs("tiny", 256, 218, 512, 128, True)
    assert abs(d["config"]["valid_frames_per_step"] - bench_stub.expected_frames(lens.tolist(), 0.92)) < 1e-6
    pl = batching.plan(lens.tolist(), 8, max_pad_frac=0.08, ragged=True)
    assert sorted(i for sh in pl.shards for i in sh) == list(range(512)) and all(len(sh) == 64 for sh in pl.shards)
    assert abs(d["plan_slot_imbalance"] - pl.stats["imbalance"]) < 1e-9 and pl.stats["imbalance"] < 0.05
    assert d["config"]["sub_batch_plan"]["sizes_rank0"] == [len(b) for b in pl.buckets[0]]
    assert "not a BASELINE.json config" in d["config"]["workload"]  # --model tiny is an override and is named as one


def _partial(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["partial"] is True and d["value"] is None
    return d


def test_bench_eight_ranks_one_hangs_after_the_rendezvous_partial_line_not_a_hang():
    """A rank whose device hangs after start-up: the other seven reach the barrier in front of the timed region and
    would wait for c10d's watchdog.  The phase monitor's deadline fires first: rank 0 prints ONE partial line
    (`ranks_seen` 7 of 8 from the roll call on the rendezvous store, the phase, no value), every waiting rank exits
    with code 4, and the launcher comes back in seconds."""
    import time
    t0 = time.time()
    diag = os.path.join(ROOT, "gpurun_out", "scale_diag_test_hang.json")
    if os.path.exists(diag):
        os.remove(diag)
    p = _run({"WETTS_STUB_HANG_RANK": "5", "WETTS_STUB_HANG_AT": "load", "WETTS_BENCH_PHASE_DEADLINE_S": "15",
              "WETTS_DIST_TIMEOUT_S": "60", "WETTS_SCALE_DIAG": diag}, gpus=8, batch=2)
    assert p.returncode != 0
    d = _partial(p)
    ev = _diag(diag)  # the post-mortem file: 8 ranks bound a device, rank 0 said which phase failed and who was there
    assert sorted(e["rank"] for e in ev if e["event"] == "init_process_group") == list(range(8))
    failed = [e for e in ev if e["event"] == "phase_failed" and e["rank"] == 0]
    assert failed and failed[0]["phase"] == "timed" and failed[0]["ranks_seen"] == 7
    assert [e for e in ev if e["event"] == "partial_line"]
    os.remove(diag)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 7 and d["failed_phase"] == "timed"
    assert "deadline" in d["error"] or "SIGTERM" in d["error"]
    assert time.time() - t0 < 180  # (the stub's rank would sleep for an hour; c10d's own default timeout is 10+ minutes)
    assert "ranks that reached it: 7 of 8" in p.stderr and "[wetts rank 0/8] gloo on" in p.stderr


def test_bench_two_ranks_one_never_reaches_the_rendezvous():
    p = _run({"WETTS_STUB_HANG_RANK": "1", "WETTS_STUB_HANG_AT": "device", "WETTS_BENCH_PHASE_DEADLINE_S": "15",
              "WETTS_DIST_TIMEOUT_S": "60"})
    assert p.returncode != 0
    d = _partial(p)
    assert d["ranks_seen"] == 1 and d["failed_phase"] == "rendezvous"


def test_bench_two_ranks_one_crashes_partial_line_on_the_launchers_sigterm():
    """A rank that dies: torchrun sends SIGTERM to the survivors, rank 0 -- blocked inside a collective -- still
    prints the partial line (the monitor thread reads the signal from the wake-up fd)."""
    p = _run({"WETTS_STUB_CRASH_RANK": "1", "WETTS_DIST_TIMEOUT_S": "60"})
    assert p.returncode != 0
    d = _partial(p)
    assert d["ranks_seen"] <= 2 and ("SIGTERM" in d["error"] or "Error" in d["error"] or "deadline" in d["error"])
    assert "this rank dies while loading its weights" in p.stderr


def test_scale_line_names_the_same_workload_as_the_single_gpu_line():
    """The driver computes scaling efficiency from its N = 1 / 2 / 4 / 8 runs of `bench.py --gpus N` and checks the N = 1
    one against BENCH: weak scaling, per-GPU work fixed, so `config.workload` -- which names the preset, the model, the
    per-GPU batch, the precision and the BASELINE.json label -- must be the SAME string at every N for the default
    invocation (only `global_batch` and `parallelism` carry N), and `value_excludes` says what `value` leaves out."""
    lines = {}
    for n in (1, 2):
        env = dict(os.environ)
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        env.update(WETTS_BENCH_TEST_BACKEND="tests.bench_stub:StubBackend", WETTS_DIST_BACKEND="gloo",
                   PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), OMP_NUM_THREADS="1",
                   WETTS_SCALE_DIAG=os.devnull)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
               "--presteps-s", "0.02", "--no-cpu-baseline"]
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines[n] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    a, b = lines[1], lines[2]
    assert a["config"]["workload"] == b["config"]["workload"] and "BASELINE.json configs[1]" in a["config"]["workload"]
    assert "not a BASELINE" not in a["config"]["workload"]
    assert a["metric"] == b["metric"] and a["unit"] == b["unit"] and a["scaling"] == b["scaling"] == "weak"
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and b["config"]["global_batch"] == 2 * a["config"]["global_batch"]
    assert "H2D" in a["value_excludes"] and "pcie_inclusive_pipelined_samples_per_s" in a["value_excludes"]


def test_bench_refuses_a_corrupted_broadcast():
    p = _run({"WETTS_STUB_CORRUPT_RANK": "1"})
    assert p.returncode != 0
    assert "weight blob differs across ranks" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_cpu_baseline_is_n1_only_unless_asked(tmp_path):
    """--cpu-baseline-multi is parsed and reaches the N > 1 branch (the timing itself needs the full-size oracle
    run, which the GPU-box bench does; here only the flag plumbing: without the flag no key, see above)."""
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "2", "--cpu-baseline-multi"]
        a = bench.parse_args()
    finally:
        sys.argv = old
    assert a.cpu_baseline_multi and a.cpu_sample == 1 and abs(a.length_scale - 0.92) < 1e-9


def test_pmc_traffic_class_sums_and_the_stub_line_never_spawns_a_profiler(tmp_path):
    """tools/pmc_traffic.py (shared by tools/summarize_profiles.py and bench.py --live-traffic): HBM bytes per launch of a
    kernel class = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the class's dispatches / its launches, with the uint8
    convention of counting one launch per Conv node; and a line produced on the device-free stub (or under --live-traffic 0)
    carries no `traffic_live`: the refresh is for real single-GPU runs only."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as pt
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    mrf = "void wetts::conv_mfma_kernel<1, 4, 4, 1, 1, true, true>(wetts::ConvParams)"
    other = "void wetts::conv_mfma_kernel<1, 4, 4, 1, 3, false, true>(wetts::ConvParams)"
    for v in (100, 300):
        rows.append(f'"{mrf}",FETCH_SIZE,{v}')
    rows.append(f'"{other}",FETCH_SIZE,999')
    f = tmp_path / "fetch_counter_collection.csv"
    f.write_text("\n".join(rows))
    w = tmp_path / "write_counter_collection.csv"
    w.write_text("\n".join(["Kernel_Name,Counter_Name,Counter_Value", f'"{mrf}",WRITE_SIZE,50', f'"{mrf}",WRITE_SIZE,150']))
    b, n, fb, wb = pt.class_traffic(pt.agg(str(f), "FETCH_SIZE"), pt.agg(str(w), "WRITE_SIZE"), pt.is_mrf)
    assert n == 2 and b == (2 * 400 + 200) * 1024 / 2 and fb == 400 * 1024 / 2 and wb == 200 * 1024 / 2
    assert pt.is_mrf16("void wetts::rb2_stage16_kernel<32, false, 2, 2, 3, true>(x)") and not pt.is_mrf(other)
    assert pt.is_pw("void wetts::pw_gemm_kernel<16, 4, true>(x)") and not pt.is_pw("void wetts::pw_gemm_kernel<16, 4, false>(x)")
    val, why = pt.live(["-c", "pass"], "no-such-class")
    assert val is None and "class" in why or "rocprofv3" in why
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(WETTS_BENCH_TEST_BACKEND="tests.bench_stub:StubBackend", PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--phonemes", "16",
           "--presteps-s", "0.02", "--no-cpu-baseline", "--live-traffic", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "traffic_live" not in d["roofline"]
