"""B = 1 latency anatomy: run under `rocprofv3 --kernel-trace` (see tools/gpu_b1.sh), then
`python tools/trace_b1.py --summarize <dir>` groups the kernel trace into calls (separated by host
sleeps) and prints, per call: launches, summed kernel time, span, idle share and the top kernels."""
import argparse, csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args):
    import numpy as np, torch
    from wetts_amd import SynthesizerTrn, checkpoint, config, synth
    from wetts_amd.session import DecoderSession, EncoderSession, get_chunks
    dev = torch.device("cuda:0")
    net = SynthesizerTrn(256, 513, 32, n_speakers=1, **config.MODEL_CONFIGS[args.model]).to(dev)
    sd = synth.make_state_dict(net.cfg, seed=0)
    net.load_blob(checkpoint.pack_blob(net.cfg, sd).to(dev))
    ids = np.random.RandomState(0).randint(0, 256, (1, args.phonemes))
    feeds = {"input": ids, "input_lengths": np.array([args.phonemes], dtype=np.int64),
             "scales": np.array([[0.667, 1.0, 0.8]], dtype=np.float32), "sid": np.array([0], dtype=np.int64)}
    enc = EncoderSession(net)
    dec = DecoderSession(net, use_graph=bool(args.graph))
    z = enc.run(None, feeds)[0]
    w0 = get_chunks(z.shape[1], 40, 10)[0]
    for _ in range(3):
        enc.run(None, feeds); dec.run(None, {"z": z[:, w0[0]:w0[1]], "sid": feeds["sid"]})
    torch.cuda.synchronize(); time.sleep(0.3)
    for _ in range(args.reps):
        t0 = time.perf_counter(); enc.run(None, feeds); torch.cuda.synchronize()
        print("enc_ms", (time.perf_counter() - t0) * 1e3, flush=True); time.sleep(0.05)
    time.sleep(0.3)
    for _ in range(args.reps):
        t0 = time.perf_counter(); dec.run(None, {"z": z[:, w0[0]:w0[1]], "sid": feeds["sid"]})
        torch.cuda.synchronize(); print("dec_ms", (time.perf_counter() - t0) * 1e3, flush=True); time.sleep(0.05)


def summarize(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    groups, cur = [], []
    for r in rows:
        if cur and r[0] - cur[-1][1] > 20_000_000:
            groups.append(cur); cur = []
        cur.append(r)
    if cur: groups.append(cur)
    for gi, g in enumerate(groups):
        busy = sum(e - s for s, e, _ in g); span = g[-1][1] - g[0][0]
        if len(g) < 20: continue
        print(f"call {gi}: {len(g)} launches, kernel time {busy/1e6:.3f} ms, span {span/1e6:.3f} ms, "
              f"idle {100*(1-busy/max(span,1)):.0f} %, mean kernel {busy/len(g)/1e3:.1f} us, "
              f"mean gap {(span-busy)/max(len(g)-1,1)/1e3:.1f} us")
    # kernel table for the last encoder-like call and last decoder-like call
    big = [g for g in groups if len(g) >= 20]
    for g in (big[len(big)//2 - 1], big[-1]) if len(big) >= 2 else big:
        agg = {}
        for s, e, n in g:
            n = n.split("(")[0][-70:]
            a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
        print(f"--- call with {len(g)} launches")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"{t/1e3:9.1f} us {c:4d} x {t/c/1e3:7.1f} us  {n}")
        if os.environ.get("B1_SEQ"):
            print("    sequence (us):", " ".join(f"{n.split('(')[0].split('::')[-1][:14]}:{(e-s)/1e3:.0f}" for s, e, n in g))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarize"); ap.add_argument("--model", default="v1")
    ap.add_argument("--phonemes", type=int, default=64); ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--graph", type=int, default=0)
    a = ap.parse_args()
    summarize(a.summarize) if a.summarize else run(a)
