#!/usr/bin/env python3
"""Socket power and shader clock while a workload runs (GPU box only): is a kernel class bound by the power cap?

Samples the amdgpu hwmon files of the first GPU every 10 ms in this process (power1_average / power1_input in uW,
freq1_input = sclk in Hz, power1_cap) while each workload runs as a child process, and reports, over the samples
taken while the GPU was busy (power above half of the window's maximum): mean / max power and mean shader clock.
Falls back to `rocm-smi --showpower --showclocks --json` (a few samples per second) without hwmon files."""
import glob, json, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_hwmon():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for pw in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(d, pw)) and os.path.exists(os.path.join(d, "freq1_input")):
                return d, pw
    return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.d, self.pw = find_hwmon()
        self.samples = []
        self.stop = False

    def read(self):
        if self.d:
            try:
                w = int(open(os.path.join(self.d, self.pw)).read()) / 1e6
                f = int(open(os.path.join(self.d, "freq1_input")).read()) / 1e6
                return w, f
            except (OSError, ValueError):
                return None
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                 timeout=5).stdout
            card = next(iter(json.loads(out).values()))
            w = next(float(v) for k, v in card.items() if "ower" in k and "(W)" in k)
            f = next(float(str(v).strip("()Mhz")) for k, v in card.items() if k.startswith("sclk"))
            return w, f
        except Exception:
            return None

    def run(self):
        while not self.stop:
            r = self.read()
            if r:
                self.samples.append((time.time(),) + r)
            time.sleep(0.01 if self.d else 0.05)


def main():
    s = Sampler()
    cap = None
    if s.d and os.path.exists(os.path.join(s.d, "power1_cap")):
        cap = int(open(os.path.join(s.d, "power1_cap")).read()) / 1e6
    print(f"source: {s.d or 'rocm-smi'}  power cap: {cap} W", flush=True)
    s.start()
    py = sys.executable
    pair = {"WETTS_BENCH_ITERS": "1500", "WETTS_PAIR": "1", "WETTS_CONV_FLAGS": "16"}
    work = [
        ("idle", ["sleep", "1.5"], {}),
        ("bf16 decoder step x250 (bench.py --decoder-dtype bf16)",
         [py, "bench.py", "--decoder-dtype", "bf16", "--steps", "250", "--warmup", "5", "--no-cpu-baseline"], {}),
        ("f32 headline step x50 (bench.py)", [py, "bench.py", "--steps", "50", "--warmup", "3", "--no-cpu-baseline"], {}),
        ("stress48k f16 step x200", [py, "bench.py", "--config", "stress48k", "--steps", "200", "--warmup", "5",
                                    "--no-cpu-baseline"], {}),
        ("pair16 C=128 k=11, d=1,3,5 x1500 launches each", [py, "tools/bench_conv.py", "16"], dict(pair, WETTS_SHAPES="128:11")),
        ("pair16 C=32 k=3, d=1,3,5 x1500 launches each", [py, "tools/bench_conv.py", "16"], dict(pair, WETTS_SHAPES="32:3")),
    ]
    for name, cmd, env in work:
        t0 = time.time()
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True)
        t1 = time.time()
        win = [x for x in s.samples if t0 <= x[0] <= t1]
        if not win:
            print(f"{name}: no samples")
            continue
        pmax = max(x[1] for x in win)
        busy = [x for x in win if x[1] >= 0.5 * pmax] if name != "idle" else win
        pm = sum(x[1] for x in busy) / len(busy)
        fm = sum(x[2] for x in busy) / len(busy)
        tail = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("C=") or ln.startswith("{")]
        extra = ""
        if tail and tail[-1].startswith("{"):
            try:
                d = json.loads(tail[-1])
                extra = f"  | {d['ms_per_step']:.2f} ms/step, roofline.frac {d['roofline']['frac']:.3f}"
            except Exception:
                pass
        elif tail:
            extra = "  | " + " ; ".join(t[:60].strip() for t in tail[-3:])
        print(f"{name}: {len(busy)} busy samples over {t1 - t0:.1f} s: power mean {pm:.0f} W, max {pmax:.0f} W"
              f"{f' (cap {cap:.0f})' if cap else ''}, shader clock mean {fm:.0f} MHz{extra}", flush=True)
    s.stop = True


if __name__ == "__main__":
    main()
