"""Register / scratch / occupancy table of every kernel of a HIP source, and the diff of two such tables.

    python tools/kernel_resources.py wetts_amd/csrc/conv_mfma.hip > new.txt      # compiles with -Rpass-analysis
    python tools/kernel_resources.py --diff old.txt new.txt

A change to a shared epilogue can move every instantiation of a kernel template over a register granule (round 4: a
tanhf / expf branch in the generic tail of conv_mfma_body made the MRF kernels spill, -14 % on the headline) -- run
this before spending GPU time on an A/B."""
import re, subprocess, sys


def parse(text):
    out, cur = {}, None
    for l in text.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = cur.split("(")[0]
            out[cur] = {}
        for k in ("SGPRs:", "VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "LDS Size"):
            if cur and k in l:
                mm = re.findall(r"(\d+)", l.split(k)[-1])
                if mm:
                    out[cur][k.rstrip(":").replace(" ", "")] = int(mm[0])
    return out


def table(d):
    return "\n".join(f"{k}\t" + " ".join(f"{a}={b}" for a, b in v.items()) for k, v in sorted(d.items()))


def untable(text):
    d = {}
    for l in text.splitlines():
        if "\t" in l:
            k, v = l.split("\t")
            d[k] = dict((a.split("=")[0], int(a.split("=")[1])) for a in v.split())
    return d


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        a, b = untable(open(sys.argv[2]).read()), untable(open(sys.argv[3]).read())
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                print(k, "\n   old", a.get(k), "\n   new", b.get(k))
    elif sys.argv[1] == "--parse":
        print(table(parse(open(sys.argv[2]).read())))
    else:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", sys.argv[1],
                            "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        print(table(parse(r.stderr)))
