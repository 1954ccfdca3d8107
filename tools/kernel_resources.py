"""Register / scratch / occupancy table of every kernel of a HIP source, and the diff of two such tables.

    python tools/kernel_resources.py --lib wetts_amd/lib/libwetts_hip.so > new.txt   # from the built code objects (seconds)
    python tools/kernel_resources.py wetts_amd/csrc/conv_mfma.hip > new.txt          # compiles with -Rpass-analysis
    python tools/kernel_resources.py --diff old.txt new.txt

A change to a shared epilogue can move every instantiation of a kernel template over a register granule (round 4: a
tanhf / expf branch in the generic tail of conv_mfma_body made the MRF kernels spill, -14 % on the headline) -- run
this before spending GPU time on an A/B."""
import re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    """The gfx code objects of the clang offload bundles embedded in a host object / shared library (.hip_fatbin)."""
    d = open(path, "rb").read()
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = d.find(magic, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", d, i + 24)
        off = i + 32
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + ts].decode()
            off += ts
            if "gfx" in triple and es:
                yield triple, d[i + eo:i + eo + es]
        pos = i + 24


def library_table(path):
    """{demangled kernel: {VGPRs, AGPRs, SGPRs, ScratchSize, VGPRSpill, SGPRSpill, LDSSize}} from the AMDGPU metadata
    notes of every code object in `path` -- what the loader will really use, no recompilation."""
    out = {}
    keys = {"vgpr_count": "VGPRs", "agpr_count": "AGPRs", "sgpr_count": "SGPRs", "private_segment_fixed_size": "ScratchSize",
            "vgpr_spill_count": "VGPRSpill", "sgpr_spill_count": "SGPRSpill", "group_segment_fixed_size": "LDSSize"}
    for _, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        # one metadata map per kernel; `.name:` is the kernel symbol (argument entries carry `.name:` too, so anchor on
        # the `.symbol:` line that closes a kernel's map)
        for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
            blk = ".agpr_count:" + blk
            m = re.search(r"\.symbol:\s+(\S+)\.kd", blk)
            if not m:
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            row = {}
            for k, v in keys.items():
                mm = re.search(r"\." + k + r":\s+(\d+)", blk)
                if mm:
                    row[v] = int(mm.group(1))
            out[name] = row
    return out


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
    elif sys.argv[1] == "--lib":
        print(table(library_table(sys.argv[2])))
    elif sys.argv[1] == "--parse":
        print(table(parse(open(sys.argv[2]).read())))
    else:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", sys.argv[1],
                            "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        print(table(parse(r.stderr)))
