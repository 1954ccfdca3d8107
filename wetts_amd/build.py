"""Builds libwetts_hip.so (gfx950) in-tree with hipcc.  No torch, no JIT cache: the .so lands in
wetts_amd/lib/ so it travels with the repo snapshot to the GPU box."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libwetts_hip.so")
SOURCES = ["conv_mfma.hip", "gemm_pw.hip", "conv_small.hip", "conv_bf16.hip", "resblock16.hip", "resblock2_stage16.hip", "wn16.hip", "qconv_u8.hip", "resblock32.hip", "resblock_chain32.hip", "kernels.hip", "dds_fused.hip", "attention.hip", "mas.hip", "model.hip"]
# measurement tooling (tools/bench_*.py): its own library, linked against the product one
BENCH_SOURCES = ["bench_conv.hip"]
BENCH_LIB = os.path.join(LIBDIR, "libwetts_bench.so")
HEADERS = ["bench_abi.h", "qconv_u8.h", "common.h", "kernels.h", "conv_bf16.h", "conv16_dev.h", "resblock32.h", os.path.join("..", "..", "include", "wetts_hip.h"),
           os.path.join("..", "..", "include", "wetts_vits_model.hpp"),
           os.path.join("..", "..", "tests", "native", "vits_model_main.cpp")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + BENCH_SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _obj_digest(src):
    h = hashlib.sha256()
    for f in [src] + [x for x in HEADERS if x.endswith(".h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        if open(stamp).read().strip() == dig:
            return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES + BENCH_SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        # per-object stamp: the source, every header and the flags (an edit to one .hip recompiles one object)
        odig = _obj_digest(src)
        ostamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        if os.path.exists(ostamp):
            os.remove(ostamp)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[wetts_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, ostamp, odig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, ostamp, odig, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode(errors="replace") if not verbose else "")
        else:
            with open(ostamp, "w") as f:
                f.write(odig)
    if failed:
        raise RuntimeError("hipcc failed")
    bench_objs = objs[len(SOURCES):]
    objs = objs[:len(SOURCES)]
    for lib, obs, extra in ((LIB, objs, []),
                            (BENCH_LIB, bench_objs, ["-L", LIBDIR, "-lwetts_hip", "-Wl,-rpath,$ORIGIN"])):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + obs + extra
        if verbose:
            print("[wetts_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # native C++ host used by tests/test_gpu_native.py (twin of the reference's VitsModel class)
    native_src = os.path.join(HERE, "..", "tests", "native", "vits_model_main.cpp")
    if os.path.exists(native_src):
        cmd = [hipcc, "-O2", "-std=c++17", "-x", "c++", "-D__HIP_PLATFORM_AMD__",
               "-I", os.path.join(HERE, "..", "include"), "-I", "/opt/rocm/include", native_src,
               "-L", LIBDIR, "-lwetts_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
               "-Wl,-rpath," + LIBDIR, "-o", os.path.join(LIBDIR, "vits_model_main")]
        if verbose:
            print("[wetts_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
