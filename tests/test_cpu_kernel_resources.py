"""Register / scratch budget of the built kernels, read from the code objects inside libwetts_hip.so (no GPU needed).

Why this is a test: the hot kernels are built around an occupancy (4 waves per SIMD = at most 128 VGPRs, 3 = 168) and the
compiler silently trades it away -- in round 4 a tanhf / expf branch added to an epilogue tail that every instantiation of
conv_mfma_body shares pushed the MRF kernels to 167 VGPRs + 540 bytes of scratch and cost 14 % of the headline, with every
parity test still green (profiles/r04_wn_gate_ab.txt).  tools/kernel_resources.py prints the same table for a diff."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402
from wetts_amd import _lib  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(kernel_resources.READELF) or shutil.which("c++filt") is None or
                                not os.path.exists(_lib.LIB_PATH),
                                reason="needs llvm-readelf, c++filt and a built wetts_amd/lib/libwetts_hip.so "
                                       "(the library is git-ignored: run __graft_entry__.build() first)")

# The budgets below are properties of (sources, compiler): they were measured with this hipcc.  Another compiler may
# allocate registers differently without anything being wrong in the sources, so a budget miss under a different
# compiler is reported as an expected failure (re-measure with tools/kernel_resources.py and update both).
BUDGETS_MEASURED_WITH = "HIP version: 7.2"


def _compiler_matches():
    import subprocess
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return False
    return BUDGETS_MEASURED_WITH in out


@pytest.fixture(autouse=True)
def _other_compiler_is_xfail(request):
    if not _compiler_matches():
        request.node.add_marker(pytest.mark.xfail(reason="register budgets were measured with " + BUDGETS_MEASURED_WITH,
                                                  strict=False))

# kernels that are allowed to touch scratch, with a bound in bytes per lane: spills outside their MFMA loops (checked in
# the ISA when they were admitted), or an indexed local array
SCRATCH_ALLOWED = {
    "wetts::spline_inverse_kernel": 128,              # bin tables indexed at run time
}


@pytest.fixture(scope="module")
def table():
    t = kernel_resources.library_table(_lib.LIB_PATH)
    assert len(t) > 200, "kernel metadata not found in the library"
    return t


def test_no_unexpected_scratch(table):
    bad = {k: v["ScratchSize"] for k, v in table.items() if v.get("ScratchSize", 0) > SCRATCH_ALLOWED.get(k, 0)}
    assert not bad, f"kernels spilling to scratch: {bad}"


def test_mrf_conv_kernels_keep_four_waves_per_simd(table):
    """128-row x 128-column tiles (C >= 128 stages: 45 % of the headline step) at <= 128 VGPRs; the 64-row x 256-column tiles
    (C = 64) at <= 168 (three waves: their LDS tile allows no more anyway)."""
    names = [k for k in table if "conv_mfma_kernel<1, 4, 4, 1," in k or "conv_mfma_group_kernel<1, 4, 4, 1," in k]
    assert len(names) >= 10
    for k in names:
        assert table[k]["VGPRs"] <= 128, (k, table[k])  # (vgpr_count of the metadata = VGPRs + AGPRs, the unified file)
    names = [k for k in table if "conv_mfma_kernel<1, 4, 2, 2," in k or "conv_mfma_group_kernel<1, 4, 2, 2," in k]
    assert len(names) >= 10
    for k in names:
        assert table[k]["VGPRs"] <= 168, (k, table[k])


def test_pointwise_gemm_and_16bit_kernels_keep_their_occupancy(table):
    for k, v in table.items():
        if "pw_gemm_kernel<" in k:  # amdgpu_waves_per_eu(4, 4)
            assert v["VGPRs"] <= 128, (k, v)
        # three waves per SIMD (the 128 x 128-tile, 64-channel-chunk variant <4, 2, 2, 64> runs at two)
        if "resblock_pair16_kernel<" in k or ("conv_bf16_kernel<" in k and "<4, 2, 2, 64," not in k):
            assert v["VGPRs"] <= 168, (k, v)
        if "resblock_chain32_kernel<" in k:  # two blocks per CU (LDS); 256 registers and spills before the four A sets
            assert v["VGPRs"] <= 192 and v["ScratchSize"] == 0, (k, v)
        if "rb2_stage16_kernel<" in k:  # two blocks of four waves per CU
            assert v["VGPRs"] <= 256 and v["ScratchSize"] == 0, (k, v)
        # the uint8 conv: its waves wait on memory, and the third one per SIMD was worth 12 % of the uint8 step (round 6,
        # profiles/r06_ab_uint8.txt: 188 -> 166 registers)
        if "qconv_i8_kernel<" in k:
            assert v["VGPRs"] <= 168 and v["ScratchSize"] == 0, (k, v)


def test_no_serialised_load_round_trips_in_the_kernels_fixed_for_them():
    """tools/isa_scan.py: a run of `load ... s_waitcnt vmcnt(0)` pairs = every load waits for memory before the next is
    issued.  Round 5 found that shape in LayerNorm (`if (add) v[j] += add[..]` inside the load loop: PER round trips per
    launch, 8.5 -> 5.9 us once batched) and in conv_mfma_kernel's accumulator init (1.7 % of the headline); LayerNorm's ISA
    is simple enough to pin: no run longer than 3 pairs.  (conv_mfma_kernel keeps runs of 64 in its EDGE-tile paths by
    design, so its fix is pinned by the SQ-counter profile instead, profiles/r05_sq_counters_mrf.txt.)"""
    import isa_scan
    if not os.path.exists(isa_scan.OBJDUMP):
        pytest.skip("needs llvm-objdump")
    runs = isa_scan.scan(_lib.LIB_PATH)
    ln = {k: v for k, v in runs.items() if "layernorm_reg_kernel<" in k}
    assert len(ln) >= 3, sorted(runs)[:5]
    for k, (run, pairs) in ln.items():
        assert run <= 3, (k, run, pairs)
