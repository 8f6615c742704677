"""Code-generation guards for the kernels whose speed depends on it (CPU: hipcc cross-compiles gfx950 without a GPU).

The hash kernels are built for five waves per SIMD (`amdgpu_waves_per_eu(5, 5)`: at most 102 VGPRs) and must not spill: a spill
showed up as extra HBM traffic in round 2, and one more VGPR than the budget costs a wave of occupancy on every SIMD.  The
NTT and quotient kernels run at four waves (at most 128 VGPRs) without scratch.  The compiler's own resource remarks are the
evidence (`-Rpass-analysis=kernel-resource-usage`)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def remarks(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("codegen") / "prover.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-o", str(out), os.path.join(ROOT, "plonky2-aes_amd", "csrc", "prover_gpu.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    info = {}
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = info.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return info


def _kernel(info, fragment):
    names = [n for n in info if fragment in n]
    assert names, fragment
    return [info[n] for n in names]


@pytest.mark.parametrize("fragment", ["k_hash_leaves", "k_hash_fri_leaves", "k_merkle_level"])
def test_hash_kernels_hold_five_waves_without_scratch(remarks, fragment):
    for k in _kernel(remarks, fragment):
        assert k["ScratchSize"] == 0, k
        assert k["VGPRs"] + k.get("AGPRs", 0) <= 102 and k["Occupancy"] >= 5, k


@pytest.mark.parametrize("fragment", ["k_ntt_r16", "k_quotient", "k_pow", "k_challenger"])
def test_other_hot_kernels_do_not_spill(remarks, fragment):
    for k in _kernel(remarks, fragment):
        assert k["ScratchSize"] == 0, k
        assert k["VGPRs"] + k.get("AGPRs", 0) <= 128 and k["Occupancy"] >= 4, k
