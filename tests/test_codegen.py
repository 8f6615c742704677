"""Code-generation guards for the kernels whose speed depends on it (CPU: hipcc cross-compiles gfx950 without a GPU).

The hash kernels are built for five waves per SIMD (`amdgpu_waves_per_eu(5, 5)`: at most 102 VGPRs) and must not spill: a spill
showed up as extra HBM traffic in round 2, and one more VGPR than the budget costs a wave of occupancy on every SIMD.  The
NTT and quotient kernels run at four waves (at most 128 VGPRs) without scratch.  The compiler's own resource remarks are the
evidence (`-Rpass-analysis=kernel-resource-usage`).

The same cross-compile also emits the gfx950 assembly, which tests/isa_lint.py walks for the hazard the field arithmetic of
gl.h manages by hand (two wait states between a VALU write of an SGPR and a VALU read of it), and the asm strings of csrc/ are
checked for scalar-ALU instructions (which clobber SCC behind the compiler's back)."""
import os
import re
import subprocess

import pytest

import isa_lint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def device_build(tmp_path_factory):
    """One cross-compile of the product's device code: (resource remarks per function, assembly text)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("codegen") / "prover.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
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
    return info, open(out).read()


@pytest.fixture(scope="module")
def remarks(device_build):
    return device_build[0]


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


# ---- ISA lint (tests/isa_lint.py)
def test_no_scalar_alu_instruction_inside_an_asm_string():
    """Round 2's GPU memory-access fault: an `s_or_b64` inside an asm string of gl::add clobbered SCC, which the compiler tracks
    only for its own instructions.  Only `s_nop` (wait states) may appear in the asm strings of csrc/."""
    csrc = os.path.join(ROOT, "plonky2-aes_amd", "csrc")
    n_templates = 0
    for f in sorted(os.listdir(csrc)):
        path = os.path.join(csrc, f)
        assert isa_lint.salu_in_asm_strings(path) == [], path
        n_templates += len(isa_lint.asm_templates(open(path).read()))
    assert n_templates >= 30  # the scanner does see the asm statements of gl.h / poseidon_fast.h
    planted = 'u64 f(u64 a) { sg m; asm("s_or_b64 %0, %1, %2\\n\\tv_mov_b32 %3, 0" : "=s"(m) : "s"(a), "s"(a)); return m; }'
    assert [mn for _, mn in isa_lint.salu_in_asm_strings_text(planted)] == ["s_or_b64"]


def test_the_hazard_lint_sees_a_planted_hazard():
    ok = "f:\n\tv_mad_u64_u32 v[0:1], s[2:3], v2, v3, v[4:5]\n\ts_nop 1\n\tv_subb_co_u32_e64 v6, s[4:5], v7, v8, s[2:3]\n\ts_endpgm\n.Lfunc_end0:\n"
    assert isa_lint.sgpr_hazards(ok) == []
    one_short = ok.replace("s_nop 1", "s_nop 0")
    assert {(h[2], h[3]) for h in isa_lint.sgpr_hazards(one_short)} == {("v_subb_co_u32_e64", "s2"), ("v_subb_co_u32_e64", "s3")}
    # across a branch: the writer sits in the predecessor block, the reader at the branch target
    br = ("f:\n\tv_cmp_gt_u32_e32 vcc, v0, v1\n\ts_cbranch_scc1 .LBB0_2\n\tv_mov_b32_e32 v9, v8\n\tv_mov_b32_e32 v9, v8\n.LBB0_2:\n"
          "\tv_cndmask_b32_e32 v0, v1, v2, vcc\n\ts_endpgm\n.Lfunc_end0:\n")
    assert len(isa_lint.sgpr_hazards(br)) == 2
    # a scalar-unit write in between ends the window (gl::add: the compiler's own s_or_b64 of two carry masks)
    salu = "f:\n\tv_mad_u64_u32 v[0:1], s[2:3], v2, v3, v[4:5]\n\ts_or_b64 s[2:3], s[2:3], s[6:7]\n\tv_cndmask_b32_e64 v6, v7, v8, s[2:3]\n\ts_endpgm\n.Lfunc_end0:\n"
    assert isa_lint.sgpr_hazards(salu) == []


def test_emitted_isa_keeps_two_wait_states_between_valu_sgpr_write_and_read(device_build):
    """Every VALU read of an SGPR (carry-in of v_*_co_*, mask of v_cndmask_b32_e64, constant operand) in the gfx950 code object
    sits at least two wait states behind the last VALU write of that SGPR, on every control-flow path -- in particular in the
    kernels built from gl.h's hand-scheduled carry chains."""
    asm = device_build[1]
    funcs = isa_lint.parse_functions(asm)
    for fragment in ("k_hash_leaves", "k_merkle_level", "k_quotient", "k_ntt_r16", "k_hash_fri_leaves", "k_pow", "k_challenger"):
        assert any(fragment in f for f in funcs), fragment
    assert sum(len(b[1]) for blocks in funcs.values() for b in blocks) > 100000  # the parser saw the code, not an empty file
    hazards = isa_lint.sgpr_hazards(asm)
    assert hazards == [], hazards[:20]


def test_ntt_kernels_issue_the_loads_of_a_step_together(device_build):
    """Round 3's largest NTT gain was an ORDER: the compiler had serialised the load stage of the LDE into sixteen dependent
    round trips (load a point, wait, multiply, load the next).  loads_issued() pins all loads of a step in front of its first
    product; this holds the emitted code to it -- 40 of the half-column kernel's 48 loads in one batch (the rest follow behind
    the first points for the register budget), points + coset scales of the first pass of the large transform, the 16 points and
    the twiddles of the whole-column kernel.  And the planted case: a load-wait-load-wait stream counts as batches of one."""
    asm = device_build[1]
    assert isa_lint.longest_load_batch(asm, "k_ntt_r16ILb1") >= 40
    assert isa_lint.longest_load_batch(asm, "k_ntt_pass1_r16") >= 32
    assert isa_lint.longest_load_batch(asm, "k_ntt_r16ILb0") >= 32
    planted = "f:\n" + "".join("\tglobal_load_dwordx2 v[%d:%d], v[0:1], off\n\ts_waitcnt vmcnt(0)\n" % (2 * i + 2, 2 * i + 3) for i in range(8)) + "\ts_endpgm\n.Lfunc_end0:\n"
    assert isa_lint.longest_load_batch(planted, "f") == 1
    pipelined = "g:\n" + "".join("\tglobal_load_dwordx2 v[%d:%d], v[0:1], off\n" % (2 * i + 2, 2 * i + 3) for i in range(8)) + "\ts_waitcnt vmcnt(6)\n\ts_endpgm\n.Lfunc_end1:\n"
    assert isa_lint.longest_load_batch(pipelined, "g") == 8

