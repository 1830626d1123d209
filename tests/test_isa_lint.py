"""Build-time lint of the attention kernels' ISA (no GPU needed: hipcc cross-compiles gfx950 here).

Round 5 found two ways in which the compiler silently serialises a hand-pipelined key loop (`attn2_kernel` row):

* with the LDS-DMA *builtin* it tracks each transfer as a store to LDS and puts ``s_waitcnt vmcnt(0)`` in front of the next LDS read --
  a wait for the tile just requested, every tile; the kernels now issue the transfers as inline asm (``dma16``, amx_common.h);
* once the transfers are invisible to it, its wait for the Q-fragment loads moves to their first use INSIDE the loop, where
  ``vmcnt(0)`` again waits for every tile in flight; a visible use of the fragments in front of the loop keeps that wait outside.

Both are invisible in the source and in the results (the output is bitwise the same either way), so the check is on the ISA: inside the
key loops of ``attn_kernel`` / ``attn2_kernel`` every ``s_waitcnt vmcnt`` must be one of the hand-placed ones (inside an ASM block).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def attention_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "amx_attention.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/allophant_amd/csrc",
           "--cuda-device-only", "-S", "-o", str(out), f"{ROOT}/allophant_amd/csrc/amx_attention.hip"]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    return out.read_text().splitlines()


def _kernels(lines):
    """(name, body lines) of every attention kernel in the listing."""
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN3amx\S*attn2?_kernel\S*):", lines[i])
        if m:
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            yield m.group(1), lines[i:end]
            i = end
        i += 1


def test_no_compiler_inserted_vmcnt_wait_inside_the_key_loops(attention_isa):
    found, checked = [], 0
    for name, body in _kernels(attention_isa):
        checked += 1
        # attn_kernel: the key loop is the only loop (depth 1); attn2_kernel: the item loop is depth 1, the key loop depth 2
        loop_depth = "Depth=2" if "attn2_kernel" in name else "Depth=1"
        in_asm, label = False, ""
        for line in body:
            if line.startswith(".LBB"):
                label = line
            if "#ASMSTART" in line:
                in_asm = True
            if "#ASMEND" in line:
                in_asm = False
            if "vmcnt" in line and not in_asm and loop_depth in label:
                found.append((name, label.split(":")[0], line.strip()))
    assert checked >= 16, f"only {checked} attention kernels found in the listing"
    assert not found, "compiler-inserted vmcnt waits inside a key loop:\n" + "\n".join(map(str, found[:10]))


def test_key_loops_hold_no_scratch_traffic_and_the_dma_is_inline_asm(attention_isa):
    for name, body in _kernels(attention_isa):
        loop_depth = "Depth=2" if "attn2_kernel" in name else "Depth=1"
        in_asm, label = False, ""
        for line in body:
            if line.startswith(".LBB"):
                label = line
            if "#ASMSTART" in line:
                in_asm = True
            if "#ASMEND" in line:
                in_asm = False
            text = line.strip()
            if text.startswith("scratch_") and loop_depth in label:
                raise AssertionError(f"{name}: spill traffic inside the key loop: {text}")
            if "offen lds" in text:
                assert in_asm, f"{name}: LDS-DMA issued through the builtin, not dma16: {text}"


@pytest.fixture(scope="module")
def gemm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_gemm") / "amx_gemm.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/allophant_amd/csrc",
           "--cuda-device-only", "-S", "-o", str(out), f"{ROOT}/allophant_amd/csrc/amx_gemm.hip"]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    return out.read_text().splitlines()


def test_ping_pong_gemm_k_loop_has_only_hand_placed_waits_and_no_spills(gemm_isa):
    """`gemm_pp_kernel` (70 % of the step): inside the K loop (depth 2 of the persistent tile loop) every `s_waitcnt vmcnt` is one
    of the ring's counted waits (inline asm), nothing is spilled, and the loop holds MFMAs at all (the right blocks are looked at)."""
    lines, checked, i = gemm_isa, 0, 0
    while i < len(lines):
        m = re.match(r"^(_ZN3amx\S*gemm_pp_kernel\S*):", lines[i])
        if m:
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            in_asm, label, mfma_in_loop = False, "", 0
            for line in lines[i:end]:
                if line.startswith(".LBB"):
                    label = line
                if "#ASMSTART" in line:
                    in_asm = True
                if "#ASMEND" in line:
                    in_asm = False
                text = line.strip()
                if "Depth=2" in label:
                    mfma_in_loop += text.startswith("v_mfma")
                    assert not ("vmcnt" in text and not in_asm), f"{m.group(1)}: compiler-inserted wait in the K loop: {text}"
                    assert not text.startswith("scratch_"), f"{m.group(1)}: spill traffic in the K loop: {text}"
            assert mfma_in_loop >= 48, f"{m.group(1)}: K loop not found ({mfma_in_loop} MFMAs at depth 2)"
            checked += 1
            i = end
        i += 1
    assert checked >= 8, f"only {checked} gemm_pp_kernel instances found"
