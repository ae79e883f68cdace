"""ISA lint of the shipped library (no GPU): the 128x128 GEMM family's LDS ring must wait with a COUNTED vmcnt directly in front of a bare s_barrier and nowhere else.

Why a test: this round's first four-stage ring closed its iterations with __syncthreads() — a fence, for which hipcc emits `s_waitcnt vmcnt(0)` before the s_barrier —
and so waited for every panel pair in flight; it measured "no gain" and a wrong conclusion went into the notebook (profiles/NOTEBOOK.md, "a wrong turn").  hipcc can also put
a vmcnt wait in front of an LDS read that follows an LDS-DMA it cannot prove apart (it does in flash_attn_kernel).  Either would silently turn the ring back into a
two-stage loop; the bits would not change, only the time.  So the disassembly of the library the tests load is checked: llvm-objdump --offloading, then -d per kernel."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# (mangled kernel, LDS-DMA instructions per wave and panel pair): bf16 EPI_STORE, geometry WM x WN, four stages
RING_KERNELS = [("_ZN2rz11gemm_kernelIDF16bLi0EDF16bLb0ELi2ELi2ELi4EEEvNS_8GemmArgsE", 8),
                ("_ZN2rz11gemm_kernelIDF16bLi0EDF16bLb0ELi2ELi1ELi4EEEvNS_8GemmArgsE", 12),
                ("_ZN2rz11gemm_kernelIDF16bLi0EDF16bLb0ELi1ELi1ELi4EEEvNS_8GemmArgsE", 16),
                ("_ZN2rz11gemm_kernelIDF16bLi9EDF16bLb0ELi2ELi1ELi4EEEvNS_8GemmArgsE", 12),      # EPI_RESID_SCALE_LN: out-projection / fc2 of one 518^2 image
                ("_ZN2rz11gemm_kernelIDF16_Li4EDF16_Lb1ELi2ELi2ELi4EEEvNS_8GemmArgsE", 8)]      # fp32 mode, MX form (f16 + block-scaled fp8 panels)


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    from radzero_amd import _lib
    lib = _lib.build()
    d = tmp_path_factory.mktemp("isa")
    shutil.copy(lib, d / "lib.so")
    r = subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    objs = sorted(glob.glob(str(d / "lib.so.*gfx950")))
    assert objs, "no gfx950 code object in the library"
    return objs


def disassemble(objs, symbol):
    for o in objs:
        r = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={symbol}", o], capture_output=True, text=True)
        ins = [ln.split("//")[0].strip() for ln in r.stdout.splitlines() if "//" in ln]
        if len(ins) > 50:
            return ins
    return None


@pytest.mark.parametrize("symbol,nper", RING_KERNELS)
def test_ring_waits_are_counted_and_sit_in_front_of_the_barrier(code_objects, symbol, nper):
    ins = disassemble(code_objects, symbol)
    assert ins is not None, f"{symbol} not in the library (instantiation renamed? update RING_KERNELS)"
    mfma = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
    assert sum(x.startswith("global_load_lds_dwordx4") for x in ins) >= 2 * nper, "LDS-DMA staging not found"
    # (1) between the first and the last MFMA (the K loop; the epilogue's own loads and waits come later) a vmcnt wait exists only as the ring's wait: in front of s_barrier
    for i in range(mfma[0], mfma[-1] + 1):
        if re.match(r"s_waitcnt .*vmcnt\(\d+\)", ins[i]):
            assert ins[i + 1].startswith("s_barrier"), f"{symbol}: '{ins[i]}' inside the K loop is not the ring's wait (next: '{ins[i + 1]}'): a compiler-inserted wait would serialise the ring"
    # (2) every barrier of the kernel has the ring's wait in front of it (no fence-style vmcnt(0) + barrier pair added by the compiler), and the waits are the counted ones
    seen = set()
    for i, x in enumerate(ins):
        if x.startswith("s_barrier"):
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)$", ins[i - 1])
            assert m, f"{symbol}: s_barrier at {i} behind '{ins[i - 1]}', not behind the ring's own wait"
            seen.add(int(m.group(1)))
    assert seen == {0, nper, 2 * nper}, f"{symbol}: vmcnt immediates {sorted(seen)}, expected 0 / {nper} / {2 * nper} (three panel pairs in flight)"


@pytest.mark.parametrize("symbol", ["_ZN2rz14gemm_kernel_v8IDF16bLi10ELb0ELb0ELin1EEEvNS_8GemmArgsE",       # bf16 fc1 (EPI_GELU_LN)
                                    "_ZN2rz14gemm_kernel_v8IDF16bLi9ELb0ELb0ELin1EEEvNS_8GemmArgsE",        # bf16 out-projection / fc2 (EPI_RESID_SCALE_LN)
                                    "_ZN2rz14gemm_kernel_v8IDF16bLi1ELb0ELb0ELin1EEEvNS_8GemmArgsE"])       # bf16 fc1, plain GELU epilogue
def test_persistent_gemm_loop_has_only_its_counted_waits(code_objects, symbol):
    """The headline GEMM kernel (gemm8.hip): between its first and last MFMA — the staggered K loop and the epilogue that runs under the next tile's loop — every vector-memory
    wait is one of the counted ones the schedule was built on; a `vmcnt(0)` there (a fence, a compiler-inserted wait in front of an LDS read) would drain the operand ring once per
    phase.  Round 6 found exactly that kind of wait elsewhere twice; this pins the kernel that the headline number rests on."""
    ins = disassemble(code_objects, symbol)
    assert ins is not None, f"{symbol} not in the library (instantiation renamed? update the list)"
    mfma = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
    span = ins[mfma[0]:mfma[-1] + 1]
    waits = [x for x in span if re.match(r"s_waitcnt .*vmcnt\(\d+\)", x)]
    assert len(mfma) >= 128 and sum(x.startswith("s_barrier") for x in span) >= 8 and len(waits) >= 4, "not the K loop this test was written for"
    drained = [x for x in waits if "vmcnt(0)" in x]
    assert not drained, f"{symbol}: {len(drained)} x '{drained[0]}' inside the K loop"
