"""Static checks of the generated gfx950 code of the two hot convolution kernels (no GPU needed: hipcc cross-compiles).

Why they exist: in round 3 a prologue change made the compiler treat a buffer resource descriptor as divergent; every LDS-DMA
instruction of the K loop was then wrapped in a waterfall loop (v_readfirstlane x4 + compare + s_and_saveexec), +25 % VALU per K step
and -6 % on the whole plain-layer family, with every parity test still green.  Nothing but the ISA shows that."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ("modconv_w2d", "modconv_up2d")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa")

    def build(name):
        dst = str(out / f"{name}.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f"-I{REPO}/include",
                        f"{REPO}/maua_stylegan2_amd/csrc/{name}.hip", "-o", dst], check=True, capture_output=True)
        return name, open(dst).read()

    with ThreadPoolExecutor(len(SOURCES)) as pool:
        return dict(pool.map(build, SOURCES))


@pytest.mark.parametrize("name", SOURCES)
def test_lds_dma_loads_are_not_waterfalled(device_asm, name):
    lines = device_asm[name].split("\n")
    dma = [i for i, line in enumerate(lines) if "buffer_load" in line and " lds" in line]
    assert dma, "the kernels stage their operands with buffer_load ... lds"
    for i in dma:
        before = " ".join(lines[max(0, i - 10):i])
        assert not ("v_readfirstlane" in before and "s_and_saveexec" in before), (
            f"{name}.hip: the LDS-DMA load at asm line {i} sits in a waterfall loop — its buffer descriptor is no longer uniform "
            "(keep __builtin_amdgcn_make_buffer_rsrc ahead of thread-dependent prologue loops)")


@pytest.mark.parametrize("name", SOURCES)
def test_hot_kernels_do_not_spill(device_asm, name):
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", device_asm[name])]
    sspills = [int(v) for v in re.findall(r"\.sgpr_spill_count:\s+(\d+)", device_asm[name])]
    assert spills and max(spills) == 0 and max(sspills) == 0, (spills, sspills)


def _main_loop(asm, mangled_fragment):
    """Instruction lines of the loop with the most MFMAs of the kernel whose mangled name contains ``mangled_fragment``."""
    lines = asm.split("\n")
    start = next(i for i, line in enumerate(lines) if line.startswith("_Z") and mangled_fragment in line.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    labels = {m.group(1): i for i, line in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", line))}
    loops = []
    for i, line in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))
    lo, hi = max(loops, key=lambda ab: sum("v_mfma" in line for line in body[ab[0]:ab[1]]))
    return [line.strip() for line in body[lo:hi + 1] if line.strip() and line.strip()[0] not in ".;"]


def test_phase_locked_k_step_has_a_load_segment_and_a_matrix_segment(device_asm):
    """modconv_w2d_kernel<4, 2, 1, true> (two four-wave groups per workgroup, locked in opposite phases): the K step must be
    [8 LDS-DMA pieces, window reads, input transforms] s_barrier [weight-row reads, 48 MFMAs] s_waitcnt vmcnt(0) s_barrier — no
    matrix instruction and no VMEM wait in the load segment (a `__syncthreads()` there drains vmcnt and exposes the latency of the DMA
    the segment has just issued; transforms that slide behind the barrier steal issue slots from the partner's matrix segment)."""
    loop = _main_loop(device_asm["modconv_w2d"], "modconv_w2d_kernelILi4ELi2ELi1ELb1")
    barriers = [i for i, op in enumerate(loop) if op.startswith("s_barrier")]
    assert len(barriers) == 2, barriers
    load, matrix = loop[:barriers[0]], loop[barriers[0] + 1:barriers[1]]
    assert sum("buffer_load" in op and " lds" in op for op in load) == 8
    assert not any(op.startswith("v_mfma") for op in load) and not any("vmcnt" in op for op in load)
    assert sum(op.startswith("v_mfma") for op in matrix) == 48
    valu_in_matrix = [op for op in matrix if op.startswith("v_") and not op.startswith("v_mfma")]
    assert len(valu_in_matrix) <= 8, valu_in_matrix  # (address bumps only: the transforms live in the load segment)
    assert any("vmcnt(0)" in op for op in matrix[-8:])
