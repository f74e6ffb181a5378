"""Compile-only checks (hipcc cross-compiles gfx950 without a GPU) of two properties of the generated ISA that cost
double-digit percentages when they were lost:

* the staged scoring kernels must not wait for the NEXT frame's LDS-DMA inside a step: hipcc puts `s_waitcnt vmcnt(0)` in
  front of LDS reads whose address it cannot tell apart from the staging slots, and in front of every LDS store / atomic;
* the Sobel / NMS bit-plane kernel must issue its six tile loads before the first LDS write: a load behind a branch is waited for
  before the next one goes out.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def device_asm(name, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (name + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "pyscenedetect_amd", "csrc"), os.path.join(ROOT, "pyscenedetect_amd", "csrc", name + ".hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def kernel_bodies(asm):
    return {m.group(1): m.group(2).split("\n")
            for m in re.finditer(r"^(_ZN3psd\w+):\s*;.*?\n(.*?)\.amdhsa_kernel \1", asm, re.S | re.M)}


def _instructions(lines):
    """instruction mnemonics + operands in layout order; hand-written asm blocks are tagged"""
    out, in_asm = [], False
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t and not t.startswith((";", ".")) and not t.endswith(":"):
            out.append(("asm " if in_asm else "") + t)
    return out


def staged_kernel_wait_signature(lines):
    """(compiler-inserted vmcnt waits directly in front of an LDS table read, ... in front of a compiler-visible LDS add)"""
    ins = _instructions(lines)
    before_read = before_add = 0
    for i, (a, b) in enumerate(zip(ins, ins[1:])):
        if a.startswith("s_waitcnt") and "vmcnt" in a:
            # (a read right in front of the step's barrier -- the accumulator the sums were just added to, when a step without
            #  a predecessor skips the additions -- is the end of the step like the additions themselves: harmless)
            at_step_end = any(x.startswith("s_barrier") for x in ins[i + 2:i + 6])
            before_read += b.startswith("ds_read_b32") and not at_step_end
            before_add += b.startswith("ds_add_u32") or (b.startswith("ds_read_b32") and at_step_end)
    return before_read, before_add


def test_staged_kernels_do_not_wait_for_the_prefetch_inside_a_step(tmp_path):
    bodies = kernel_bodies(device_asm("psd_score_kernels", tmp_path))
    checked = 0
    for name, lines in bodies.items():
        if "score_frames_dma_kernel" not in name and "luma_hist_kernel" not in name:
            continue
        before_read, before_add = staged_kernel_wait_signature(lines)
        # table reads never wait for global memory; the only LDS adds the compiler sees are the per-step (or final) sums,
        # one group per copy of the step (the time walk runs two steps per iteration with the register sets of the previous
        # and the current frame swapped), behind which the wait is harmless (the step is over)
        # (per copy of the step: the additions, and the read behind them that a step without a predecessor jumps to)
        assert before_read == 0 and before_add <= 4, (name, before_read, before_add)
        checked += 1
    assert checked >= 6


def test_sobel_tile_kernel_issues_its_loads_together(tmp_path):
    bodies = kernel_bodies(device_asm("psd_edge_kernels", tmp_path))
    # (the instance for dword-aligned rows; the RAGGED one assembles its tile from byte loads)
    lines = next(v for k, v in bodies.items() if "sobel_nms_bits_kernelILb0E" in k)
    # the two dword-load forms of the tile loader (interior tiles / clamped border tiles; frames whose width is not a multiple
    # of 4 assemble their dwords from byte loads elsewhere): in each, all six loads are issued before the first LDS write
    events = [("w" if "ds_write_b32" in l else "l", i) for i, l in enumerate(lines)
              if "ds_write_b32" in l or "global_load_dword " in l or "global_load_dword\t" in l]
    clusters, run = [], 0
    for kind, _ in events:
        if kind == "l":
            run += 1
        elif run:
            clusters.append(run)
            run = 0
    if run:
        clusters.append(run)
    assert len(clusters) == 2 and min(clusters) >= 6, clusters
