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


def _loops(lines):
    """(first, last) line index of every backward branch's span"""
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    out = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def test_hysteresis_step_shifts_rows_with_dpp_not_through_the_lds(tmp_path):
    """Round 4: the fix-point step of a hysteresis tile takes the rows above / below from the neighbouring lanes with whole-wave
    DPP shifts; as ds_bpermute_b32 (what __shfl_up / __shfl_down compile to) they were eight LDS round trips per step."""
    bodies = kernel_bodies(device_asm("psd_edge_kernels", tmp_path))
    checked = 0
    for name, lines in bodies.items():
        if "hysteresis_frame_kernel" not in name:
            continue
        inner = [span for span in _loops(lines) if any("wave_shr:1" in l for l in lines[span[0]:span[1] + 1])]
        assert inner, name
        a, b = min(inner, key=lambda s: s[1] - s[0])          # the fix-point loop itself
        body = lines[a:b + 1]
        assert sum("wave_shr:1" in l for l in body) == 2 and sum("wave_shl:1" in l for l in body) == 2, name
        assert not any("ds_bpermute" in l or "ds_read" in l for l in body), name
        assert not any("ds_bpermute" in l for l in lines), name   # ... and nowhere else in the kernel
        checked += 1
    assert checked == 3


def test_busy_sobel_tiles_select_their_neighbours_without_branching(tmp_path):
    """Round 4: on a busy tile every pixel is a candidate; its two neighbours along the gradient come out of registers by selects.
    Written with && / nested ?: hipcc made eight divergent branches per pixel of it (69 in the loop over a thread's eight pixels)."""
    bodies = kernel_bodies(device_asm("psd_edge_kernels", tmp_path))
    lines = next(v for k, v in bodies.items() if "sobel_nms_bits_kernelILb0E" in k)
    a, b = max(_loops(lines), key=lambda s: s[1] - s[0])       # the longest loop: the eight pixels of a thread, unrolled
    body = [l.strip() for l in lines[a:b + 1] if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
    branches = sum(1 for l in body if l.startswith(("s_cbranch", "s_branch")))
    selects = sum(1 for l in body if l.startswith("v_cndmask"))
    assert len(body) < 650 and branches <= 16 and selects >= 40, (len(body), branches, selects)
