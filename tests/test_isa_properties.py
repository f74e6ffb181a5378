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


def _frame_loop(lines, marker="v_dot2_u32_u16"):
    """The smallest loop that holds every `marker` instruction: the per-frame step of a time-walking kernel."""
    at = [i for i, l in enumerate(lines) if marker in l]
    assert at, "no " + marker
    spans = [s for s in _loops(lines) if s[0] <= at[0] and s[1] >= at[-1]]
    assert spans
    a, b = min(spans, key=lambda s: s[1] - s[0])
    return lines[a:b + 1]


def test_fused_downscale_kernel_reads_its_taps_as_dwords(tmp_path):
    """Round 6: the pixel path of resize_walk_kernel.  Per destination pixel and frame: two `ds_read2_b32` + two
    `ds_read_b32` for the four taps (no byte-wide LDS reads: the kernel it replaces issued sixteen, and an unaligned `ds_read_b64`
    per row -- what hipcc makes of the obvious source -- runs on a slow path of the LDS), six `v_dot2_u32_u16` and six
    `v_mul_hi_u32_u24` for the interpolation, no exec-mask branch per pixel slot."""
    bodies = kernel_bodies(device_asm("psd_resize_kernels", tmp_path))
    checked = 0
    for name, lines in bodies.items():
        m = re.search(r"resize_walk_kernelILb([01])ELb([01])ELi(\d)ELb([01])ELb([01])ELb([01])E", name)
        if not m:
            continue
        store, hsv, g, luma, seg = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
        luma |= int(m.group(6))        # (the V-plane instances count V with the luma instances' histogram machinery)
        # (hipcc may unswitch the frame loop on the uniform "exact 2 x 2 decimation" flag: the general copy is the one with the
        #  vertical products)
        body = [l.strip() for l in _frame_loop(lines, "v_mul_hi_u32_u24") if l.startswith("\t") and l.strip()]
        whole = [l.strip() for l in lines if l.startswith("\t") and l.strip()]
        count = lambda op, where=body: sum(1 for l in where if l.split()[0].startswith(op))      # noqa: E731
        # no byte-wide or unaligned-wide LDS reads anywhere in the kernel (the only 8-byte LDS read left is the luma instances'
        # aligned read of two histogram bins in the per-frame flush)
        assert count("ds_read_u8", whole) == 0 and count("ds_read_u16", whole) == 0 and count("ds_read_b96", whole) == 0, name
        assert count("ds_read_b64", whole) <= (2 if luma else 0), (name, count("ds_read_b64", whole))
        # (per copy of the step -- one, or two where the loop was unswitched: 2 G paired dword reads, 6 G dot products, 4 G realignments;
        #  the vertical products only exist in the general copy)
        copies = count("v_dot2_u32_u16", whole) // (6 * g)
        assert copies in (1, 2) and count("v_dot2_u32_u16", whole) == 6 * g * copies, (name, count("v_dot2_u32_u16", whole))
        assert count("ds_read2_b32", whole) == 2 * g * copies and count("v_alignbyte_b32", whole) == 4 * g * copies, name
        assert count("v_mul_hi_u32_u24", whole) == 6 * g, name
        valu = sum(1 for l in body if l.startswith("v_"))
        # (set-up and flushes are amortised over the slots: the narrow instances carry them on few pixels)
        limit = {1: 150, 2: 120, 4: 100, 8: 95}[g] + (15 if luma else 0) + (10 if store else 0)
        assert valu / g <= limit, (name, valu / g)
        checked += 1
    assert checked == 40


def test_fused_downscale_seg_instance_reads_the_flag_behind_the_dma_issue(tmp_path):
    """The clip-start flag of the next frame is requested BEHIND that frame's LDS-DMA issue and consumed at the top of its own step:
    between the issue and the end of the step the compiler must not wait for global memory (an `s_waitcnt vmcnt(0)` there drains
    the prefetch the wave has just started: psd_score_kernels.hip found that out in round 5)."""
    bodies = kernel_bodies(device_asm("psd_resize_kernels", tmp_path))
    checked = 0
    for name, lines in bodies.items():
        if not re.search(r"resize_walk_kernelILb[01]ELb1ELi\dELb[01]ELb1E", name):
            continue
        loops = _loops(lines)
        in_loop = 0
        for i, l in enumerate(lines):
            if "global_load_ubyte" not in l:
                continue
            spans = [sp for sp in loops if sp[0] <= i <= sp[1]]
            if not spans:
                continue                      # the first frame's flag, requested in front of the walk
            in_loop += 1
            a, b = max(spans, key=lambda sp: sp[1] - sp[0])      # the frame loop (the kernel's only outer loop): one step of the walk
            step = [x.strip() for x in lines[a:b + 1] if x.startswith("\t") and x.strip()]
            at = next(k for k, x in enumerate(step) if x.startswith("global_load_ubyte"))
            # DMA issues in front of it in the step -- in the text, or in blocks hipcc placed out of line (the instance that also stores
            # its pixels: the issue sits behind the loop body and is entered by branches from in front of the flag's load) --, no wait
            # for global memory behind it
            dma_blocks, label = set(), None
            for x in lines:               # (the whole kernel: such blocks may sit behind the loop's back edge)
                m = re.match(r"^(\.LBB\d+_\d+):", x)
                if m:
                    label = m.group(1)
                elif x.strip().startswith("global_load_lds_dwordx4") and label:
                    dma_blocks.add(label)
            enters = any(x.startswith("s_cbranch") and x.split()[-1] in dma_blocks for x in step[:at])
            assert any(x.startswith("global_load_lds_dwordx4") for x in step[:at]) or enters, name
            # (a wait in a block hipcc placed behind the loop body that branches back IN FRONT of the step's barrier belongs to the
            #  step's own wait at its top -- the ladder of s_waitcnt vmcnt(n) alternatives, fifth session of round 6 -- not to the
            #  stretch behind the flag's load)
            bar0 = next(k for k, x in enumerate(step) if x.startswith("s_barrier"))
            label_pos, n_ins = {}, 0
            for x in lines[a:b + 1]:
                lm = re.match(r"^(\.LBB\d+_\d+):", x)
                if lm:
                    label_pos[lm.group(1)] = n_ins
                elif x.startswith("\t") and x.strip():
                    n_ins += 1

            def top_of_step_wait(k):
                for y in step[k + 1:]:
                    if y.startswith(("s_branch", "s_cbranch")):
                        return label_pos.get(y.split()[-1], 1 << 30) < bar0
                return False
            assert not any(x.startswith("s_waitcnt") and "vmcnt(0)" in x and not top_of_step_wait(at + k) for k, x in enumerate(step[at:])), name
            # ... and the flag becomes an SGPR in front of the step's barrier (right behind the step's own wait)
            bar = next(k for k, x in enumerate(step) if x.startswith("s_barrier"))
            assert any(x.startswith("v_readfirstlane_b32") for x in step[:bar]), name
        assert in_loop in (1, 2), (name, in_loop)     # (two where the frame loop was unswitched)
        checked += 1
    assert checked == 16      # (HSV | HSV + store | HSV + luma | HSV + V plane) x four pixel-slot counts


def test_fused_downscale_kernel_issues_its_stores_behind_the_dma_of_the_step(tmp_path):
    """Round 6, fifth session: a step of the walk waits with `s_waitcnt vmcnt(n)`, n = the store instructions the wave
    issued in the step before, for the LDS-DMA of its frame -- correct only because those stores (the pixels / V plane of the frame
    before last, a tile's partial histogram) are YOUNGER than the DMA: in every instance that stores inside the frame loop the step's
    DMA issue -- in the text or in blocks hipcc placed out of line and entered by a branch -- lies between the step's barrier and its
    first 16-byte / dword store."""
    bodies = kernel_bodies(device_asm("psd_resize_kernels", tmp_path))
    checked = 0
    for name, lines in bodies.items():
        m = re.search(r"resize_walk_kernelILb([01])ELb([01])ELi(\d)ELb([01])ELb([01])ELb([01])E", name)
        if not m:
            continue
        store, hsv, g, luma, seg, vout = (int(x) for x in m.groups())
        if not (store or vout or luma):
            continue
        dma_blocks, label = set(), None
        for x in lines:
            lm = re.match(r"^(\.LBB\d+_\d+):", x)
            if lm:
                label = lm.group(1)
            elif x.strip().startswith("global_load_lds_dwordx4") and label:
                dma_blocks.add(label)
        # the frame loop: the depth-1 loop header whose blocks (its own and those marked "in Loop: Header=...", wherever hipcc
        # placed them) hold a barrier; its instructions in layout order from the header on
        step = None
        for h, x in enumerate(lines):
            hm = re.match(r"^\.L(BB\d+_\d+):.*Loop Header: Depth=1", x)
            if not hm:
                continue
            inside, body = True, []
            mine = {hm.group(1)} | {cm.group(1) for y in lines for cm in [re.match(r"^\.L(BB\d+_\d+):.*Parent Loop " + hm.group(1) + r"\b", y)] if cm}
            for y in lines[h + 1:]:
                if re.match(r"^\.LBB\d+_\d+:", y):
                    inside = any(("Header=" + k + " ") in y + " " or ("Parent Loop " + k + " ") in y + " " for k in mine)
                elif inside and y.startswith("\t") and y.strip():
                    body.append(y.strip())
            if any(y.startswith("s_barrier") for y in body):
                step = body
                break
        assert step, name
        bar = next(k for k, x in enumerate(step) if x.startswith("s_barrier"))
        stores = [k for k, x in enumerate(step) if x.startswith(("global_store_dwordx4", "global_store_dword "))]
        assert stores and min(stores) > bar, name
        head = step[bar:min(stores)]
        assert any(x.startswith("global_load_lds_dwordx4") for x in head) or \
            any(x.startswith("s_cbranch") and x.split()[-1] in dma_blocks for x in head), name
        checked += 1
    assert checked >= 24, checked
