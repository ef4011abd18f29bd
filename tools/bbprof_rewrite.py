#!/usr/bin/env python3
"""Basic-block execution counts of the PRODUCTION trace kernel (development aid; profiles/r06/isa_mix_by_section.md is made from it).

    tools/bbprof_rewrite.py <dev_exact.s> <dev_exact_mod.s> [<map.json>]        (run by tools/build_asm_rewrite.sh bbprof tools/bbprof_rewrite.py -gline-tables-only)

Rewrites the DEVICE assembly of rayhs_trace_exact between compiler and assembler: at the head of every straight-line segment (function
entry, every branch-target label, the fall-through behind every conditional branch) ONE scalar instruction is inserted,

    s_atomic_add s101, s[100:101], <segment * 4>

which adds the high word of the wave's own counter-area address (a constant of the run, never zero) to that segment's counter.  The
production kernel leaves s100, s101 and m0 unused (.amdhsa_next_free_sgpr 100), so nothing else changes: same vector and scalar
registers, same occupancy, same instruction stream -- plus one scalar memory instruction per segment executed.  Every wave counts into
an area of its own (s[100:101] = KArgs.counters + wave_slot * stride, set up by eight instructions in front of the kernel's first), so
the atomics never meet; the host divides each wave's counters by that wave's high word and adds the waves up (tools/bbprof_run.py).
Per instruction the dynamic count is its segment's count; tools/bbprof_report.py attributes them to source sections through the .loc
directives (-gline-tables-only) and checks the totals against the SQ_INSTS_* counters of a rocprofv3 pass of the un-instrumented kernel.
"""
import json
import re
import sys

KERNEL = "rayhs_trace_exact"
STRIDE_SEGMENTS = 2048           # counters per wave (uint32); the host allocates waves * STRIDE_SEGMENTS * 4 bytes
BRANCH = re.compile(r"^\s+(s_cbranch_\w+)\b")
INSTR = re.compile(r"^\s+([a-z][a-z0-9_]+)\b")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
LOC = re.compile(r"^\s+\.loc\s+\d+\s+(\d+)\s+\d+.*?;\s*(\S+?):(\d+):\d+(.*)$")
CHAIN = re.compile(r"@\[\s*(\S+?):(\d+):\d+")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    mapping = sys.argv[3] if len(sys.argv) > 3 else dst + ".bbmap.json"
    lines = open(src).read().split("\n")
    begin = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(begin, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[i]))
    # KArgs.counters: 16 bytes in front of KArgs.frames, whose offset the kernel's own first load of it shows (trace_kernel.hip: `const Workspace ws = {kargs()->frames, ...`)
    frames_off = None
    for i in range(begin, end):
        m = re.match(r"^\s+s_load_dwordx2 s\[\d+:\d+\], s\[0:1\], (0x[0-9a-f]+)", lines[i])
        if m:
            frames_off = int(m.group(1), 16)
            break
    assert frames_off is not None, "no s_load_dwordx2 from the kernarg segment found (KArgs.frames)"
    counters_off = frames_off - 16
    out = lines[:begin + 1]
    segments = []          # per segment: list of [opcode, [source lines of the .loc in force, innermost first], is_asm_block]
    cur_loc = []

    def open_segment():
        seg = len(segments)
        assert seg < STRIDE_SEGMENTS
        segments.append([])
        out.append(f"\ts_atomic_add s101, s[100:101], {seg * 4:#x}")
        return seg

    prologue_done = False
    pending_new = False     # the next instruction starts a new segment (fall-through behind a conditional branch)
    for i in range(begin + 1, end):
        l = lines[i]
        m = LOC.match(l)
        if m:
            chain = [(m.group(2), int(m.group(3)))] + [(f, int(n)) for f, n in CHAIN.findall(m.group(4))]
            here = [n for f, n in chain if f.endswith("trace_kernel.hip") and n > 0]
            if here:              # (line 0 = "no source position": such instructions stay with the position in force before them)
                cur_loc = here
            out.append(l)
            continue
        if LABEL.match(l):
            out.append(l)
            open_segment()
            pending_new = False
            continue
        mi = INSTR.match(l)
        if not mi or l.lstrip().startswith((".", ";")):
            out.append(l)
            continue
        op = mi.group(1)
        if not prologue_done:
            # s[100:101] = KArgs.counters + (workgroup * 4 + wave of the workgroup) * stride; only VCC and SCC (dead at entry) are touched
            out += [f"\ts_load_dwordx2 s[100:101], s[0:1], {counters_off:#x}",
                    "\tv_readfirstlane_b32 vcc_lo, v0",
                    "\ts_lshr_b32 vcc_lo, vcc_lo, 6",
                    "\ts_lshl_b32 vcc_hi, s2, 2",
                    "\ts_add_u32 vcc_lo, vcc_lo, vcc_hi",
                    f"\ts_mul_i32 vcc_lo, vcc_lo, {STRIDE_SEGMENTS * 4:#x}",
                    "\ts_waitcnt lgkmcnt(0)",
                    "\ts_add_u32 s100, s100, vcc_lo",
                    "\ts_addc_u32 s101, s101, 0"]
            open_segment()
            prologue_done = True
        elif pending_new:
            open_segment()
        pending_new = False
        if op == "s_endpgm":
            out += ["\ts_waitcnt lgkmcnt(0)", "\ts_dcache_wb"]
        segments[-1].append([op, cur_loc, l.split(";")[0].strip()])
        out.append(l)
        if BRANCH.match(l):
            pending_new = True
    out += lines[end:]
    text = "\n".join(out)
    # the two scalar registers now in use: kernel descriptor and the symbol the metadata is computed from
    k0 = text.index(f".amdhsa_kernel {KERNEL}\n")
    k1 = text.index(".end_amdhsa_kernel", k0)
    desc = text[k0:k1]
    assert ".amdhsa_next_free_sgpr 100" in desc, "the production kernel is expected to leave s100 and s101 free"
    text = text[:k0] + desc.replace(".amdhsa_next_free_sgpr 100", ".amdhsa_next_free_sgpr 102") + text[k1:]
    text = text.replace(f".set {KERNEL}.numbered_sgpr, max(100,", f".set {KERNEL}.numbered_sgpr, max(102,")
    open(dst, "w").write(text)
    json.dump({"kernel": KERNEL, "stride_segments": STRIDE_SEGMENTS, "counters_offset": counters_off, "segments": segments}, open(mapping, "w"))
    n_inst = sum(len(s) for s in segments)
    print(f"bbprof: {len(segments)} segments, {n_inst} instructions of {KERNEL} instrumented -> {dst}, map {mapping}", file=sys.stderr)


if __name__ == "__main__":
    main()
