#!/usr/bin/env python3
"""Dynamic instruction mix of the trace kernel per source section, from the basic-block profile (development aid).

    python3 tools/bbprof_report.py <bbmap.json> <bbprof_<workload>.json> [--pmc key=value ...] [--lines N] > profiles/r06/isa_mix_by_section_<workload>.md

bbmap.json: tools/bbprof_rewrite.py (every instruction of rayhs_trace_exact with its straight-line segment and the source lines of its
.loc chain); bbprof_<workload>.json: tools/bbprof_run.py (executions of every segment, summed over the waves of one steady-state frame).
An instruction executes as often as its segment; it is charged to the section of trace_kernel.hip it was WRITTEN in (the innermost frame
of its inline chain that lies in begin() / traverse() / visit() / meshes_reference(), else the one in trace_body(): helpers like dot() or
ws_store() are charged to their caller).  --pmc SQ_INSTS_VALU=... SQ_INSTS_SALU=... : hardware counters of the un-instrumented kernel on
the same frame (tools/inst_counts.sh), printed beside the totals as the reconciliation."""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rayhs_amd", "csrc", "trace_kernel.hip")
if "--src" in sys.argv:   # the source the profiled library was built from, when that is not the working tree's (its line numbers are what the map holds)
    SRC = sys.argv[sys.argv.index("--src") + 1]
src = open(SRC).read().split("\n")


def find(pat, start=0):
    for i in range(start, len(src)):
        if pat in src[i]:
            return i + 1
    raise KeyError(pat)


# (name, first line) in source order; a section ends where the next one begins
MARKS = [("(helpers above the tracer)", 1),
         ("begin(): preamble (masks, scene head, K)", find("__device__ void begin(")),
         ("begin(): plane loops", find("auto plane_range = ")),
         ("begin(): spheres", find("const auto spheres = (cptr<DSphere>)")),
         ("begin(): hand-off to the reference tree / arming", find("bool occluded = __builtin_amdgcn_inverse_ballot_w64(m_occ);")),
         ("meshes_reference() (reference tree walk)", find("__device__ __forceinline__ bool meshes_reference(")),
         ("visit(): per-lane node visit", find("template <class ND> __device__ __forceinline__ void visit(")),
         ("traverse(): per-mesh set-up, root cull, packet decision", find("__device__ void traverse(bool shadow")),
         ("packet walk: node fetch, two box tests, child order", find("const stack_ptr wstack = stack")),
         ("packet walk: leaf triangles (fp64 test, masks, tie rule)", find("auto leaf = [&](const int c, const unsigned long long m_side) {")),
         ("packet walk: inner children (selects), push / pop, loop foot", find("if ((unsigned)ka > (unsigned)FNODE_NONE) leaf(ka, ma);")),
         ("per-lane walk: start, node loop control", find("if (enter) w.t_cur = me.froot;")),
         ("per-lane walk: leaf triangles", find("// phase 2: the postponed leaf")),
         ("per-lane walk: pop, loop end", find("if (w.t_pa == FNODE_NONE && w.t_cur == FNODE_NONE && w.t_sp > 0) {")),
         ("(workspace / context helpers)", find("// ---- global per-lane workspace")),
         ("trace_body(): kernel set-up", find("static __device__ __forceinline__ void trace_body()")),
         ("main loop top + chunk refill + primary rays", find("while (true) {", find("static __device__ __forceinline__ void trace_body()"))),
         ("advance: dispatch (state machine entry)", find("// ---- advance: consume the last query's answer")),
         ("hit section", find("if (state == ST_HIT) {")),
         ("shadow-return section", find("} else if (state == ST_SHADOW) {")),
         ("exit section (refraction)", find("} else if (state == ST_EXIT) {")),
         ("light section", find("if (step == A_LIGHT_NEXT) {")),
         ("specular section", find("if (step == A_SPEC) {")),
         ("return section", find("if (step == A_RETURN) {")),
         ("query site, epilogue", find("// ---- the single scene-query site")),
         ("(end)", len(src) + 1)]
TRACER = (find("__device__ void begin("), find("// ---- global per-lane workspace"))          # begin() .. traverse()
BODY = (find("static __device__ __forceinline__ void trace_body()"), find("}  // namespace rayhs", find("static __device__ __forceinline__ void trace_body()")))


def load_map(path):
    import gzip
    return json.load(gzip.open(path, "rt") if path.endswith(".gz") else open(path))


def section_of(locs):
    line = -1
    for lo, hi in (TRACER, BODY):
        hit = next((l for l in locs if lo <= l < hi), None)
        if hit is not None:
            line = hit
            break
    if line < 0:
        line = locs[0] if locs else -1
    for k in range(len(MARKS) - 1):
        if MARKS[k][1] <= line < MARKS[k + 1][1]:
            return k, line
    return 0, line


def kind(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "BRANCH"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_atomic", "s_dcache", "s_memtime", "s_memrealtime")):
        return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_sleep", "s_setprio", "s_barrier")):
        return "SOPP"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "VMEM"
    return "other"


def vclass(op):
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op): return "trans64"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_cvt"): return "cvt"
    if re.search(r"_f64", op): return "f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", op): return "trans32"
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32", op) or re.match(r"v_pk_(add|mul|fma)_f32", op): return "f32"
    if re.search(r"_f32", op): return "minmax32"       # v_min / v_max / v_min3 / v_max3 / v_med3 _f32: no class counter covers them
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")): return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    return "int"


VCOST = {"f64": 4, "trans64": 16, "f32": 2, "minmax32": 2, "trans32": 8, "mov": 2, "cmp": 4, "cndmask": 4, "cvt": 4, "lane": 4, "int": 4}
VNAMES = ("f64", "trans64", "f32", "minmax32", "int", "cvt", "cmp", "cndmask", "mov", "lane", "trans32")


def main():
    bbmap = load_map(sys.argv[1])
    prof = json.load(open(sys.argv[2]))
    pmc = dict(a.split("=") for a in sys.argv[3:] if re.match(r"^SQ_\w+=", a))
    top_lines = int(sys.argv[sys.argv.index("--lines") + 1]) if "--lines" in sys.argv else 40
    counts = prof["segment_counts"]
    segs = bbmap["segments"]
    assert len(counts) >= len(segs)
    tot = collections.Counter()
    sec = collections.defaultdict(collections.Counter)
    secv = collections.defaultdict(collections.Counter)
    by_line = collections.defaultdict(collections.Counter)
    static = collections.defaultdict(collections.Counter)
    for seg, n in zip(segs, counts):
        for op, locs, *_ in seg:
            k = kind(op)
            s, line = section_of(locs)
            tot[k] += n
            sec[s][k] += n
            static[s][k] += 1
            by_line[line][k] += n
            if k == "VALU":
                c = vclass(op)
                secv[s][c] += n
                tot["v:" + c] += n
    allv = sum(tot[k] for k in ("VALU", "SALU", "SMEM", "BRANCH", "SOPP", "LDS", "VMEM", "other"))
    scalar_unit = tot["SALU"] + tot["SMEM"] + tot["BRANCH"]
    vcycles = sum(VCOST[c] * tot["v:" + c] for c in VNAMES)
    M = lambda v: f"{v / 1e6:.2f}"
    print(f"# Dynamic instruction mix of `rayhs_trace_exact` per source section -- {prof['workload']} ({prof['scene']}, {prof['spp']} spp), one steady-state frame\n")
    print(f"Made by `tools/bbprof_report.py` from the basic-block profile (`tools/bbprof_rewrite.py`: one `s_atomic_add` per straight-line segment of the production kernel's own assembly, "
          f"{len(segs)} segments, {sum(len(s) for s in segs)} instructions; `tools/bbprof_run.py`: {prof['waves']} waves, frame == oracle digest: {prof['frame_sha256_equals_oracle_digest']}, "
          f"every counter an exact multiple of its wave's key: {prof['every_counter_a_multiple_of_its_wave_high_word']}).  Counts in millions of wave-level instructions per frame.\n")
    print("## Totals, and the hardware counters of the un-instrumented kernel on the same frame\n")
    print("| class | basic-block profile | rocprofv3 counter | ratio |\n|---|---|---|---|")
    rows = [("VALU", tot["VALU"], "SQ_INSTS_VALU"), ("SALU (without s_waitcnt / s_nop / s_setprio)", tot["SALU"], "SQ_INSTS_SALU"), ("scalar memory", tot["SMEM"], "SQ_INSTS_SMEM"),
            ("branches", tot["BRANCH"], "SQ_INSTS_BRANCH"), ("LDS", tot["LDS"], "SQ_INSTS_LDS"), ("vector memory", tot["VMEM"], "SQ_INSTS_VMEM"),
            ("s_waitcnt, s_nop, s_setprio, s_endpgm", tot["SOPP"], None)]
    for name, v, key in rows:
        hw = float(pmc[key]) if key and key in pmc else None
        print(f"| {name} | {M(v)} | {M(hw) if hw else '-'} | {v / hw:.3f} |" if hw else f"| {name} | {M(v)} | - | - |")
    print(f"\n(The profile's own `s_atomic_add`s -- one per segment execution, {M(sum(counts[:len(segs)]))} M -- are not among the instructions counted.)  "
          f"Scalar-unit instructions (SALU + scalar memory + branches): **{M(scalar_unit)} M**; vector-pipe cycles at the unit costs of "
          f"`profiles/r05/micro/issue_rates.txt` ({', '.join(f'{c} {VCOST[c]}' for c in VNAMES)}): **{vcycles / 1e9:.3f} G**.\n")
    print("VALU by class: " + ", ".join(f"{c} {M(tot['v:' + c])}" for c in VNAMES) + "\n")
    print("## Per section\n")
    print("| section | static V / S | VALU | SALU | SMEM | branch | LDS | VMEM | share of all instructions | VALU pipe cycles (G) | share of VALU cycles | share of scalar-unit instructions |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k in range(len(MARKS) - 1):
        c = sec[k]
        if not sum(c.values()):
            continue
        cyc = sum(VCOST[v] * secv[k][v] for v in VNAMES)
        su = c["SALU"] + c["SMEM"] + c["BRANCH"]
        everything = sum(c.values())
        print(f"| {MARKS[k][0]} | {static[k]['VALU']} / {static[k]['SALU'] + static[k]['SMEM'] + static[k]['BRANCH']} | {M(c['VALU'])} | {M(c['SALU'])} | {M(c['SMEM'])} | {M(c['BRANCH'])} | {M(c['LDS'])} | {M(c['VMEM'])} | "
              f"{everything / allv:.3f} | {cyc / 1e9:.3f} | {cyc / vcycles:.3f} | {su / scalar_unit:.3f} |")
    print("\n## VALU classes per section (millions)\n")
    print("| section | " + " | ".join(VNAMES) + " |\n|---|" + "---|" * len(VNAMES))
    for k in range(len(MARKS) - 1):
        if sum(secv[k].values()):
            print(f"| {MARKS[k][0]} | " + " | ".join(M(secv[k][v]) for v in VNAMES) + " |")
    print(f"\n## The {top_lines} source lines that execute most instructions (line of trace_kernel.hip the instruction was written on; helpers charged to their caller)\n")
    print("| line | instructions (M) | VALU | scalar unit | source |\n|---|---|---|---|---|")
    ranked = sorted(by_line.items(), key=lambda kv: -sum(kv[1].values()))[:top_lines]
    for line, c in ranked:
        text = src[line - 1].strip()[:110].replace("|", "\\|") if 0 < line <= len(src) else "(other files)"
        print(f"| {line} | {M(sum(c.values()))} | {M(c['VALU'])} | {M(c['SALU'] + c['SMEM'] + c['BRANCH'])} | `{text}` |")


def hot_segments():
    """--segments N: the N straight-line segments that execute most instructions, instruction by instruction (for reading, not for the table)"""
    bbmap = load_map(sys.argv[1])
    prof = json.load(open(sys.argv[2]))
    n_top = int(sys.argv[sys.argv.index("--segments") + 1])
    segs, counts = bbmap["segments"], prof["segment_counts"]
    order = sorted(range(len(segs)), key=lambda i: -counts[i] * len(segs[i]))[:n_top]
    for i in order:
        print(f"\n== segment {i}: executed {counts[i] / 1e6:.2f} M times x {len(segs[i])} instructions = {counts[i] * len(segs[i]) / 1e6:.1f} M")
        for ins in segs[i]:
            op, locs = ins[0], ins[1]
            text = ins[2] if len(ins) > 2 else op
            print(f"   L{'<'.join(str(l) for l in locs[:3]):16s} {text}")


if __name__ == "__main__":
    if "--segments" in sys.argv:
        hot_segments()
    else:
        main()
