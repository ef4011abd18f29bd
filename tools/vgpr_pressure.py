"""Live VGPRs per instruction of the production trace kernel, from its assembly (development aid).

    tools/vgpr_pressure.py [limit] [extra hipcc flags ...]      e.g.  tools/vgpr_pressure.py 96 -DRAYHS_WAVES_PER_EU=1

Builds rayhs_trace_exact with line tables, runs a backward liveness analysis over the physical VGPRs of its
instruction stream (CFG from the labels and branches of the .s file) and prints, per source line of trace_kernel.hip,
the highest number of live VGPRs reached on that line and the loop depth -- i.e. WHERE the kernel needs more registers
than an occupancy target allows (96 for five waves per SIMD).  A VALU write under a partial exec mask is treated as a
full definition, as the register allocator does (a value is only read under a subset of the mask it was written under).
"""
import collections, os, re, subprocess, sys

SRC = "/root/repo/rayhs_amd/csrc/trace_kernel.hip"
limit = int(sys.argv[1]) if len(sys.argv) > 1 else 96
extra = sys.argv[2:]
os.makedirs("/tmp/vp", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DRAYHS_KERNEL_SUFFIX=exact",
                "-gline-tables-only", "--cuda-device-only", "-save-temps", "-c", SRC, "-I/root/repo/rayhs_amd/csrc", "-o", "trace.co"] + extra,
               cwd="/tmp/vp", stderr=subprocess.DEVNULL, check=True)
text = open("/tmp/vp/trace_kernel-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("rayhs_trace_exact:"))
end = next(i for i, l in enumerate(text) if i > start and l.startswith("\t.section") or (i > start and l.startswith(".Lfunc_end")))
body = text[start + 1:end]

NO_DEF = re.compile(r"^(global_store|scratch_store|flat_store|buffer_store|ds_write|ds_store|global_atomic_(?!.*\bglc\b)|s_|v_cmp|v_cmpx|v_readlane|v_readfirstlane|v_nop|v_writelane)")
READS_DEST = re.compile(r"^(v_fmac|v_mac|v_writelane|v_pk_fmac|v_dot2c|v_movrel)")


def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m: return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


insts = []          # (opcode, defs, uses, line, depth, label-target or None, falls_through)
labels = {}
line = 0
depth = 0
for l in body:
    s = l.strip()
    m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
    if m: line = int(m.group(1)); continue
    m = re.match(r"(\.LBB\d+_\d+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        d = re.search(r"Depth=(\d+)", l)
        depth = int(d.group(1)) if d else 0
        continue
    if not s or s.startswith((".", ";", "//")): continue
    s = s.split(";")[0].strip()
    parts = s.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    toks = []
    for o in ops:
        o = re.sub(r"\b(abs|neg|sext)\(|\)|\||^-", "", o).strip()
        toks.append(o.split()[0] if o else o)
    defs, uses = [], []
    vr = [regs(t) for t in toks]
    if NO_DEF.match(op):
        for r in vr: uses += r
        if op.startswith("v_writelane"): defs += vr[0]
    else:
        if vr and vr[0]:
            defs += vr[0]
            if READS_DEST.match(op): uses += vr[0]
        for r in vr[1:]: uses += r
        if op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")) and len(vr) > 1 and not vr[1]:
            pass  # second destination is an SGPR pair / vcc
    target = None
    falls = True
    if op.startswith("s_cbranch"): target = toks[0]
    elif op == "s_branch": target, falls = toks[0], False
    elif op in ("s_endpgm", "s_setpc_b64"): falls = False
    insts.append((op, set(defs), set(uses), line, depth, target, falls))

n = len(insts)
succ = [[] for _ in range(n)]
for i, (op, d, u, ln, dp, tgt, falls) in enumerate(insts):
    if falls and i + 1 < n: succ[i].append(i + 1)
    if tgt in labels and labels[tgt] < n: succ[i].append(labels[tgt])
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]: out |= live_in[s_]
        new = insts[i][2] | (out - insts[i][1])
        if new != live_in[i]:
            live_in[i] = new; changed = True
per_line = collections.defaultdict(lambda: [0, 0, 0])   # line -> [max live, depth at max, instruction count]
peak = 0
for i in range(n):
    out = set()
    for s_ in succ[i]: out |= live_in[s_]
    live = len(live_in[i] | out | insts[i][1])
    peak = max(peak, live)
    e = per_line[insts[i][3]]
    e[2] += 1
    if live > e[0]: e[0], e[1] = live, insts[i][4]
print(f"{n} instructions, peak live VGPRs {peak}; source lines where more than {limit} are live (line: max live, loop depth, instructions):")
src = open(SRC).read().split("\n")
for ln in sorted(per_line):
    m, dp, c = per_line[ln]
    if m > limit: print(f"  {ln:5d}: {m:4d}  depth {dp}  n={c:3d}   {src[ln - 1].strip()[:110] if 0 < ln <= len(src) else ''}")
# pressure histogram by loop depth
by_depth = collections.defaultdict(int)
for i in range(n):
    by_depth[insts[i][4]] = max(by_depth[insts[i][4]], len(live_in[i]))
print("max live-in by loop depth:", dict(sorted(by_depth.items())))
if os.environ.get("VP_DUMP"):
    # registers live into the instruction at the given source line (first occurrence): where each is defined / used
    want = int(os.environ["VP_DUMP"])
    i0 = next(i for i in range(n) if insts[i][3] == want)
    print(f"live into first instruction of line {want} (#{i0}, {insts[i0][0]}): {len(live_in[i0])} registers")
    for r in sorted(live_in[i0]):
        ds = [i for i in range(n) if r in insts[i][1]]
        us = [i for i in range(n) if r in insts[i][2]]
        print(f"  v{r}: {len(ds)} defs at lines {sorted(set(insts[i][3] for i in ds))[:12]}; {len(us)} uses at lines {sorted(set(insts[i][3] for i in us))[:14]}")
