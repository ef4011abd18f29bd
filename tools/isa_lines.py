"""Static instruction counts of the production trace kernel per SOURCE LINE RANGE (development aid).

    tools/isa_lines.py [asm]        (asm: hipcc ... -gline-tables-only -save-temps output; built into /tmp/kl if absent)
Attributes every instruction of rayhs_trace_exact to the innermost trace_kernel.hip line of its .loc (inlined-at chains
are ignored: the .loc line is where the instruction's expression was written) and prints VALU / SALU / memory counts for
the sections of the kernel -- a static mix, to be weighed with the trip counts of a -DRAYHS_PROFILE_LOOPS run."""
import collections, os, re, subprocess, sys
SRC = "/root/repo/rayhs_amd/csrc/trace_kernel.hip"
asm = next((a for a in sys.argv[1:] if a.endswith(".s")), None)
if asm is None:
    os.makedirs("/tmp/kl", exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DRAYHS_KERNEL_SUFFIX=exact",
                    "-gline-tables-only", "--cuda-device-only", "-save-temps", "-c", SRC, "-I/root/repo/rayhs_amd/csrc", "-o", "trace.co"],
                   cwd="/tmp/kl", stderr=subprocess.DEVNULL)
    asm = "/tmp/kl/trace_kernel-hip-amdgcn-amd-amdhsa-gfx950.s"
OBJ = os.path.join(os.path.dirname(asm), "trace_kernel-hip-amdgcn-amd-amdhsa-gfx950.out")
LLVM = "/opt/rocm/lib/llvm/bin/"
# the kernel's address range, its instructions, and for every instruction the inlined-at chain of its source position
sym = subprocess.run([LLVM + "llvm-objdump", "-t", OBJ], capture_output=True, text=True).stdout
start = size = None
for ln in sym.split("\n"):
    f = ln.split()
    if f and f[-1] == "rayhs_trace_exact" and "F" in f: start, size = int(f[0], 16), int(f[f.index(".text") + 1], 16)
dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", f"--start-address={start:#x}", f"--stop-address={start + size:#x}", OBJ],
                     capture_output=True, text=True).stdout
insts = []
for ln in dis.split("\n"):
    m = re.match(r"\s+([a-z_0-9]+)\b.*// ([0-9A-F]+):", ln)
    if m: insts.append((int(m.group(2), 16), m.group(1)))
out = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + OBJ, "--inlines", "--functions=short"], input="\n".join(hex(a) for a, _ in insts) + "\n",
                     capture_output=True, text=True).stdout
chains = [c.strip().split("\n") for c in out.strip().split("\n\n")]
assert len(chains) == len(insts), (len(chains), len(insts))
cnt = collections.defaultdict(lambda: [0, 0, 0])
vclass = collections.defaultdict(collections.Counter)
listing = []   # (address, line, opcode) for --dump
for (addr, op), ch in zip(insts, chains):
    frames = [(ch[i], ch[i + 1]) for i in range(0, len(ch) - 1, 2)]   # innermost first: (function, file:line:col)
    line = -1
    # the section an instruction belongs to = where it sits in begin() / traverse() if it was inlined through one of
    # them, else where it sits in trace_body() (helpers like add(), dot(), ws_store() are charged to their caller)
    for want in (("begin", "traverse"), ("trace_body",)):
        for fn, pos in frames:
            if fn.split("<")[0] in want and "trace_kernel.hip" in pos:
                line = int(pos.rsplit(":", 2)[1]); break
        if line >= 0: break
    k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith(("global_", "ds_", "scratch_", "flat_", "buffer_")) else None
    listing.append((addr, line, op))
    if k is None: continue
    cnt[line][k] += 1
    if k == 0:   # what the vector pipe is asked for, by cost class (profiles/r05/micro/issue_rates.txt): fp32 arithmetic and v_mov_b32 2 cycles, the rest 4, fp64 rcp / rsq / sqrt 16
        c = ("trans64" if re.match(r"v_(rcp|rsq|sqrt)_f64", op) else "cmp" if op.startswith("v_cmp") else "f64" if re.search(r"_f64", op) and not op.startswith("v_cvt") else
             "cvt" if op.startswith("v_cvt") else "f32" if re.search(r"_f32", op) else "cndmask" if op.startswith("v_cndmask") else
             "mov" if op.startswith(("v_mov_b32", "v_accvgpr")) else "lane" if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) else "int")
        vclass[line][c] += 1
src = open(SRC).read().split("\n")
def find(pat, start=0):
    for i in range(start, len(src)):
        if pat in src[i]: return i + 1
    raise KeyError(pat)
marks = [("(unattributed)", 1),
         ("begin(): preamble", find("__device__ void begin(")),
         ("begin(): plane loops", find("auto plane_range = ")),
         ("begin(): spheres", find("const auto spheres = (cptr<DSphere>)")),
         ("begin(): reference-tree hand-off", find("bool occluded = __builtin_amdgcn_inverse_ballot_w64(m_occ);")),
         ("meshes_reference()", find("__device__ __forceinline__ bool meshes_reference(")),
         ("visit() (per-lane node)", find("template <class ND> __device__ __forceinline__ void visit(")),
         ("traverse(): set-up, root cull", find("__device__ void traverse(bool shadow")),
         ("traverse(): packet walk", find("const stack_ptr wstack = stack")),
         ("traverse(): per-lane walk", find("if (enter) w.t_cur = me.froot;")),
         ("workspace / context helpers", find("// ---- global per-lane workspace")),
         ("trace_body(): set-up", find("static __device__ __forceinline__ void trace_body()")),
         ("main loop: refill + new sample", find("if (__ballot(state != ST_DONE) == 0ull)")),
         ("hit section", find("if (state == ST_HIT) {")),
         ("shadow-return section", find("} else if (state == ST_SHADOW) {")),
         ("exit section", find("} else if (state == ST_EXIT) {")),
         ("light section", find("if (step == A_LIGHT_NEXT) {")),
         ("specular section", find("if (step == A_SPEC) {")),
         ("return section", find("if (step == A_RETURN) {")),
         ("query site + epilogue", find("// ---- the single scene-query site")),
         ("end", len(src) + 1)]
print(f"{'section':52s} {'VALU':>6s} {'SALU':>6s} {'mem':>5s}")
tot = [0, 0, 0]
for (name, a), (_, b) in zip(marks, marks[1:]):
    s = [sum(cnt[l][k] for l in cnt if a <= l < b) for k in range(3)]
    for k in range(3): tot[k] += s[k]
    print(f"{name:52s} {s[0]:6d} {s[1]:6d} {s[2]:5d}")
o = cnt.get(-1, [0, 0, 0])
print(f"{'(other files: hip headers, libm)':52s} {o[0]:6d} {o[1]:6d} {o[2]:5d}")
print(f"{'total':52s} {tot[0] + o[0]:6d} {tot[1] + o[1]:6d} {tot[2] + o[2]:5d}")

if "--classes" in sys.argv:   # static VALU instructions per section by cost class
    names = ("f64", "trans64", "f32", "mov", "cmp", "cndmask", "cvt", "lane", "int")
    print("\n" + f"{'section (static VALU by class)':52s} " + " ".join(f"{n:>7s}" for n in names) + "   pipe-cycles")
    cost = {"f64": 4, "trans64": 16, "f32": 2, "mov": 2, "cmp": 4, "cndmask": 4, "cvt": 4, "lane": 4, "int": 4}
    for (name, a), (_, b) in zip(marks, marks[1:]):
        c = collections.Counter()
        for l in vclass:
            if a <= l < b: c.update(vclass[l])
        if sum(c.values()): print(f"{name:52s} " + " ".join(f"{c[n]:7d}" for n in names) + f"   {sum(cost[n] * c[n] for n in names):7d}")

if "--dump" in sys.argv:   # tools/isa_lines.py asm --dump "<section name prefix>": the section's instructions in address order, with their line
    want = sys.argv[sys.argv.index("--dump") + 1]
    full = {}
    for ln in dis.split("\n"):
        m = re.search(r"// ([0-9A-F]+):", ln)
        if m: full[int(m.group(1), 16)] = ln.split("//")[0].rstrip()
    for (name, a), (_, b) in zip(marks, marks[1:]):
        if not name.startswith(want): continue
        for addr, line, op in listing:
            if a <= line < b: print(f"{addr:06x} L{line:<5d} {full.get(addr, op)}")
