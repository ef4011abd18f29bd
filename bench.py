#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (and Mpixels/s) of the rayhs trace loop on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" renders one full frame of the workload: every rank traces its
interleaved 8-row stripes with the HIP kernel (scene resident in HBM, output in
an HBM buffer), then the rows are gathered to rank 0 (one RCCL gather) and
assembled.  `value` = rays of all ranks / wall time of K steps (max over ranks,
barrier + synchronize on both sides).  A ray = one scene query
(closestIntersection or shadowIntersection) that was actually
traced; rays per frame come from one untimed instrumented run of the same kernel
(`reference_rays_per_frame` also counts the shadow queries of the reference that
are provably irrelevant -- zero light term -- and therefore skipped).  The frame is fixed, so
scaling is "strong".

Extra objects on the JSON line:
  roofline     the roof that BINDS this kernel: instruction throughput.  Two units of a CU are priced with
               costs measured on this chip (tools/micro/issue_rates.hip) and the instruction counts of this very
               kernel source (committed rocprofv3 passes, or three short --pmc child runs of this script) over
               the launch's mean HIP-event duration measured here: the four vector pipes (2 / 4 / 16 cycles per
               wave64 instruction by class) and the one scalar unit all four SIMDs share (1.05 cycles per scalar
               instruction).  `bound` names the busier one, `frac` is its occupancy.  `hbm_contract` keeps
               SURVEY.md 8(d)'s figures: algorithmic bytes (32 B per box test, 36 B per triangle test, 40 B per
               plane/sphere test, 60 B per shaded hit, 12 B per pixel) / the same duration against the 8 TB/s
               HBM peak -- a work rate served by the caches (the scene is 13 MB), not a bandwidth.
  cpu_baseline the CPU oracle (a C port of the reference algorithm: exhaustive
               traversal, recursive shading) timed on this host on a bounded sample
               of the same frame.  Reported, not the target.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.chdir(ROOT)  # scene files name meshes relative to the working directory (src/Mesh.hs:218-221)

WORKLOADS = {
    # name: (scene, spp, description)
    "cornell": ("scenes/cornell_1080p_d0.json", 1, "Cornell box 1920x1080, 1 spp, maxDepth 0 = primary + shadow rays (BASELINE configs[1])"),
    "dragon_low": ("scenes/dragon_low_1080p.json", 4, "dragon_low.obj in Cornell box 1920x1080, 4 spp, maxDepth 3 (BASELINE configs[2])"),
    "dragon": ("scenes/dragon_1080p.json", 16, "dragon.obj (27,228 tris) in Cornell box 1920x1080, 16 spp, maxDepth 3 (BASELINE configs[3])"),
    "dragon4k": ("scenes/dragon_4k.json", 64, "dragon.obj in Cornell box 3840x2160, 64 spp, maxDepth 3 (BASELINE configs[4])"),
}
STRIPE_ROWS = 8
# frames enqueued at a time, each on its own scene handle and stream (see main): two when there is a gather to hide
# behind the next frame's trace (N > 1), one on a single GPU (two buy nothing there and blur a profiler's per-kernel rows).
# RAYHS_BENCH_IN_FLIGHT=3 is opt-in: a one-GPU rehearsal of a 1/8 share (tools/overlap_frames.py: 1.05 ms per frame with one in
# flight, 0.73 with two, 0.69 with three) says it pays at N = 8, but three handles per rank triple the workspace, frame and
# output allocations and no 8-GPU run has confirmed it yet.
_WORLD = int(os.environ.get("WORLD_SIZE", "1"))
IN_FLIGHT = int(os.environ.get("RAYHS_BENCH_IN_FLIGHT", "0")) or (2 if _WORLD > 1 else 1)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# The two instruction-throughput roofs of a CU (same guide: 256 CUs x 4 SIMDs, 2400 MHz max clock; unit costs measured by
# tools/micro/issue_rates.hip, profiles/r05/micro/issue_rates.txt -- SIMD busy span / instructions executed, residency measured):
#   vector pipe: a wave64 instruction occupies its SIMD for 2 cycles (fp32 add / mul / fma / min / max, v_mov: the guide's SIMD-32), 4 cycles
#                (fp64 add / mul / fma, integer, convert; compares and selects when they follow one another), 16 (fp64 rcp / rsq / sqrt);
#   scalar unit: ONE per CU, shared by its four SIMDs, 0.95 SALU / scalar-memory / branch instructions per cycle (1.05 cycles each).
# Vector and scalar instructions of different waves issue side by side: a launch is bounded by the busier of the two.
N_CUS, N_SIMDS, MAX_CLOCK_GHZ = 256, 1024, 2.4
VALU_COST = {"f32_arith": 2.0, "f64_arith": 4.0, "f64_trans": 16.0, "f32_trans": 8.0, "int_cvt": 4.0, "other_nominal": 2.0, "other_measured": 4.0}
SCALAR_COST_CU_CYCLES = 1.05


def algorithmic_bytes(st):
    return (32 * st["nodes_tested"] + 36 * st["tris_tested"] + 40 * st["prims_tested"] + 60 * st["shaded_hits"] + 12 * st["pixels"])


def _code_only(text):
    """C++ source without comments, blank lines and trailing blanks (string literals respected): what the compiler sees."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        # a literal: copied whole.  An apostrophe only opens a CHARACTER literal when it cannot be a C++14 digit separator (1'000'000) or
        # an apostrophe of unquoted preprocessor text (#error don't): the character before it is not alphanumeric and the literal closes within
        # a few characters on the same line; any other apostrophe is copied as it stands and the scanner stays in step.
        is_char = (c == "'" and not (i > 0 and (text[i - 1].isalnum() or text[i - 1] == "_"))
                   and re.match(r"'(\\.[^'\n]{0,8}|[^'\\\n])'", text[i:i + 12]) is not None)
        if c == '"' or is_char:
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
        elif text.startswith("/*", i):
            i = text.find("*/", i + 2)
            i = n if i < 0 else i + 2
        else:
            out.append(c); i += 1
    return "\n".join(ln.rstrip() for ln in "".join(out).split("\n") if ln.strip()) + "\n"


def kernel_source_sha256():
    """Identity of the kernel being benchmarked: what tools/profile_round.sh records next to its counters.  Taken over the CODE of the
    four files (comments and blank lines stripped): correcting a comment must not orphan the committed profiles."""
    import hashlib
    lines = []
    for f in ("rayhs_amd/csrc/trace_kernel.hip", "rayhs_amd/csrc/device_scene.h", "rayhs_amd/csrc/short_div.h", "rayhs_amd/csrc/abi.cpp"):
        lines.append(f"{hashlib.sha256(_code_only(open(os.path.join(ROOT, f)).read()).encode()).hexdigest()}  {f}\n")
    return hashlib.sha256("".join(lines).encode()).hexdigest()


def profile_counters(workload, kernel_ms):
    """Counter-based figures of the trace kernel from the committed rocprofv3 PMC passes of this workload
    (profiles/rNN/<workload>/summary.json; PMC counters cannot be read from inside this process).  They are used only
    when that profile was taken on THIS kernel source (sha256 recorded by tools/profile_round.sh); otherwise every
    counter-based field is null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", workload, "summary.json")))
    if not files:
        return {"profile": None, "matches_this_kernel": False}, None
    d = json.load(open(files[-1]))
    src = {"profile": os.path.relpath(files[-1], ROOT), "kernel_source_sha256": d.get("kernel_source_sha256"),
           "profile_kernel_ms": (d.get("timed_launches") or {}).get("avg_ms")}
    src["matches_this_kernel"] = (bool(src["kernel_source_sha256"]) and src["kernel_source_sha256"] == kernel_source_sha256()
                                  and not os.environ.get("RAYHS_BENCH_IGNORE_PROFILE"))  # (test hook: rehearse the live-counter fallback)
    c = next((v for k, v in d.get("pmc_mean_per_launch", {}).items() if "count" not in k), None)
    if not src["matches_this_kernel"] or not c or "hbm_bytes_per_launch_read_doubled" not in c:
        return src, None
    waves = ((d.get("dispatch") or {}).get("launch") or {}).get("waves_per_simd", 4)
    sec = kernel_ms * 1e-3
    return src, {
        "traffic": c["hbm_bytes_per_launch_read_doubled"],
        "hbm_traffic_frac": c["hbm_bytes_per_launch_read_doubled"] / sec / 1e9 / HBM_PEAK_GBS,
        "hbm_read_frac": 2 * c["hbm_read_bytes_raw"] / sec / 1e9 / HBM_PEAK_GBS,
        "hbm_write_bytes": c["hbm_write_bytes"],
        "l2_hit_rate": c.get("l2_hit_rate"),
        "valu_busy": c["SQ_ACTIVE_INST_VALU_per_wave_cycle"] * waves if "SQ_ACTIVE_INST_VALU_per_wave_cycle" in c else None,
        "valu_lane_util": c.get("valu_lane_utilisation"),
        "wave_wait_frac": c.get("SQ_WAIT_ANY_per_wave_cycle"),
        "salu_per_valu": c.get("salu_per_valu"),
        "valu_insts_per_launch": c.get("SQ_INSTS_VALU"),
        # round 3's instruction-mix pass: what the vector PIPE is occupied for (fp64 at half rate), what a wave does with its life
        "valu_f64_share_of_insts": c.get("valu_f64_share_of_insts"),
        "fp64_tflops_issued": (c["fp64_flops_per_launch"] / sec / 1e12) if "fp64_flops_per_launch" in c else None,
        "fp64_peak_tflops": 78.6,
        "wave_issuing_frac": c.get("SQ_ACTIVE_INST_ANY_per_wave_cycle"),
        "wave_waitcnt_frac": c.get("SQ_WAIT_INST_ANY_per_wave_cycle"),
        "branches_per_valu": c.get("branches_per_valu"),
        # round 5: the two roofs (tools/summarize_profiles.py): instruction counts by cost class
        "insts_per_launch": c.get("insts_per_launch"),
        "valu_class_counts": c.get("valu_class_counts"),
        "scalar_unit_insts_per_launch": c.get("scalar_unit_insts_per_launch"),
        "waves_per_simd": waves,
        "scalar_dcache_hit_rate": c.get("scalar_dcache_hit_rate"),
        "scalar_icache_hit_rate": c.get("scalar_icache_hit_rate"),
    }


def live_instruction_counts(workload, fma):
    """Fallback when no committed profile matches this kernel source: count the trace kernel's instructions NOW, with two
    short rocprofv3 --pmc passes of this very script as child processes (three: instruction classes, the fp64 mix, the fp32 / integer mix) (counters in their own runs, nothing else traced;
    `--pmc-child` keeps the children from doing the same).  Each pass gets at most 75 s (a pass takes ~15 s on a warm box, most
    of it the interpreter and torch coming up).  Returns (fields the two roofs need, None) or (None, why not)."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("RAYHS_BENCH_NO_LIVE_PMC"):
        return None, "live counter passes switched off (RAYHS_BENCH_NO_LIVE_PMC)"
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    passes = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
              ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"],
              ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"]]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="rayhs_pmc_", dir="/tmp")
    try:
        for i, counters in enumerate(passes):
            out = os.path.join(tmp, f"p{i}")
            cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--workload", workload,
                   "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-4k", "--pmc-child"] + (["--fma"] if fma else [])
            try:
                r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=75)
            except subprocess.TimeoutExpired:
                return None, f"counter pass {i} ({counters[0]}, ...) did not finish within 75 s"
            if r.returncode != 0:
                return None, f"counter pass {i} exited with {r.returncode}: {(r.stderr or '').strip()[-300:]}"
            acc = {}
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Kernel_Name"].startswith("rayhs_trace") and "count" not in row["Kernel_Name"]:
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in counters:
                if c not in acc:
                    return None, f"counter pass {i} wrote no row for {c} on the trace kernel"
                vals[c] = sum(acc[c]) / len(acc[c])
        insts = sum(vals[c] for c in passes[0])
        arith = vals["SQ_INSTS_VALU_ADD_F64"] + vals["SQ_INSTS_VALU_MUL_F64"] + vals["SQ_INSTS_VALU_FMA_F64"]
        trans = vals["SQ_INSTS_VALU_TRANS_F64"]
        f32a = vals["SQ_INSTS_VALU_ADD_F32"] + vals["SQ_INSTS_VALU_MUL_F32"] + vals["SQ_INSTS_VALU_FMA_F32"]
        intc = vals["SQ_INSTS_VALU_INT32"] + vals["SQ_INSTS_VALU_INT64"] + vals["SQ_INSTS_VALU_CVT"]
        classes = {"f32_arith": f32a, "f64_arith": arith, "f64_trans": trans, "f32_trans": vals["SQ_INSTS_VALU_TRANS_F32"], "int_cvt": intc,
                   "other": max(vals["SQ_INSTS_VALU"] - arith - trans - f32a - vals["SQ_INSTS_VALU_TRANS_F32"] - intc, 0.0)}
        return {"insts_per_launch": insts, "valu_class_counts": classes,
                "scalar_unit_insts_per_launch": vals["SQ_INSTS_SALU"] + vals["SQ_INSTS_SMEM"] + vals["SQ_INSTS_BRANCH"],
                "valu_f64_share_of_insts": (arith + trans) / vals["SQ_INSTS_VALU"], "salu_per_valu": vals["SQ_INSTS_SALU"] / vals["SQ_INSTS_VALU"],
                "branches_per_valu": vals["SQ_INSTS_BRANCH"] / vals["SQ_INSTS_VALU"], "valu_insts_per_launch": vals["SQ_INSTS_VALU"],
                "fp64_flops_per_launch": 64.0 * (vals["SQ_INSTS_VALU_ADD_F64"] + vals["SQ_INSTS_VALU_MUL_F64"] + 2 * vals["SQ_INSTS_VALU_FMA_F64"])}, None
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def golden_digest(workload):
    """sha256 of the oracle's whole fp32 frame of this workload (tests/golden/frame_digests.json, made once in the
    build container by tests/golden/make_frame_digests.py), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "frame_digests.json")))[workload]["sha256_f32"]
    except Exception:
        return None


def cpu_baseline(scene, spp, seconds, threads, frame=None):
    """Time the oracle on evenly spread single rows of the same frame for about `seconds`.  `frame` (numpy fp32
    [h, w, 3], the frame the GPU just rendered): the oracle's rows are also compared with it, bit for bit."""
    from oracle import oracle as O
    jd = O.load_render_desc(scene)
    sc = O.Scene(jd["scene"])
    cam = O.make_camera(jd["camera"])
    w, h, d = jd["width"], jd["height"], jd["maxDepth"]
    # probe two mid rows to size the sample
    t = time.perf_counter()
    _, st = sc.render(cam, w, h, d, spp=spp, row0=h // 2, row1=h // 2 + 2, nthreads=threads)
    per_row = (time.perf_counter() - t) / 2
    n = int(max(4, min(h, seconds / max(per_row, 1e-6))))
    rows = sorted({int((i + 0.5) * h / n) for i in range(n)})
    rays = 0
    images = []
    t = time.perf_counter()
    for y in rows:
        img, st = sc.render(cam, w, h, d, spp=spp, row0=y, row1=y + 1, nthreads=threads)
        rays += st["rays"]
        images.append(img)
    dt = time.perf_counter() - t
    checked = None
    if frame is not None:
        import numpy as np
        bad = [y for y, img in zip(rows, images) if not np.array_equal(img[0].astype(np.float32), frame[y])]
        checked = {"rows": len(rows), "bit_exact": not bad, "first_bad_rows": bad[:8]}
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "gpu_rows_checked": checked,
            "sample": f"{len(rows)} rows spread evenly over the {w}x{h} frame at {spp} spp ({rays} rays in {dt:.1f} s); "
                      "C restatement of the reference algorithm (oracle/), OpenMP over pixels",
            "mpixels_per_s": len(rows) * w / dt / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--fma", action="store_true", help="use the fused-multiply-add kernels (not bit-exact with the reference)")
    ap.add_argument("--no-4k", dest="also_4k", action="store_false",
                    help="skip the extra (untimed for `value`) leg on BASELINE configs[4], the 4K / 64 spp frame")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # a counter pass of live_instruction_counts(): no nested passes
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 also renders the whole frame by itself and checks that the gathered frame is identical")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rayhs_amd import RenderDesc, Rendering, lib as L, tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if L.load().rayhs_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the trace path has no CPU fallback")
    # Test hooks (tests/test_gpu_parity.py): rehearse the N > 1 path on a one-GPU box -- every rank on
    # device 0, gloo instead of RCCL.  Never set by the driver.
    if os.environ.get("RAYHS_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("RAYHS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        ranks_seen = dist.get_world_size()
        if ranks_seen != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {ranks_seen} ranks")

    def run_workload(workload, steps, warmup, want_first_frame=True):
        """Set-up, warm-up and the timed loop of one workload; returns its local variables."""
        scene, spp, desc = WORKLOADS[workload]
        rd = RenderDesc(scene)
        # Two frames in flight: each slot has its own scene handle (13 MB of HBM, its own launch workspace and tile order),
        # its own stream and its own gather buffers.  A persistent kernel ends with its slowest wave, so the last part of
        # frame i leaves most of the chip idle; frame i + 1, enqueued on the other stream, starts in those idle CUs, and the
        # gather of frame i (RCCL's stream) runs beside it.  (tools/overlap_frames.py, per frame: a 1/8 share of the 1080p
        # frame 1.39 ms alone, 1.08 with two in flight; the whole frame 7.62 -> 7.58 ms.  Three in flight: 0.98 ms for the
        # share in that tool, but in this loop the whole frame gets slower, 7.57 -> 7.8 ms: two trace kernels then run
        # side by side for most of their time instead of tail to head.)
        renderings = [Rendering(rd, device=local_rank) for _ in range(IN_FLIGHT)]
        r = renderings[0]
        w, h = r.width, r.height
        slots = [tiles.FrameGather(rank, world, h, w, STRIPE_ROWS, dev, torch.float32) for _ in range(IN_FLIGHT)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(IN_FLIGHT)]
        g = slots[0]
        flags = L.FLAG_FMA if args.fma else 0
        kw = dict(spp=spp, seed=24, stripe_rows=STRIPE_ROWS, stripe_rank=rank, stripe_count=world, out_format=L.OUT_F32)
        stream = torch.cuda.current_stream()

        pending = []

        def step(i, ev=None):
            j = i % IN_FLIGHT
            cur = slots[j]
            with torch.cuda.stream(streams[j]):  # the collective waits for, and the assembly runs on, the slot's stream
                # the library records the two HIP events on the stream right around the trace kernel itself
                renderings[j].ray_trace_device(cur.local.data_ptr(), streams[j].cuda_stream, flags=flags,
                                               events=(ev[0].cuda_event, ev[1].cuda_event) if ev else None, **kw)
                cur.gather_start()
            pending.append((j, cur))
            if len(pending) == IN_FLIGHT:  # the oldest frame in flight: its transfer ran beside the traces enqueued since
                k, prev = pending.pop(0)
                with torch.cuda.stream(streams[k]):
                    prev.gather_finish()

        def drain():
            while pending:
                k, prev = pending.pop(0)
                with torch.cuda.stream(streams[k]):
                    prev.gather_finish()

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # one instrumented (untimed) run of the same kernel: rays and algorithmic bytes of a frame
        counters = torch.zeros(16, dtype=torch.int64, device=dev)
        r.ray_trace_device(g.local.data_ptr(), stream.cuda_stream, counters.data_ptr(), flags=flags | L.FLAG_COUNT, **kw)
        torch.cuda.synchronize()
        names = ["rays_primary", "rays_reflect", "rays_refract", "rays_refract_exit", "rays_shadow", "nodes_tested", "tris_tested",
                 "prims_tested", "shaded_hits", "pixels", "shadow_skipped"]
        counters[9] //= spp  # the kernel counts finished samples; pixels = samples / spp
        mine = dict(zip(names, counters[:11].tolist()))
        total = counters.clone()
        if world > 1:
            dist.all_reduce(total)
        tot = dict(zip(names, total[:11].tolist()))
        # rays_* count the REFERENCE's scene queries of the frame; shadow queries whose light term is exactly zero are not
        # traced at all (trace_kernel.hip, light section) and are NOT counted in `value`
        reference_rays_frame = sum(v for k, v in tot.items() if k.startswith("rays_"))
        rays_frame = reference_rays_frame - tot["shadow_skipped"]

        # set-up, like the instrumented run above: the first frame a scene handle renders runs in scan order and measures
        # the chunk times that order every later frame; the warm-up and timed steps are steady state
        for j in range(IN_FLIGHT):
            renderings[j].ray_trace_device(slots[j].local.data_ptr(), streams[j].cuda_stream, flags=flags, **kw)
            torch.cuda.synchronize()
        # ... and the clocks: a GPU that has been idle runs its first ~40 ms of work below its sustained clock (consecutive frames of one
        # handle, tools/frame_times.py: 1.23, 1.23, 1.22, 1.20, 1.18, 1.16, 1.15, 1.14, 1.13 ms on dragon_low).  Set-up keeps rendering
        # until 60 ms of frames have gone by (at most 400 frames), so that warm-up and the timed steps see the clock a running host sees.
        t_settle = time.perf_counter()
        settle_frames = 0
        for i in range(400):
            j = i % IN_FLIGHT
            renderings[j].ray_trace_device(slots[j].local.data_ptr(), streams[j].cuda_stream, flags=flags, **kw)
            settle_frames += 1
            if i % 4 == 3:
                torch.cuda.synchronize()
                if time.perf_counter() - t_settle > 0.060:
                    break
        torch.cuda.synchronize()
        for i in range(warmup):
            step(i)
        drain()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for i, (a, b) in enumerate(events):  # torch creates the underlying hipEvent lazily, on first record
            a.record(streams[i % IN_FLIGHT])
            b.record(streams[i % IN_FLIGHT])
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, events[i])
        drain()  # every frame is gathered and assembled inside the timed region
        sync_all()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))
        # N > 1: what the collective layer saw, for whoever reads the line of a scaling run -- every rank's mean trace-kernel time and the
        # time of one gather + assembly by itself (three in a row, nothing else in flight; outside the timed region)
        kernel_ms_per_rank, gather_ms = [kernel_ms], None
        if world > 1:
            km = torch.zeros(world, dtype=torch.float64, device=dev)  # (an all-reduce of one-hot rows: every backend the rehearsals use has it for device tensors)
            km[rank] = kernel_ms
            dist.all_reduce(km)
            kernel_ms_per_rank = km.tolist()
            sync_all()
            tg = time.perf_counter()
            for _ in range(3):
                with torch.cuda.stream(streams[0]):
                    slots[0].gather()
            sync_all()
            gather_ms = (time.perf_counter() - tg) / 3 * 1e3

        # The first frame of a handle -- the only frame a one-shot `rayhs -oX scene.json` ever renders (src/RayHs.hs:215-234)
        # -- has no measured chunk times to order its tiles by and uses the static estimate instead.  Timed (untimed for
        # `value`) on a FRESH handle now that the kernel's code is resident and the clocks are up: first_frame_ms is the
        # scheduling effect alone.
        first_frame_ms = None
        if not want_first_frame:
            return locals()
        fresh = Rendering(rd, device=local_rank)
        ff = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for e in ff:
            e.record(streams[0])
        torch.cuda.synchronize()
        fresh.ray_trace_device(slots[0].local.data_ptr(), streams[0].cuda_stream, flags=flags, events=(ff[0].cuda_event, ff[1].cuda_event), **kw)
        torch.cuda.synchronize()
        first_frame_ms = ff[0].elapsed_time(ff[1])
        fresh.close()
        return locals()

    m = run_workload(args.workload, args.steps, args.warmup)
    scene, spp, desc, r, w, h, slots, stream = m["scene"], m["spp"], m["desc"], m["r"], m["w"], m["h"], m["slots"], m["stream"]
    mine, rays_frame, reference_rays_frame = m["mine"], m["rays_frame"], m["reference_rays_frame"]
    dt, kernel_ms, first_frame_ms, flags = m["dt"], m["kernel_ms"], m["first_frame_ms"], m["flags"]

    # north_star quotes its 1 -> 8 GPU scaling target on the 4K tile-sharded render (BASELINE configs[4]); the headline
    # `value` stays on configs[3].  The same loop on that frame, a few steps, every rank taking part: lets the driver's
    # scaling run show both frames.  Not part of `value`, `ms_per_step` or the roofline.
    also_4k, launches_after_timed = None, 1 + (1 if args.verify else 0)
    if args.also_4k and args.workload != "dragon4k":
        x = run_workload("dragon4k", 3, 1, want_first_frame=False)
        launches_after_timed += IN_FLIGHT + x["settle_frames"] + 1 + 3   # its set-up, clock-settling, warm-up and timed frames
        if rank == 0:
            import hashlib
            f4 = x["slots"][(3 - 1) % IN_FLIGHT].frame
            want4 = None if args.fma else golden_digest("dragon4k")
            ok4 = None if want4 is None else hashlib.sha256(f4.cpu().numpy().tobytes()).hexdigest() == want4
            if ok4 is False:
                raise SystemExit("bench.py: the 4K frame differs from the oracle's frame digest (tests/golden/frame_digests.json)")
            also_4k = {"workload": x["desc"], "steps": 3, "warmup": 1, "ms_per_step": x["dt"] / 3 * 1e3, "kernel_ms_rank0": x["kernel_ms"],
                       "value": x["rays_frame"] * 3 / x["dt"] / 1e6, "unit": "Mrays/s", "rays_per_frame": x["rays_frame"],
                       "whole_frame_sha256_equals_oracle_digest": ok4}
        for q in x["renderings"]:
            q.close()
        del x

    if rank == 0:
        frame = slots[(args.steps - 1) % IN_FLIGHT].frame
        assert torch.isfinite(frame).all() and frame.shape == (h, w, 3)
        verified = None
        if args.verify:
            whole = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
            r.ray_trace_device(whole.data_ptr(), stream.cuda_stream, flags=flags, spp=spp, seed=24, out_format=L.OUT_F32)
            torch.cuda.synchronize()
            verified = bool(torch.equal(whole, frame))
            if not verified:
                raise SystemExit("bench.py --verify: the gathered frame differs from the single-rank render")
        # parity of the frame the timed loop produced: (a) sha256 against the oracle's whole frame (committed digest),
        # (b) below, the rows the CPU-baseline leg renders anyway, bit for bit.  A mismatch fails the run.
        host_frame = frame.cpu().numpy() if world >= 1 else None
        import hashlib
        want_sha = None if args.fma else golden_digest(args.workload)
        golden_ok = None if want_sha is None else hashlib.sha256(host_frame.tobytes()).hexdigest() == want_sha
        if golden_ok is False:
            raise SystemExit("bench.py: the rendered frame differs from the oracle's frame digest (tests/golden/frame_digests.json)")
        alg = algorithmic_bytes(mine)
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        # roofline = the roof that binds (instruction issue, frac <= 1); hbm_contract inside it = SURVEY.md 8(d)'s line:
        # algorithmic bytes per launch / kernel time against the HBM peak.  The scene is cache-resident, so that figure is
        # served by L1 / scalar cache / L2 and exceeds the HBM peak: a work rate, not a bandwidth.
        src, ctr = profile_counters(args.workload, kernel_ms) if world == 1 else ({"profile": None, "matches_this_kernel": False}, None)
        if ctr is None and world == 1 and not args.pmc_child:
            # no committed profile of THIS kernel source: count its instructions now (three short rocprofv3 --pmc child runs);
            # the HBM / cache / wave-life figures, which need more passes, stay null
            live, why_not = live_instruction_counts(args.workload, args.fma)
            if not live:
                src = {**src, "live_pmc_passes_in_this_run": False, "live_pmc_failure": why_not}
            if live:
                names = ("traffic", "hbm_traffic_frac", "hbm_read_frac", "hbm_write_bytes", "l2_hit_rate", "valu_busy", "valu_lane_util", "wave_wait_frac",
                         "wave_issuing_frac", "wave_waitcnt_frac", "scalar_dcache_hit_rate", "scalar_icache_hit_rate")
                ctr = {**{k: None for k in names}, "fp64_peak_tflops": 78.6, "waves_per_simd": 4, **live,
                       "fp64_tflops_issued": live["fp64_flops_per_launch"] / (kernel_ms * 1e-3) / 1e12}
                src = {**src, "live_pmc_passes_in_this_run": True}
        sec = kernel_ms * 1e-3
        cls = ctr and ctr.get("valu_class_counts")
        valu = scalar = None
        if cls:
            base = sum(VALU_COST[k] * cls[k] for k in ("f32_arith", "f64_arith", "f64_trans", "f32_trans", "int_cvt"))
            peak = N_SIMDS * MAX_CLOCK_GHZ
            valu = {"unit": "G SIMD-cycles/s", "peak": peak,
                    "achieved": (base + VALU_COST["other_nominal"] * cls["other"]) / sec / 1e9,
                    "achieved_measured_costs": (base + VALU_COST["other_measured"] * cls["other"]) / sec / 1e9}
            valu["frac"] = valu["achieved"] / peak
            valu["frac_measured_costs"] = valu["achieved_measured_costs"] / peak
        if ctr and ctr.get("scalar_unit_insts_per_launch"):
            peak = N_CUS * MAX_CLOCK_GHZ
            scalar = {"unit": "G CU-cycles/s", "peak": peak, "achieved": SCALAR_COST_CU_CYCLES * ctr["scalar_unit_insts_per_launch"] / sec / 1e9}
            scalar["frac"] = scalar["achieved"] / peak
        binding = None
        if valu and scalar:
            binding = ("valu", valu) if valu["frac"] >= scalar["frac"] else ("scalar-issue", scalar)
        # (without instruction counts -- no committed profile of this kernel source and no live passes -- no roof is named: `bound` is null)
        roofline = {"bound": binding[0] if binding else None, "achieved": binding[1]["achieved"] if binding else None,
                    "peak": binding[1]["peak"] if binding else N_CUS * MAX_CLOCK_GHZ, "unit": binding[1]["unit"] if binding else "G CU-cycles/s",
                    "frac": binding[1]["frac"] if binding else None,
                    "traffic": ctr["traffic"] if ctr else None,
                    "model": "the larger of two instruction-throughput roofs, both priced with unit costs measured on this chip (tools/micro/issue_rates.hip, "
                             "profiles/r05/micro/issue_rates.txt) and instruction counts of this kernel source (rocprofv3 SQ_INSTS_* passes) over this run's mean "
                             "HIP-event duration at 2.4 GHz: 'valu' = cycles the four vector pipes of a CU are occupied (2 per fp32 add/mul/fma and per instruction no "
                             "class counter covers, 4 per fp64 add/mul/fma / integer / convert, 16 per fp64 rcp/rsq/sqrt); 'scalar-issue' = the ONE scalar unit of a CU, "
                             "1.05 cycles per SALU / scalar-memory / branch instruction of any of its 16 waves",
                    "valu_pipe": valu, "scalar_unit": scalar,
                    "valu_class_counts": cls, "scalar_unit_insts_per_launch": ctr and ctr.get("scalar_unit_insts_per_launch"),
                    "insts_per_launch": ctr and ctr.get("insts_per_launch"),
                    "bound_actual": "latency chains under issue contention: the scalar unit of a CU (shared by its 16 resident waves) and the vector pipes are both about 60 % "
                                    "busy (valu_pipe.frac_measured_costs prices compares, selects and lane reads at the 4 cycles they take back to back, an upper bound), a wave "
                                    "issues 37 % of its life and is parked at s_waitcnt 39 %; with 4 waves per SIMD (128 VGPRs of fp64 state, 40 KiB of LDS: both exhausted) the "
                                    "frame's time follows the length of every wave's chain of dependent steps -- measured on the packet walk: +1 dependent scalar-cache hit per visit "
                                    "+1.0 %, +4 SALU per visit +0.5 %, +4 VALU +0.3 % (profiles/r06/experiments/); HBM (3 % of peak, the scene stays in cache) is nowhere near "
                                    "(DESIGN.md section 4)",
                    "scalar_dcache_hit_rate": ctr and ctr.get("scalar_dcache_hit_rate"), "scalar_icache_hit_rate": ctr and ctr.get("scalar_icache_hit_rate"),
                    # SURVEY.md 8(d)'s contract figures: algorithmic bytes / kernel time against the HBM peak (> 1: the bytes are served by the caches)
                    "hbm_contract": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_launch": alg, "note": "a work rate, not a bandwidth: the 13 MB scene is L2 / scalar-cache resident"},
                    "hbm_traffic_frac": ctr and ctr["hbm_traffic_frac"], "hbm_read_frac": ctr and ctr["hbm_read_frac"],
                    "hbm_write_bytes": ctr and ctr["hbm_write_bytes"], "l2_hit_rate": ctr and ctr["l2_hit_rate"],
                    "valu_busy": ctr and ctr["valu_busy"], "valu_lane_util": ctr and ctr["valu_lane_util"],
                    "wave_wait_frac": ctr and ctr["wave_wait_frac"], "salu_per_valu": ctr and ctr["salu_per_valu"],
                    "valu_insts_per_launch": ctr and ctr["valu_insts_per_launch"],
                    **({k: ctr.get(k) for k in ("valu_f64_share_of_insts", "fp64_tflops_issued", "fp64_peak_tflops", "wave_issuing_frac",
                                                "wave_waitcnt_frac", "branches_per_valu")} if ctr else {}),
                    "counters_from": src}
        out = {
            "metric": "Mrays/s", "value": rays_frame * args.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (the reference's own Cornell-box/dragon sample scene; no dataset)",
            "config": {"workload": desc, "scene": scene, "width": w, "height": h, "spp": spp, "max_depth": r.max_depth,
                       "frames_in_flight": IN_FLIGHT, "sharding": f"{STRIPE_ROWS}-row stripes interleaved over {world} rank(s), one gather to rank 0",
                       "kernels": "fma" if args.fma else "exact (-ffp-contract=off, bit-identical to the CPU oracle)",
                       "scheduling": "tiles handed out slowest-first, from the chunk times earlier frames measured (ordering kernels behind the first four frames of a view, then behind every fourth: RAYHS_REORDER_EVERY)"},
            "mpixels_per_s": w * h * args.steps / dt / 1e6,
            "rays_per_frame": rays_frame, "reference_rays_per_frame": reference_rays_frame,
            "kernel_ms_rank0": kernel_ms, "first_frame_ms": first_frame_ms,
            # what the collective layer saw (N > 1: the backend torch.distributed really runs -- "nccl" is RCCL on ROCm -- the ranks of its
            # process group, every rank's mean trace-kernel time, one gather + assembly by itself)
            "distributed": {"backend": (dist.get_backend() if world > 1 else None), "ranks_seen": ranks_seen, "devices_visible": torch.cuda.device_count(),
                            "kernel_ms_per_rank": m["kernel_ms_per_rank"], "gather_ms": m["gather_ms"]},
            "north_star_4k": also_4k, "trace_launches_after_timed": launches_after_timed,
            "verified_against_single_rank": verified,
            "verified_against_oracle": {"whole_frame_sha256_equals_oracle_digest": golden_ok, "rows_against_live_oracle": None},
            "roofline": roofline,
        }
        if world == 1 and args.cpu_seconds > 0:
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            out["cpu_baseline"] = cpu_baseline(scene, spp, args.cpu_seconds, threads, None if args.fma else host_frame)
            chk = out["cpu_baseline"].pop("gpu_rows_checked")
            out["verified_against_oracle"]["rows_against_live_oracle"] = chk
            if chk and not chk["bit_exact"]:
                raise SystemExit(f"bench.py: GPU frame differs from the live oracle on rows {chk['first_bad_rows']}")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
