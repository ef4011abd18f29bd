#!/usr/bin/env python3
"""GPU side of the basic-block profile (development aid):   python3 tools/bbprof_run.py [workload ...]   -> gpurun_out/bbprof_<workload>.json

Renders steady-state frames of the workload with the library tools/build_asm_rewrite.sh bbprof tools/bbprof_rewrite.py built (every
straight-line segment of rayhs_trace_exact counts its executions per wave with one s_atomic_add), checks the frame against the oracle's
digest, decodes the per-wave counters (each holds count x high word of the wave's own counter area) and writes the per-segment totals."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
os.environ.setdefault("RAYHS_LIB", os.path.join(ROOT, "build_exp", "bbprof", "librayhs_hip.so"))
import numpy as np
import torch
from rayhs_amd import build_rendering, lib as L

W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4),
     "cornell": ("scenes/cornell_1080p_d0.json", 1), "dragon4k": ("scenes/dragon_4k.json", 64)}
bbmap = json.load(open(os.path.join(os.path.dirname(os.environ["RAYHS_LIB"]), "bbmap.json")))
STRIDE, NSEG = bbmap["stride_segments"], len(bbmap["segments"])
WAVES = 8192   # more than any grid this library launches (CUs x 4 workgroups x 4 waves = 4096)
TAG = os.path.basename(os.path.dirname(os.environ["RAYHS_LIB"]))   # bbprof, or the experiment build's directory (output: gpurun_out/<TAG>_<workload>.json)
digests = json.load(open("tests/golden/frame_digests.json"))
os.makedirs("gpurun_out", exist_ok=True)
for name in (sys.argv[1:] or ["dragon"]):
    scene, spp = W[name]
    r = build_rendering(scene)
    out = torch.zeros((r.height, r.width, 3), dtype=torch.float32, device="cuda:0")
    buf = torch.zeros(WAVES * STRIDE, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    ms = []
    for i in range(4):   # the first frames of a handle order their tiles (static estimate, then measured times); the last one is counted
        buf.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record(); torch.cuda.synchronize()
        r.ray_trace_device(out.data_ptr(), stream, buf.data_ptr(), events=(e0.cuda_event, e1.cuda_event), spp=spp, seed=24, flags=0, out_format=L.OUT_F32)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
        print(f"bbprof {name}: frame {i} {ms[-1]:.1f} ms", flush=True)
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()
    same = sha == digests[name]["sha256_f32"]
    raw = buf.cpu().numpy().view(np.uint32).reshape(WAVES, STRIDE).astype(np.uint64)
    base = buf.data_ptr()
    hi = np.array([((base + w * STRIDE * 4) >> 32) for w in range(WAVES)], dtype=np.uint64)
    assert (hi > 0).all()
    used = raw[:, :NSEG]
    rem = used % hi[:, None]
    counts = used // hi[:, None]
    waves_used = int((counts[:, 0] > 0).sum())
    res = {"workload": name, "scene": scene, "spp": spp, "frame_ms_instrumented": ms, "frame_sha256_equals_oracle_digest": bool(same),
           "every_counter_a_multiple_of_its_wave_high_word": bool((rem == 0).all()), "waves": waves_used,
           "segment_counts": counts.sum(axis=0).tolist()}
    json.dump(res, open(f"gpurun_out/{TAG}_{name}.json", "w"))
    print(f"bbprof {name}: {waves_used} waves, {int(counts.sum())} segment executions, frame == oracle digest: {same}, counters exact multiples: {res['every_counter_a_multiple_of_its_wave_high_word']}", flush=True)
    r.close()
