"""Diagnostic (needs a -DRAYHS_PROFILE_TIME build selected with RAYHS_LIB): how long waves idle at the end of a frame.

    make -C rayhs_amd/csrc KDEFS=-DRAYHS_PROFILE_TIME BUILD=../../build_exp/o2 BIN=../../build_exp/bin2
    RAYHS_LIB=build_exp/bin2/librayhs_hip.so python tools/wave_tail.py [scene] [spp] [world]
(world > 1: rank 0's share of a frame cut into interleaved 8-row stripes)
Renders the frame three times (the third uses the tile order measured by the second) and prints, per frame, the
kernel span, the mean life of a wave and the mean idle tail (latest end - mean end) from the library's debug line.
"""
import os, re, subprocess, sys
if os.environ.get("RAYHS_WAVE_TAIL_CHILD"):
    sys.path.insert(0, ".")
    from rayhs_amd import build_rendering, lib as L
    r = build_rendering(sys.argv[1])
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for _ in range(3):
        r.ray_trace(spp=int(sys.argv[2]), flags=L.FLAG_COUNT, out_format=L.OUT_F32, stripe_rows=8, stripe_rank=0, stripe_count=world)
    sys.exit(0)
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/dragon_1080p.json"
spp = sys.argv[2] if len(sys.argv) > 2 else "16"
env = dict(os.environ, RAYHS_WAVE_TAIL_CHILD="1", RAYHS_DEBUG_COUNTERS="1")
world = sys.argv[3] if len(sys.argv) > 3 else "1"
out = subprocess.run([sys.executable, __file__, scene, spp, world], env=env, capture_output=True, text=True).stderr
for l in out.splitlines():
    m = re.search(r"span_ticks (\d+) latest_end (\d+) earliest_start (\d+) counters\[6..15\]: (\d+) (\d+) \| (\d+) (\d+)", l)
    if m:
        span, le, es, waves, chunks, life, ends = map(int, m.groups())
        print(f"span {span / 1e5:.3f} ms  waves {waves}  chunks {chunks}  mean life {life / waves / 1e5:.3f} ms  mean idle tail {(le - ends / waves) / 1e5:.3f} ms")
    m = re.search(r"adv\[0..3\] (\d+) (\d+) (\d+) (\d+) .*counters\[6..15\]: \d+ \d+ \| \d+ \d+ \| (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", l)
    if m:
        v = list(map(int, m.groups()))
        names = ["refill+loop top", "hit/shadow/exit sections", "light section", "specular section", "return/new sample", "begin (planes, spheres)",
                 "  slice: mesh start", "  slice: node loop", "  slice: leaves+pop", "slice total"]
        total = sum(v[:6]) + v[9]
        print("   " + "; ".join(f"{n} {100 * x / total:.1f}%" for n, x in zip(names, v)))
