"""Per-phase share of a wave's life from a -DRAYHS_PROFILE_TIME build's debug line (development aid).
    RAYHS_LIB=build_exp/binpt/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 python tools/kbench.py dragon 3 2>&1 | python tools/phase_times.py"""
import re, sys
l = [x for x in sys.stdin if 'rayhs adv' in x][-1]
m = re.search(r"adv\[0..3\] (\d+) (\d+) (\d+) (\d+) .*counters\[6..15\]: \d+ \d+ \| \d+ \d+ \| (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", l)
v = list(map(int, m.groups()))
names = ["refill + loop top", "hit / shadow / exit sections", "light section", "specular section", "return / new sample", "begin (planes, spheres)",
         "  traverse: mesh start", "  traverse: node loops", "  traverse: leaves + pop", "traverse total"]
tot = sum(v[:6]) + v[9]
for n, x in zip(names, v):
    print(f"{n:30s} {x / 1e9:7.2f} G cycles  {100 * x / tot:5.1f} %")
print(f"{'sum over waves':30s} {tot / 1e9:7.2f} G cycles")
