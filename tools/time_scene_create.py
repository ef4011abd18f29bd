import time, sys
sys.path.insert(0, '.')
import rayhs_amd
from rayhs_amd.api import build_rendering
for i in range(3):
    t = time.time(); r = build_rendering("scenes/dragon_1080p.json"); t1 = time.time() - t
    t = time.time(); img, st = r.ray_trace(spp=16); t2 = time.time() - t
    t = time.time(); img, st = r.ray_trace(spp=16); t3 = time.time() - t
    r.close()
    print(f"build_rendering {t1*1e3:.1f} ms, first render {t2*1e3:.1f} ms, second {t3*1e3:.1f} ms, kernel {st['kernel_ms']:.2f}")
