import time, sys, torch
sys.path.insert(0, '.')
from rayhs_amd import build_rendering, lib as L
r = build_rendering("scenes/cornell_96.json")
out = torch.empty((96, 96, 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
for n in (200, 2000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        r.ray_trace_device(out.data_ptr(), st.cuda_stream, out_format=L.OUT_F32)
    t1 = time.perf_counter() - t
    torch.cuda.synchronize(); t2 = time.perf_counter() - t
    print(f"n={n}: host enqueue {t1/n*1e6:.1f} us/call, with GPU {t2/n*1e6:.1f} us/call")
