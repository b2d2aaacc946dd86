"""Time the map preprocessing stages on the GPU against scipy/numpy on the host (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
e = Engine(0, max_batch=1, tile_size=64)
x = (np.random.default_rng(1).random((n, n, n), dtype=np.float32) - 0.3) * 2
t = torch.from_numpy(x).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = e.zoom_cubic(t, (f, f, f)); torch.cuda.synchronize(); t1 = time.perf_counter()
    zz = z.clone(); med, pct = e.normalise_map_(zz); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"GPU: zoom {n}^3 x{f} -> {tuple(z.shape)} {1e3*(t1-t0):.1f} ms ; normalise {1e3*(t2-t1):.1f} ms")
if len(sys.argv) > 3:
    from scipy.ndimage import zoom
    t0 = time.perf_counter(); r = zoom(x, (f, f, f), order=3); t1 = time.perf_counter()
    m = np.median(r); mm = (r > m) * (r - m); p = np.percentile(mm[mm > 0], 99.9); t2 = time.perf_counter()
    print(f"CPU: scipy zoom {t1-t0:.2f} s ; median+percentile {t2-t1:.2f} s ; zoom equal {np.array_equal(r, z.cpu().numpy())}")
