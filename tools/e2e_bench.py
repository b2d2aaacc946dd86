"""End-to-end timing with host buffers (PCIe inclusive): maps on the host -> four volumes on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.weights import synth_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e = Engine(0, max_batch=8, tile_size=64); e.load_state_dict(synth_state_dict(2022))
vp = VolumePredictor(e, 48, 8, 8)
maps = [np.random.default_rng(1003 + i).random((n, n, n), dtype=np.float32) for i in range(nm)]
afs = [(np.random.default_rng(2003 + i).random((24, n, n, n), dtype=np.float32) < 1e-3).astype(np.float32) for i in range(nm)]
T = int(e.lib.mica_tile_count(n, n, n, 48))
vp.predict_maps_streamed(maps[:1], afs[:1])
torch.cuda.synchronize(); t0 = time.perf_counter()
res = vp.predict_maps_streamed(maps, afs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{nm} maps of {n}^3 (+24-ch encodings), {T} tiles each, host->host incl. H2D/D2H: {dt:.2f} s = {nm*T/dt:.1f} sub-grids/s")
dev = [torch.from_numpy(m).cuda() for m in maps]; devaf = [torch.from_numpy(a).cuda() for a in afs]
torch.cuda.synchronize(); t0 = time.perf_counter()
for m, a in zip(dev, devaf): vp.predict_volume(m, a)
torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print(f"same with inputs/outputs resident in HBM: {dt2:.2f} s = {nm*T/dt2:.1f} sub-grids/s")
