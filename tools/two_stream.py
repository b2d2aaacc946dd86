"""Experiment: two engines on two HIP streams driven by two host threads (overlap HBM-bound passes of one
batch with the MFMA-bound convs of the other)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.weights import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 6
dev = torch.device("cuda", 0)
n = 256
vol = torch.from_numpy(np.random.default_rng(1001).random((n, n, n), dtype=np.float32)).to(dev)
g = torch.Generator(device=dev).manual_seed(2001)
af = (torch.rand((24, n, n, n), generator=g, device=dev) < 1e-3).float()
w = synth_state_dict(2022)
engs = [Engine(0, max_batch=B, tile_size=64) for _ in range(NS)]
for e in engs:
    e.load_state_dict(w)
vps = [VolumePredictor(e, 32, 16, B) for e in engs]
streams = [torch.cuda.Stream() for _ in range(NS)]
outs = [torch.zeros((23, n, n, n), device=dev) for _ in range(NS)]

def work(i, nsteps, base):
    with torch.cuda.stream(streams[i]):
        for k in range(nsteps):
            first = ((base + k * NS + i) * B) % 500
            rec = vps[i].run_batch(vol, af, first, B)
            engs[i].stitch_tiles(rec, outs[i], 32, 16, first)

def run(nsteps, base):
    th = [threading.Thread(target=work, args=(i, nsteps, base)) for i in range(NS)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()

run(2, 0)
t0 = time.perf_counter()
run(steps, 10)
dt = time.perf_counter() - t0
print(f"streams={NS} batch={B}: {steps * NS * B / dt:.2f} tiles/s")
