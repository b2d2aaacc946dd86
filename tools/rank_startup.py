"""What a worker rank costs to start, against what it computes (profiles/rNN_rank_startup.txt).

mica_amd/multi.py::RankPool on THIS box: rank 0 = this process, N-1 workers as fresh child processes - on a one-GPU box all on cuda:0 over
gloo (a functional rehearsal: host-staged broadcast and exchange, ranks sharing one card; the start-up marks are what is measured, the
map rate is NOT a scaling number).  Prints, per worker: python + torch import, engine (workspace) construction, the wait at the
rendezvous, the first map; then a second map through the same workers (no start-up left), next to one rank alone.
usage: python tools/rank_startup.py [n=256] [ranks=2] [backend=gloo]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mica_amd import multi
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.weights import load_checkpoint_state_dict, synth_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
tmp = tempfile.mkdtemp(prefix="mica_ranks_")
ck = os.path.join(tmp, "ckpt.pth")
torch.save({"model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in synth_state_dict(2022).items()}}, ck)

t0 = time.time()
pool = multi.RankPool(ranks, tile=64, batch=8, backend=backend, devices=[0] * ranks if backend == "gloo" else None).spawn()
# rank 0's own start, beside the workers': engine, weights, the map (stands in for getData)
eng = Engine(0, max_batch=8, tile_size=64)
t_eng = time.time()
eng.load_state_dict(load_checkpoint_state_dict(ck))
t_w = time.time()
g = torch.Generator(device="cuda").manual_seed(7)
vol = torch.rand((n, n, n), generator=g, device="cuda")
af = torch.empty((24, n, n, n), dtype=torch.uint8, device="cuda")
for c in range(24):
    af[c] = (torch.rand((n, n, n), generator=g, device="cuda") < 1e-3).to(torch.uint8)
torch.cuda.synchronize()
t_map = time.time()
print(f"rank 0 (this process, torch already imported): engine {t_eng - t0:.2f} s, checkpoint load + weight pack {t_w - t_eng:.2f} s, synthetic {n}^3 map + encodings {t_map - t_w:.2f} s")
runner = multi.EngineRunner(None, 64, 8, engine=eng, loaded_model=ck)
T = int(eng.lib.mica_tile_count(n, n, n, 48))
times = []
for m in range(2):
    t1 = time.time()
    vols = pool.predict(runner, ck, vol, af, 48, 8, to_host=True)
    times.append(time.time() - t1)
    print(f"map {m + 1}: {ranks} ranks, {T} tiles, {times[-1]:.2f} s wall in pool.predict ({T / times[-1]:.1f} sub-grids/s; rendezvous wait of rank 0 included in map 1); "
          + "; ".join(f"rank {s['rank']}: broadcast {s.get('broadcast_s', 0):.2f} s, predict {s.get('predict_s', 0):.2f} s" + (f", weights {s['load_model_s']:.2f} s" if 'load_model_s' in s else "") for s in pool.last_status))
print(pool.startup_report())
print(f"first tile of the workers could start {max(s['joined'] for s in pool.startup) - pool.t_spawn:.2f} s after spawn (start-up) + weights; "
      f"a rank's share of this map computes for ~{T / ranks / 84.0:.1f} s at 84 sub-grids/s per GPU, of a 512^3 map on 8 GPUs ~{1331 / 8 / 84.0:.1f} s")
pool.close()
t1 = time.time()
one = VolumePredictor(eng, 48, 8, 8).predict_volume(vol, af, to_host=True)
t_one = time.time() - t1
print(f"one rank alone, same map: {t_one:.2f} s ({T / t_one:.1f} sub-grids/s)")
for k in one:
    assert np.array_equal(one[k], vols[k]), k
print("volumes of the pool == one rank alone, bit for bit")
eng.close()
