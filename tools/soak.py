"""Soak run of the forward graph: the same two batches of 64^3 tiles alternate for N iterations and every result must equal the
first one of its batch bit for bit (the hand-synchronised conv kernels - counted vmcnt waits, LDS-DMA double buffering - would show a
rare race as a changed bit).  usage: python tools/soak.py [iterations=400] [batch=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mica_amd.engine import Engine, AF_PER_TILE
from mica_amd.synth import synth_af, synth_density
from mica_amd.weights import synth_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = 64
e = Engine(0, max_batch=B, tile_size=S)
e.load_state_dict(synth_state_dict())
batches = []
for seed in (11, 12):
    x = torch.from_numpy(synth_density((B, S, S, S), seed)).cuda()
    af = torch.stack([torch.from_numpy(synth_af((S, S, S), seed + i, 0.01)) if i % 3 else torch.zeros(24, S, S, S) for i in range(B)]).cuda()
    batches.append((x, af))
rec = torch.empty((B, 23, S, S, S), dtype=torch.float32, device="cuda")
first = [None, None]
bad = 0
t0 = time.time()
for it in range(n):
    k = it & 1
    e.forward_records(batches[k][0], batches[k][1], rec, af_mode=AF_PER_TILE)
    if first[k] is None:
        first[k] = rec.clone()
    elif not torch.equal(rec, first[k]):
        bad += 1
        d = (rec != first[k])
        print(f"iteration {it}: {int(d.sum())} values differ, tiles {sorted(set(d.nonzero()[:, 0].tolist()))}", flush=True)
    if it % 100 == 99:
        print(f"{it + 1} iterations, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {n} iterations of {B} tiles, {bad} mismatching results")
sys.exit(1 if bad else 0)
