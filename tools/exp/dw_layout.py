"""Round 5 experiment: does the depthwise kernel's rate depend on how contiguous a 32-channel slab is in HBM?  C = 32 (a slab IS the whole
voxel row: the kernel streams contiguous memory) against C = 64 / 128 / 256 (128-byte pieces at 256 B ... 1 KB pitch), same kernel variant
<4, 8>, enough workgroups for two rounds on every CU.  Run under rocprofv3 --kernel-trace --stats (tools/exp/dw_layout.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mica_amd.engine import Engine

e = Engine(0, max_batch=1, tile_size=16)
for C, S, B in ((32, 128, 8), (64, 128, 4), (128, 64, 16), (256, 64, 8)):
    x = torch.rand((B, C, S, S, S), device="cuda")
    w = np.random.default_rng(1).random((C, 27), dtype=np.float32)
    b = np.zeros(C, np.float32)
    for _ in range(3):
        y = e.op_depthwise3(x, w.reshape(C, 1, 3, 3, 3), b)
    torch.cuda.synchronize()
    print(f"C={C} S={S} B={B}: algorithmic bytes per launch {8 * C * S ** 3 * B / 1e9:.3f} GB", flush=True)
    del x, y
