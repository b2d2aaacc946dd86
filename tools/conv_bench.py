"""Micro-benchmark of the dense conv kernel through mica_op_conv3d (development aid; not part of the product).
Usage: python tools/conv_bench.py [cin cout k [S [reps]]]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd.engine import Engine

shapes = [(512, 256, 3), (256, 512, 3), (256, 128, 3), (128, 256, 3), (64, 32, 3), (64, 64, 3), (192, 64, 3), (512, 256, 1), (128, 64, 1)]
if len(sys.argv) >= 4:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
S = int(sys.argv[4]) if len(sys.argv) > 4 else 64
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
zero = os.environ.get('ZERO', '0') == '1'
variant = int(os.environ.get('VARIANT', '0'))      # 1: the F(4,3) kernel (cout % 128 == 0)
e = Engine(0, max_batch=1, tile_size=16)
for cin, cout, k in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((1, cin, S, S, S), generator=g, device="cuda") * 2 - 0.5
    w = ((np.random.default_rng(2).random((cout, cin, k, k, k), dtype=np.float32) * 2 - 1) * (3.0 / (cin * k ** 3)) ** 0.5).astype(np.float32)
    b = np.zeros(cout, np.float32)
    if zero:
        x = torch.zeros_like(x); w = np.zeros_like(w)
    e.set_profiling(False)
    for _ in range(reps):
        y = e.op_conv3d(x, w, b, k, variant=variant if k == 3 else 0)
    torch.cuda.synchronize()
    print(f"conv {cin}->{cout} k={k} S={S}: done, out mean {float(y.mean()):.5f}", flush=True)
