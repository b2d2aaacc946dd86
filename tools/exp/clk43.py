"""Cycle stamps of one chunk of conv_wino43_kernel per wave (dev build -DMICA43_CLOCKS: make -C mica_amd/csrc exp_abl43 ABL43_EXTRA=CLOCKS).
usage: MICA_HIP_LIB=tools/exp/libmica43_CLOCKS.so python tools/exp/clk43.py [cin cout]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mica_amd.engine import Engine

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
e = Engine(0, max_batch=1, tile_size=16)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((1, cin, 64, 64, 64), generator=g, device="cuda") * 2 - 0.5
w = ((np.random.default_rng(2).random((cout, cin, 3, 3, 3), dtype=np.float32) * 2 - 1) * (3.0 / (cin * 27)) ** 0.5).astype(np.float32)
b = np.zeros(cout, np.float32)
for _ in range(3):
    e.op_conv3d(x, w, b, 3, variant=1)
torch.cuda.synchronize()
buf = (C.c_uint * (12 * 48))()
e.lib.mica_debug_conv43.restype = C.c_int
e.lib.mica_debug_conv43.argtypes = [C.POINTER(C.c_uint), C.c_int]
print("read rc", e.lib.mica_debug_conv43(buf, 12 * 48))
t = np.array(buf, dtype=np.int64).reshape(12, 48)
M = 1 << 32
d = lambda a, b_: (a - b_) % M
t0 = t[:, 0].min()
print("wave  start | first wait | steps 0-2  3-5  6-8  9-11  12-13 | loop end  vmcnt  barrier  total")
for wv in range(12):
    r = t[wv]
    marks = [r[1], r[2 + 4], r[2 + 10], r[2 + 16], r[2 + 22], r[2 + 26]]
    seg = [d(marks[i + 1], marks[i]) for i in range(5)]
    print(f"{wv:2d}  {d(r[0], t0):6d} | {d(r[1], r[0]):6d} | " + " ".join(f"{v:6d}" for v in seg) + f" | {d(r[30], r[0]):6d} {d(r[31], r[30]):5d} {d(r[32], r[31]):6d} {d(r[32], r[0]):6d}")
print("ideal MFMA cycles per SIMD and chunk: 3 waves x 224 MFMAs x 16 =", 3 * 224 * 16, "; per wave alone 3584; a 3-step segment alone 768")
