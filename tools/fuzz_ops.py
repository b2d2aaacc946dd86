"""Random-shape sweep of the single-op C-ABI entries (`mica_op_*`, include/mica_hip.h) against torch on the CPU:

    python tools/fuzz_ops.py [seconds=240] [seed=0] [big]

Every case draws an op, channel counts, a box (edges 1..96, biased to the awkward ones: 1, odd, one over / under a tile edge, the
production width 64) and a batch, runs the HIP kernel through the Engine and compares with torch's float32 result of the same op
(`F.conv3d`, `F.instance_norm`, grouped conv) in the metric of tests/test_gpu_ops.py: max |got - ref| / max(|ref|, rms(ref)).
The persistent convs walk (tile, channel block) items over one workgroup per CU with hand-counted wait states per chunk: item counts
below / equal to / not divisible by the workgroup count, odd chunk counts, padded last chunks and ragged tiles on every face are
what the sweep is after; `big` draws edges >= 15 and boxes four times the volume (item counts well above the 256 workgroups).  One line per case goes to stdout (a hang shows as the last line); the last line is the tally.
TEST INFRASTRUCTURE (the comparator is torch on the CPU, nothing under oracle/ is needed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from mica_amd.engine import Engine
from mica_amd.weights import synth_state_dict

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
rng = np.random.default_rng(seed)
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = torch.maximum(ref.abs(), ref.pow(2).mean().sqrt().expand_as(ref))
    return float(((got - ref).abs() / scale).max())


def rand(shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(rng.random(shape, dtype=np.float32) * (hi - lo) + lo)


EDGES = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 20, 31, 32, 33, 48, 63, 64, 65, 66, 96]
if BIG:
    EDGES = [e for e in EDGES if e >= 15]


def box(max_vox):
    if BIG:
        max_vox *= 4
    while True:
        d = [int(rng.choice(EDGES)) for _ in range(3)]
        if rng.random() < 0.3:
            d[2] = 64                                    # the production width
        if d[0] * d[1] * d[2] <= max_vox:
            return tuple(d)


def case_conv3(eng, w_all):
    variant = int(rng.integers(0, 2))
    cin = int(rng.choice([1, 3, 8, 16, 17, 24, 32, 40, 64, 72, 96, 130, 192, 196, 200, 256, 384]))
    cout = int(rng.choice([64, 64, 128, 128, 256, 512] if variant == 1 else [32, 64, 96, 128, 160, 192, 256]))
    if BIG and cin * cout > 128 * 256:                   # the float32 comparator runs on the host's cores
        cin = int(rng.choice([3, 16, 24, 40, 64, 72]))
    dims, batch = box(40000 if cin * cout <= 128 * 128 else 12000), int(rng.integers(1, 4))
    x, w, b = rand((batch, cin, *dims)), rand((cout, cin, 3, 3, 3)) * (3.0 / (cin * 27)) ** 0.5, rand((cout,)) * 0.1
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=variant)
    return f"conv3 v{variant} {cin}->{cout} {dims} b{batch}", rel_err(got, F.conv3d(x, w, b, padding=1)), 1e-4


def case_conv1(eng, w_all):
    cin, cout = int(rng.choice([16, 20, 48, 64, 128, 192, 256, 512])), int(rng.choice([64, 128, 256]))
    dims, batch = box(30000), int(rng.integers(1, 4))
    x, w, b = rand((batch, cin, *dims)), rand((cout, cin, 1, 1, 1)) * (3.0 / cin) ** 0.5, rand((cout,)) * 0.1
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 1)
    return f"conv1 {cin}->{cout} {dims} b{batch}", rel_err(got, F.conv3d(x, w, b)), 1e-4


def case_fused(eng, w_all):
    variant = int(rng.integers(0, 2))
    cin, cmid = int(rng.choice([16, 32, 64, 128, 256, 512])), int(rng.choice([64, 128, 256]))
    cout = int(rng.choice([64, 128]) if variant == 1 else rng.choice([32, 64, 128]))
    dims, batch = box(8000), int(rng.integers(1, 4))
    if dims[0] * dims[1] * dims[2] < 8:
        dims = (2, 2, 4)                                 # InstanceNorm over a handful of voxels is all rounding noise
    x = rand((batch, cin, *dims)) * 2.0 + 0.5
    w1, b1 = rand((cmid, cin, 1, 1, 1)) * (3.0 / cin) ** 0.5, rand((cmid,)) * 0.1
    w3, b3 = rand((cout, cmid, 3, 3, 3)) * (3.0 / (cmid * 27)) ** 0.5, rand((cout,)) * 0.1
    ref = F.conv3d(F.conv3d(F.relu(F.instance_norm(x, eps=1e-5)), w1, b1), w3, b3, padding=1)
    got = eng.op_norm_conv1_conv3(x.cuda(), w1.numpy().reshape(cmid, cin), b1.numpy(), w3.numpy(), b3.numpy(), variant=variant)
    return f"norm+conv1+conv3 v{variant} {cin}->{cmid}->{cout} {dims} b{batch}", rel_err(got, ref), 1e-4


def case_depthwise(eng, w_all):
    c = int(rng.choice([16, 32, 64, 128, 256]))
    dims, batch = box(60000), int(rng.integers(1, 5))
    x, w, b = rand((batch, c, *dims)), rand((c, 1, 3, 3, 3)) * 0.3, rand((c,)) * 0.1
    got = eng.op_depthwise3(x.cuda(), w.numpy(), b.numpy())
    return f"depthwise {c} {dims} b{batch}", rel_err(got, F.conv3d(x, w, b, padding=1, groups=c)), 1e-5


def case_stem(eng, w_all):
    dims = box(60000)
    x = rand((2, 1, *dims), 0.0, 1.0)
    ref = torch.cat([F.conv3d(x, torch.from_numpy(w_all[f"input_processing.exp_convs.{i}.weight"]),
                              torch.from_numpy(w_all[f"input_processing.exp_convs.{i}.bias"]), padding=k // 2) for i, k in enumerate((3, 5, 7, 9))], 1)
    return f"stem {dims}", rel_err(eng.op_stem(x.cuda()), ref), 1e-5


def case_instnorm(eng, w_all):
    c, dims = int(rng.choice([8, 32, 64, 512])), box(30000)
    if dims[0] * dims[1] * dims[2] < 8:
        dims = (2, 2, 4)
    x = rand((2, c, *dims)) * 3.0 + 5.0
    ref = F.relu(F.instance_norm(x, eps=1e-5))
    got = eng.op_instnorm_relu(x.cuda())
    return f"instnorm+relu {c} {dims}", float((got.cpu() - ref).abs().max()), 1e-4


CASES = [(case_conv3, 0.45), (case_fused, 0.15), (case_conv1, 0.12), (case_depthwise, 0.12), (case_stem, 0.08), (case_instnorm, 0.08)]
w_all = synth_state_dict(2022)
eng = Engine(0, max_batch=4, tile_size=16)
eng.load_state_dict(w_all)
t0, n, bad = time.time(), 0, 0
worst = {}
probs = np.array([p for _, p in CASES]) / sum(p for _, p in CASES)
while time.time() - t0 < seconds:
    fn = CASES[int(rng.choice(len(CASES), p=probs))][0]
    name, err, tol = fn(eng, w_all)
    n += 1
    ok = err < tol and np.isfinite(err)
    bad += not ok
    key = name.split()[0] + (" " + name.split()[1] if name.split()[1].startswith("v") else "")
    worst[key] = max(worst.get(key, 0.0), err / tol)
    print(f"{n:4d} {'ok ' if ok else 'BAD'} {err:.2e} (tol {tol:.0e})  {name}", flush=True)
eng.close()
print(f"{n} random cases in {time.time() - t0:.0f} s (seed {seed}{', big boxes' if BIG else ''}), {bad} beyond tolerance; worst error / tolerance per op: "
      + ", ".join(f"{k} {v:.3f}" for k, v in sorted(worst.items())))
sys.exit(1 if bad else 0)
