"""Per-kernel parity through the C ABI against the CPU oracle's arithmetic (torch CPU fp32 = the
reference's L1 runtime).  Tolerance: 1e-4 relative to max(|ref|, rms(ref)) per element (BASELINE.json:
'within 1e-4 relative fp32 per voxel'); measured errors are ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def rel_err(got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = torch.maximum(ref.abs(), ref.pow(2).mean().sqrt().expand_as(ref))
    return float(((got - ref).abs() / scale).max())


@pytest.fixture(scope="module")
def eng(weights):
    from mica_amd.engine import Engine
    e = Engine(0, max_batch=2, tile_size=16)
    e.load_state_dict(weights)
    yield e
    e.close()


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("cin,cout,k,dims", [
    (16, 32, 3, (8, 8, 8)), (24, 64, 3, (8, 8, 16)), (64, 128, 3, (6, 10, 20)), (196, 64, 3, (8, 8, 8)),
    (96, 32, 3, (16, 16, 16)), (128, 64, 1, (8, 8, 8)), (192, 64, 1, (4, 12, 9)), (512, 256, 1, (8, 8, 8)), (48, 128, 1, (7, 8, 8)), (20, 64, 1, (3, 5, 7)),
    (32, 256, 3, (8, 8, 8)), (40, 128, 3, (5, 7, 9)), (144, 256, 3, (12, 6, 18)), (48, 192, 3, (6, 9, 14)), (40, 96, 3, (7, 9, 12)),     # 128 / 64 / 32 blocks of the persistent 16x16x32 conv
])
def test_conv3d(eng, cin, cout, k, dims):
    x = _rand((2, cin, *dims), 1)
    w = _rand((cout, cin, k, k, k), 2) * (3.0 / (cin * k ** 3)) ** 0.5
    b = _rand((cout,), 3) * 0.1
    ref = F.conv3d(x, w, b, padding=k // 2)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), k)
    assert rel_err(got, ref) < RTOL


@pytest.mark.parametrize("cin,cmid,cout,dims,batch", [
    (128, 64, 64, (8, 8, 8), 2),            # W = 8: 32 x rows per workgroup tile, operand emitted by the 1x1 kernel
    (256, 128, 32, (4, 8, 16), 2),          # two 64-channel epilogue passes
    (512, 256, 64, (3, 5, 64), 1),          # W = 64 (the production width), ragged last tile (960 voxels), four passes
    (16, 64, 32, (5, 6, 32), 2),            # a single chunk: the second chunk of the pair is missing; W = 32, ragged last tile
    (64, 64, 64, (6, 5, 9), 2),             # W = 9 does not divide the tile: raw output + operand pass
    (32, 128, 64, (2, 3, 66), 1),           # W > 64: raw output + operand pass
    (64, 64, 32, (2, 2, 4), 3),             # a volume smaller than one workgroup tile (16 voxels), W = 4
])
def test_fused_norm_conv1x1_into_winograd_conv(eng, cin, cmid, cout, dims, batch):
    """kernels_conv1x1.hip: InstanceNorm + ReLU applied on load of a RAW source, split-f16 MFMA GEMM, and the Winograd operand
    of the following 3x3x3 conv written by the epilogue - against torch's conv3d(conv3d(relu(instance_norm(x))))."""
    x = _rand((batch, cin, *dims), 21) * 2.0 + 0.5
    w1 = _rand((cmid, cin, 1, 1, 1), 22) * (3.0 / cin) ** 0.5
    b1 = _rand((cmid,), 23) * 0.1
    w3 = _rand((cout, cmid, 3, 3, 3), 24) * (3.0 / (cmid * 27)) ** 0.5
    b3 = _rand((cout,), 25) * 0.1
    ref = F.conv3d(F.conv3d(F.relu(F.instance_norm(x, eps=1e-5)), w1, b1), w3, b3, padding=1)
    got = eng.op_norm_conv1_conv3(x.cuda(), w1.numpy().reshape(cmid, cin), b1.numpy(), w3.numpy(), b3.numpy())
    assert rel_err(got, ref) < RTOL


@pytest.mark.parametrize("cin,cout,dims,batch", [
    (16, 128, (8, 8, 8), 1),                # one chunk, one tile per z/y block, W = 8 (two quads)
    (32, 128, (4, 4, 16), 2),               # exactly one tile per batch entry
    (256, 128, (8, 8, 16), 2),              # encoder.2 conv1's channels at S = 16-like sizes
    (384, 128, (6, 10, 20), 1),             # 24 chunks, ragged tiles in y and z, W = 20 (5 quads: two x tiles, the second ragged)
    (64, 256, (5, 7, 9), 2),                # W = 9: a ragged quad (zero padded operand), two channel blocks
    (40, 512, (12, 6, 18), 1),              # padded last chunk (Cin = 40), four channel blocks
    (72, 128, (16, 32, 16), 1),             # the blocked tile walk (nty = 8, ntz = 4)
    (130, 256, (3, 64, 64), 1),             # production width, items > workgroups
    (3, 128, (1, 1, 1), 3),                 # a single voxel
    (48, 128, (2, 3, 66), 1),               # W > 64
    # the 64-channel variant (round 5): the two wave groups split the taps, partial sums added in the epilogue
    (64, 64, (8, 8, 8), 2),                 # FPN smooth conv's channels; four chunks
    (16, 64, (4, 4, 16), 1),                # one chunk, one tile
    (192, 64, (6, 10, 20), 1),              # backbone head conv1: 12 chunks, ragged tiles
    (196, 64, (8, 8, 16), 2),               # ca head conv1: 13 chunks (odd), padded last chunk
    (200, 64, (5, 7, 9), 1),                # aa head conv1; ragged quad
    (64, 64, (3, 64, 64), 1),               # production width, items > workgroups
    (24, 64, (16, 32, 16), 2),              # the blocked tile walk
    (5, 64, (1, 1, 1), 2),                  # a single voxel
])
def test_conv3d_winograd_f43_kernel(eng, cin, cout, dims, batch):
    """conv_wino43_kernel<128 | 64> (kernels_conv43.hip: Winograd F(4,3) along x, the kernel of encoder.2's convs and - its tap-split
    64-channel variant - of the FPN smooth convs and the heads' conv1) as a single op against
    torch's conv3d: operand producer, weight packer, 6-position slab, output transform, ragged tiles, persistent item walk."""
    x = _rand((batch, cin, *dims), 31)
    w = _rand((cout, cin, 3, 3, 3), 32) * (3.0 / (cin * 27)) ** 0.5
    b = _rand((cout,), 33) * 0.1
    ref = F.conv3d(x, w, b, padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1)
    assert rel_err(got, ref) < RTOL


def test_conv3d_winograd_f43_tap_split_variant_asymmetric_deltas_and_determinism(eng):
    """The 64-channel variant: delta inputs with asymmetric weights (a tap assigned to the wrong wave group, a swapped pair-step or a
    wrong tap-8 half shows at once), and bitwise determinism of the partial-sum meeting (one LDS add per address)."""
    x = torch.zeros((1, 32, 8, 8, 16))
    x[0, 3, 2, 5, 7] = 1.0
    x[0, 27, 6, 1, 12] = -2.0
    x[0, 5, 0, 0, 0] = 0.5
    x[0, 17, 7, 7, 15] = 3.0
    w = torch.arange(64 * 32 * 27, dtype=torch.float32).reshape(64, 32, 3, 3, 3) / 4000.0
    b = torch.arange(64, dtype=torch.float32)
    ref = F.conv3d(x, w, b, padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1)
    assert rel_err(got, ref) < 2e-5
    x = _rand((2, 196, 8, 16, 32), 5, 0.0, 4.0)
    w = _rand((64, 196, 3, 3, 3), 6) * 0.03
    b = _rand((64,), 7)
    a1 = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1)
    a2 = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1)
    assert torch.equal(a1, a2)
    ref64 = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    e43, e23 = rel_err(a1, ref64), rel_err(eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=0), ref64)
    print(f"196->64 vs float64: F(4,3) tap-split {e43:.2e}, F(2,3) {e23:.2e}")
    assert e43 < 2e-5


def test_conv3d_winograd_f43_asymmetric_identity_and_precision(eng):
    """Delta inputs with asymmetric weights (transposed operand / C-D maps, position or tap mix-ups show at once), then the
    accuracy of the F(4,3) arithmetic on a deep layer against float64: about 4x the F(2,3) kernel's error, still fp32-grade."""
    x = torch.zeros((1, 16, 8, 8, 16))
    x[0, 3, 2, 5, 7] = 1.0
    x[0, 11, 6, 1, 12] = -2.0
    x[0, 5, 0, 0, 0] = 0.5
    x[0, 7, 7, 7, 15] = 3.0
    w = torch.arange(128 * 16 * 27, dtype=torch.float32).reshape(128, 16, 3, 3, 3) / 4000.0
    b = torch.arange(128, dtype=torch.float32)
    ref = F.conv3d(x, w, b, padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1)
    assert rel_err(got, ref) < 2e-5
    x = _rand((1, 512, 8, 8, 16), 5, 0.0, 4.0)
    w = _rand((256, 512, 3, 3, 3), 6) * 0.02
    b = torch.zeros(256)
    ref64 = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    e43 = float((eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=1).cpu().double() - ref64).abs().max() / ref64.abs().max())
    e23 = float((eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3, variant=0).cpu().double() - ref64).abs().max() / ref64.abs().max())
    print(f"512->256 vs float64: F(4,3) {e43:.2e}, F(2,3) {e23:.2e}")
    assert e43 < 1.5e-5 and e23 < 2e-6


@pytest.mark.parametrize("cin,cmid,cout,dims,batch", [
    (512, 256, 512, (4, 4, 16), 1),         # encoder.2's dual_attn.fusion -> transition: four epilogue passes, F(4,3) operand emitted
    (128, 64, 128, (8, 8, 8), 2),           # W = 8
    (256, 128, 128, (3, 5, 64), 1),         # W = 64 (the production width), ragged last tile
    (64, 64, 128, (6, 5, 9), 2),            # W = 9 does not divide the tile: raw output + the F(4,3) operand pass
    (64, 64, 128, (2, 2, 4), 3),            # W = 4: one quad per row
])
def test_fused_norm_conv1x1_into_winograd_f43_conv(eng, cin, cmid, cout, dims, batch):
    """The 1x1 kernel's F(4,3) epilogue (kernels_conv1x1.hip, WINO = 2) feeding conv_wino43_kernel."""
    x = _rand((batch, cin, *dims), 21) * 2.0 + 0.5
    w1 = _rand((cmid, cin, 1, 1, 1), 22) * (3.0 / cin) ** 0.5
    b1 = _rand((cmid,), 23) * 0.1
    w3 = _rand((cout, cmid, 3, 3, 3), 24) * (3.0 / (cmid * 27)) ** 0.5
    b3 = _rand((cout,), 25) * 0.1
    ref = F.conv3d(F.conv3d(F.relu(F.instance_norm(x, eps=1e-5)), w1, b1), w3, b3, padding=1)
    got = eng.op_norm_conv1_conv3(x.cuda(), w1.numpy().reshape(cmid, cin), b1.numpy(), w3.numpy(), b3.numpy(), variant=1)
    assert rel_err(got, ref) < RTOL


@pytest.mark.parametrize("seed", range(8))
def test_conv3d_persistent_schedule_odd_shapes(eng, seed):
    """The persistent conv distributes (batch, tile, channel block) items over one workgroup per CU: item counts below,
    equal to and not divisible by the number of workgroups, single-chunk layers (Cin <= 16), ragged volumes whose
    tiles hang over every face, the blocked and the linear tile walk (nty % 8, ntz % 4)."""
    rng = np.random.default_rng(100 + seed)
    cin = int(rng.choice([3, 16, 24, 40, 72, 130]))
    cout = int(rng.choice([32, 64, 96, 128, 160, 192, 256]))
    dims = [(5, 6, 7), (4, 4, 34), (17, 3, 5), (16, 32, 16), (9, 33, 18), (3, 64, 64), (20, 8, 40), (1, 1, 1)][seed]
    batch = int(rng.integers(1, 4))
    x = _rand((batch, cin, *dims), 200 + seed)
    w = _rand((cout, cin, 3, 3, 3), 300 + seed) * (3.0 / (cin * 27)) ** 0.5
    b = _rand((cout,), 400 + seed) * 0.1
    ref = F.conv3d(x, w, b, padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3)
    assert rel_err(got, ref) < RTOL


def test_conv3d_asymmetric_identity(eng):
    """A = delta input, asymmetric weights: catches transposed MFMA operand / C-D maps."""
    x = torch.zeros((1, 16, 8, 8, 16))
    x[0, 3, 2, 5, 7] = 1.0
    x[0, 11, 6, 1, 12] = -2.0
    w = torch.arange(32 * 16 * 27, dtype=torch.float32).reshape(32, 16, 3, 3, 3) / 1000.0
    b = torch.arange(32, dtype=torch.float32)
    ref = F.conv3d(x, w, b, padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3)
    assert rel_err(got, ref) < 1e-5


def test_conv3d_precision_split_f16(eng):
    """The split-f16 (hi+lo) MFMA path must be ~fp32 accurate, not f16 accurate."""
    x = _rand((1, 256, 8, 8, 8), 5, 0.0, 4.0)
    w = _rand((64, 256, 3, 3, 3), 6) * 0.02
    b = torch.zeros(64)
    ref64 = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    got = eng.op_conv3d(x.cuda(), w.numpy(), b.numpy(), 3).cpu().double()
    ref32 = F.conv3d(x, w, b, padding=1).double()
    e_got = float((got - ref64).abs().max() / ref64.abs().max())
    e_f32 = float((ref32 - ref64).abs().max() / ref64.abs().max())
    assert e_got < 2e-6, (e_got, e_f32)


@pytest.mark.parametrize("c,dims", [(32, (8, 8, 8)), (64, (16, 16, 16)), (512, (4, 8, 8)), (8, (5, 7, 9))])
def test_instnorm_relu(eng, c, dims):
    x = _rand((2, c, *dims), 7) * 3.0 + 5.0     # large mean/std ratio stresses the variance
    ref = F.relu(F.instance_norm(x, eps=1e-5))
    got = eng.op_instnorm_relu(x.cuda())
    assert float((got.cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("c,dims,batch", [(64, (8, 8, 8), 2), (128, (6, 9, 11), 2), (256, (8, 8, 8), 2),
                                          (256, (6, 64, 64), 4)])      # the last one takes the tall-column (YO = 4) variant
def test_depthwise(eng, c, dims, batch):
    x = _rand((batch, c, *dims), 8)
    w = _rand((c, 1, 3, 3, 3), 9) * 0.3
    b = _rand((c,), 10) * 0.1
    ref = F.conv3d(x, w, b, padding=1, groups=c)
    got = eng.op_depthwise3(x.cuda(), w.numpy(), b.numpy())
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("c,dims,batch", [(64, (8, 8, 8), 2), (128, (6, 9, 11), 2), (256, (3, 64, 64), 4)])     # the last: 32-channel workgroups
def test_se_block_and_depthwise_local_branch(eng, c, dims, batch):
    """SEBlock + the local branch of DualAttention (model.py:254-258, 80-82, 99) the way the graph runs them: the SE pool is
    summed by the depthwise kernel while it loads, and the gate, which scales the depthwise conv's input in the reference,
    is folded into the InstanceNorm constants of its output.  Against torch, operation by operation."""
    x = _rand((batch, c, *dims), 41) * 2.0 + 0.3
    dw_w = _rand((c, 1, 3, 3, 3), 42) * 0.3
    dw_b = _rand((c,), 43) * 0.5                           # a sizeable bias: it must not be scaled by the gate
    fc0_w, fc0_b = _rand((c // 16, c), 44) * (3.0 / c) ** 0.5 * 4, _rand((c // 16,), 45) * 0.1
    fc3_w, fc3_b = _rand((c, c // 16), 46) * (3.0 * 16 / c) ** 0.5 * 4, _rand((c,), 47) * 0.5
    x3 = F.relu(F.instance_norm(x, eps=1e-5))
    g = torch.sigmoid(F.linear(F.relu(F.linear(x3.mean(dim=(2, 3, 4)), fc0_w, fc0_b)), fc3_w, fc3_b))
    assert float(g.max() - g.min()) > 0.2                   # the gates really differ between channels
    ref = F.relu(F.instance_norm(F.conv3d(x3 * g[:, :, None, None, None], dw_w, dw_b, padding=1, groups=c), eps=1e-5))
    got = eng.op_se_depthwise(x.cuda(), dw_w.numpy().reshape(c, 27), dw_b.numpy(), fc0_w.numpy(), fc0_b.numpy(), fc3_w.numpy(), fc3_b.numpy())
    assert rel_err(got, ref) < 5e-5                         # two InstanceNorms deep: float32 summation-order noise is ~1e-5


@pytest.mark.parametrize("dims", [(8, 8, 8), (16, 16, 16), (5, 9, 33), (4, 10, 64), (3, 5, 128), (17, 9, 64)])
def test_stem(eng, weights, dims):
    """Widths that are multiples of 64 run on the matrix cores (kernels_stem.hip: taps as the GEMM's K dimension, split-f16 products),
    the others on the f32 VALU kernel; ragged y / z blocks on both."""
    x = _rand((2, 1, *dims), 11, 0.0, 1.0)
    outs = []
    for i, k in enumerate((3, 5, 7, 9)):
        outs.append(F.conv3d(x, torch.from_numpy(weights[f"input_processing.exp_convs.{i}.weight"]),
                             torch.from_numpy(weights[f"input_processing.exp_convs.{i}.bias"]), padding=k // 2))
    ref = torch.cat(outs, 1)
    got = eng.op_stem(x.cuda())
    assert rel_err(got, ref) < 1e-5


def test_postprocess(eng):
    from oracle import model_oracle as mo
    bb = _rand((2, 4, 16, 16, 16), 12) * 5
    ca = _rand((2, 4, 16, 16, 16), 13) * 5
    aa = _rand((2, 21, 16, 16, 16), 14) * 5
    rb, rc, ra, rp = mo.postprocess(bb, ca, aa)
    gb, gc, ga, gp = eng.postprocess(bb.cuda(), ca.cuda(), aa.cuda())
    assert float((gb.cpu() - rb).abs().max()) < 1e-6
    assert float((gc.cpu() - rc).abs().max()) < 1e-6
    assert float((ga.cpu() - ra).abs().max()) < 1e-6
    assert torch.equal(gp.cpu().long(), rp)
    # the prediction is the first maximum of the softmax SCORES (predict.py:349): logits so close that their scores round to the
    # same float resolve to the lower class, like torch.max on the scores
    aa2 = torch.full((1, 21, 16, 16, 16), -30.0)
    aa2[:, 5] = 0.001
    aa2[:, 9] = torch.nextafter(torch.tensor(0.001), torch.tensor(1.0))      # a larger logit (one ulp), an equal score after rounding
    aa2[:, 1] = -1.0
    _, _, ra2, rp2 = mo.postprocess(bb[:1], ca[:1], aa2)
    _, _, ga2, gp2 = eng.postprocess(bb[:1].cuda(), ca[:1].cuda(), aa2.cuda())
    assert float(aa2[0, 9, 0, 0, 0]) > float(aa2[0, 5, 0, 0, 0]) and int(rp2.flatten()[0]) == 4      # class 5 -> index 4 of the 20 scores
    assert torch.equal(gp2.cpu().long(), rp2)


def test_forward_records_equal_forward_tiles(weights):
    """mica_forward_records writes the record layout [T,23,S^3] (bb, ca, aa_pred, aa_prob x20) that stitch_tiles and the
    multi-GPU exchange use: bit-identical to the four tensors of mica_forward_tiles."""
    from mica_amd.engine import AF_PER_TILE, Engine
    S = 16
    e = Engine(0, max_batch=2, tile_size=S)
    e.load_state_dict(weights)
    x = _rand((3, S, S, S), 31, 0.0, 1.0).cuda()
    af = (_rand((3, 24, S, S, S), 32, 0.0, 1.0) < 0.01).float().cuda()
    af[1] = 0
    bbp, cap, aap, pred = e.forward_tiles(x, af, af_mode=AF_PER_TILE)
    rec = torch.empty((3, 23, S, S, S), device="cuda")
    e.forward_records(x, af, rec, af_mode=AF_PER_TILE)
    assert torch.equal(rec[:, 0], bbp) and torch.equal(rec[:, 1], cap) and torch.equal(rec[:, 2], pred) and torch.equal(rec[:, 3:], aap)
    e.close()


# ---- ABI hygiene (include/mica_hip.h: "No entry point aborts the process"; box limits of the single-op entry points) -----------------
def _op_calls(e, x1):
    """One call per mica_op_* entry with the box (batch, d, h, w) left open: f(batch, d, h, w) -> return code.  The pointers are valid
    but tiny: an entry that did not check its box before touching them would fault, which is what the test is for."""
    import ctypes as C
    from mica_amd import _cabi
    lib, h, st = e.lib, e._h, e._stream()
    p = C.c_void_p(x1.data_ptr())
    w = np.zeros(1 << 16, np.float32).ctypes.data_as(_cabi._FP)
    return {
        "mica_op_conv3d": lambda b, d, hh, ww: lib.mica_op_conv3d(h, p, b, 16, d, hh, ww, w, w, 32, 3, p, st),
        "mica_op_conv3d_variant(1)": lambda b, d, hh, ww: lib.mica_op_conv3d_variant(h, p, b, 16, d, hh, ww, w, w, 128, 3, 1, p, st),
        "mica_op_norm_conv1_conv3": lambda b, d, hh, ww: lib.mica_op_norm_conv1_conv3(h, p, b, 16, d, hh, ww, w, w, 64, w, w, 32, p, st),
        "mica_op_norm_conv1_conv3_variant(1)": lambda b, d, hh, ww: lib.mica_op_norm_conv1_conv3_variant(h, p, b, 16, d, hh, ww, w, w, 64, w, w, 128, 1, p, st),
        "mica_op_instnorm_relu": lambda b, d, hh, ww: lib.mica_op_instnorm_relu(h, p, b, 16, d, hh, ww, p, st),
        "mica_op_depthwise3": lambda b, d, hh, ww: lib.mica_op_depthwise3(h, p, b, 16, d, hh, ww, w, w, p, st),
        "mica_op_se_depthwise": lambda b, d, hh, ww: lib.mica_op_se_depthwise(h, p, b, 32, d, hh, ww, w, w, w, w, w, w, p, st),
        "mica_op_stem": lambda b, d, hh, ww: lib.mica_op_stem(h, p, b, d, hh, ww, p, st),
    }


def test_single_op_entries_refuse_oversized_boxes(eng):
    """the exported single-op calls did not bound d, h, w - a large box reached an abort() in a launch helper or
    overflowed the kernels' 32-bit slab offsets.  Every entry now answers MICA_ERR_ARG (and says why) for an edge beyond 128, an edge
    below 1 or more than 64 tiles, before it allocates or launches anything; the context stays usable."""
    from mica_amd import _cabi
    x1 = torch.zeros(64, device="cuda")
    calls = _op_calls(eng, x1)
    for name, f in calls.items():
        for box in ((1, 129, 8, 8), (1, 8, 4096, 8), (1, 8, 8, 1 << 20), (65, 8, 8, 8), (1, 0, 8, 8), (1, 8, 8, -3)):
            rc = f(*box)
            assert rc == _cabi.MICA_ERR_ARG, (name, box, rc)
            msg = eng.lib.mica_last_error(eng._h).decode()
            assert "box limits" in msg or "bad argument" in msg, (name, msg)
    # still healthy
    x = _rand((1, 16, 8, 8, 8), 5)
    w = _rand((32, 16, 3, 3, 3), 6) * 0.1
    assert rel_err(eng.op_conv3d(x.cuda(), w.numpy(), np.zeros(32, np.float32), 3), F.conv3d(x, w, None, padding=1)) < RTOL


def test_a_pending_host_hip_error_is_neither_consumed_nor_launched_behind(eng):
    """MICA_ENTER used to call hipGetLastError() and so swallowed an error of the HOST program pending on the thread.  Now
    a call made behind such an error returns MICA_ERR_STATE without launching, and the error is still there for the host."""
    import ctypes as C
    from mica_amd import _cabi
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = C.CDLL(name)            # already mapped by torch: dlopen hands back the process's copy
            break
        except OSError:
            continue
    if hip is None:
        pytest.skip("libamdhip64 not loadable by name")
    hip.hipGetLastError.restype = C.c_int
    hip.hipPeekAtLastError.restype = C.c_int
    torch.cuda.synchronize()
    hip.hipGetLastError()
    bad = C.c_void_p()
    assert hip.hipMalloc(C.byref(bad), C.c_size_t(1 << 60)) != 0          # the host program's own failure: sets the sticky last error
    if hip.hipPeekAtLastError() == 0:
        pytest.skip("this runtime copy does not share torch's error state")
    x = _rand((1, 16, 8, 8, 8), 5).cuda()
    y = torch.empty_like(x)
    rc = eng.lib.mica_op_instnorm_relu(eng._h, C.c_void_p(x.data_ptr()), 1, 16, 8, 8, 8, C.c_void_p(y.data_ptr()), eng._stream())
    assert rc == _cabi.MICA_ERR_STATE and "pending" in eng.lib.mica_last_error(eng._h).decode()
    assert hip.hipGetLastError() != 0                                     # the host still finds its error ...
    assert hip.hipGetLastError() == 0                                     # ... once
    got = eng.op_instnorm_relu(x)                                         # and the library works again
    assert rel_err(got, F.relu(F.instance_norm(x.cpu()))) < RTOL


def test_engine_refuses_a_bad_conv_variant_before_it_allocates():
    from mica_amd.engine import Engine, MicaHipError
    free0 = torch.cuda.mem_get_info(0)[0]
    with pytest.raises(MicaHipError, match="conv_variant"):
        Engine(0, max_batch=2, tile_size=64, conv_variant=7)
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (64 << 20)
