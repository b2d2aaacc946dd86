"""Numerics of Winograd variants under the split-f16 arithmetic of the 3^3 conv kernel (CPU emulation, numpy).

Emulates exactly what the GPU path does to ONE conv layer: input transform in f32 -> x*ascale split into f16 hi + f16 lo ->
transformed weights (f32) scaled so max|w| lands in (2048, 4096] and split -> products hi*hi + hi*lo + lo*hi accumulated in
f32 -> output transform in f32.  Variants: direct (no transform), F(2,3) along x (the shipped kernel), F(4,3) along x,
F(2x2,3x3) over (x,y).  Error metric = the one the op tests use: max |got - ref64| / max(|ref64|, rms(ref64)).
Run: python tools/exp/wino_numerics.py
"""
import numpy as np

rng = np.random.default_rng(0)


def split(x, scale):
    xs = (x * np.float32(scale)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def split_dot(a, b, asc, wsc):
    """sum_k a[..., k] b[k, ...] with the three split products, f32 accumulation (pairwise, numpy) -> unscaled f32."""
    ah, al = split(a, asc)
    bh, bl = split(b, wsc)
    acc = (ah @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (al @ bh).astype(np.float32)
    return acc / np.float32(asc * wsc)


def wscale(w):
    m = np.abs(w).max()
    return 2.0 ** np.floor(np.log2(4096.0 / m))


# transforms: (B^T input, G weights, A^T output) for F(m,3)
F23 = (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float),
       np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float),
       np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float))
F43 = (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                 [0, 4, 0, -5, 0, 1]], float),
       np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                 [0, 0, 1]], float),
       np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float))


def conv_case(cin, cout, nx=24, ny=8, other_taps=9, relu_in=True):
    """x [other_taps(dz,dy folded as independent K), ny, nx+2, cin]; weights [other_taps, 3(dx), cin, cout]; 1-D conv along x plus a
    sum over the other taps (they only deepen K).  For the 2-D variant a separate generator is used."""
    x = rng.standard_normal((other_taps, ny, nx + 2, cin)).astype(np.float32)
    if relu_in:
        x = np.maximum(x, 0)                     # operands are relu(InstanceNorm(.)) in the network
    w = (rng.standard_normal((other_taps, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    return x, w


def ref_1d(x, w):
    nx = x.shape[2] - 2
    out = np.zeros((x.shape[1], nx, w.shape[3]))
    for t in range(x.shape[0]):
        for dx in range(3):
            out += x[t, :, dx:dx + nx].astype(np.float64) @ w[t, dx].astype(np.float64)
    return out


def direct_1d(x, w, asc=16.0):
    nx = x.shape[2] - 2
    K = np.concatenate([x[t, :, dx:dx + nx] for t in range(x.shape[0]) for dx in range(3)], axis=-1)
    W = np.concatenate([w[t, dx] for t in range(x.shape[0]) for dx in range(3)], axis=0)
    return split_dot(K, W, asc, wscale(W))


def wino_1d(x, w, F, asc=16.0):
    BT, G, AT = F
    m, a = AT.shape                               # outputs per tile, alpha
    nx = x.shape[2] - 2
    assert nx % m == 0
    nt = nx // m
    # input transform in f32: tiles of alpha inputs with stride m
    d = np.stack([x[:, :, i * m:i * m + a] for i in range(nt)], axis=2)            # [T, ny, nt, a, cin]
    td = np.einsum("pa,tynac->tynpc", BT.astype(np.float32), d).astype(np.float32)
    u = np.einsum("pk,tkco->tpco", G, w.astype(np.float64)).astype(np.float32)      # weight transform (packer: f32 from f64-exact constants)
    ws = wscale(u)
    mm = np.zeros((x.shape[1], nt, a, w.shape[3]), np.float32)
    for p in range(a):
        K = np.concatenate([td[t, :, :, p] for t in range(x.shape[0])], axis=-1)    # [ny, nt, T*cin]
        W = np.concatenate([u[t, p] for t in range(x.shape[0])], axis=0)
        ah, al = split(K, asc)
        bh, bl = split(W, ws)
        mm[:, :, p] = ((ah @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (al @ bh).astype(np.float32))
    y = np.einsum("mp,ynpo->ynmo", AT.astype(np.float32), mm).astype(np.float32) / np.float32(asc * ws)
    return y.reshape(x.shape[1], nx, -1)


def ref_2d(x, w):
    """x [T(dz), ny+2, nx+2, cin], w [T, 3(dy), 3(dx), cin, cout]"""
    ny, nx = x.shape[1] - 2, x.shape[2] - 2
    out = np.zeros((ny, nx, w.shape[4]))
    for t in range(x.shape[0]):
        for dy in range(3):
            for dx in range(3):
                out += x[t, dy:dy + ny, dx:dx + nx].astype(np.float64) @ w[t, dy, dx].astype(np.float64)
    return out


def wino_2d(x, w, asc=16.0):
    BT, G, AT = F23
    ny, nx = x.shape[1] - 2, x.shape[2] - 2
    ty, tx = ny // 2, nx // 2
    d = np.stack([np.stack([x[:, j * 2:j * 2 + 4, i * 2:i * 2 + 4] for i in range(tx)], axis=1) for j in range(ty)], axis=1)   # [T, ty, tx, 4, 4, cin]
    td = np.einsum("pa,qb,tjiabc->tjipqc", BT.astype(np.float32), BT.astype(np.float32), d).astype(np.float32)
    u = np.einsum("pk,ql,tklco->tpqco", G, G, w.astype(np.float64)).astype(np.float32)
    ws = wscale(u)
    mm = np.zeros((ty, tx, 4, 4, w.shape[4]), np.float32)
    for p in range(4):
        for q in range(4):
            K = np.concatenate([td[t, :, :, p, q] for t in range(x.shape[0])], axis=-1)
            W = np.concatenate([u[t, p, q] for t in range(x.shape[0])], axis=0)
            ah, al = split(K, asc)
            bh, bl = split(W, ws)
            mm[:, :, p, q] = (ah @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (al @ bh).astype(np.float32)
    y = np.einsum("mp,nq,jipqo->jminqo"[:0] + "mp,nq,jipqo->jmino", AT.astype(np.float32), AT.astype(np.float32), mm).astype(np.float32)
    y = y / np.float32(asc * ws)
    return y.reshape(ty * 2, tx * 2, -1)


def err(got, ref):
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), np.sqrt(np.mean(ref ** 2)))))


def main():
    print("%-12s %10s %10s %10s %10s" % ("Cin->Cout", "direct", "F(2,3)-x", "F(4,3)-x", "F(2x2,3x3)"))
    for cin, cout in ((64, 64), (256, 128), (512, 256)):
        x, w = conv_case(cin, cout)
        ref = ref_1d(x, w)
        e = [err(direct_1d(x, w), ref), err(wino_1d(x, w, F23), ref), err(wino_1d(x, w, F43), ref)]
        x2 = np.maximum(rng.standard_normal((3, 10, 26, cin)).astype(np.float32), 0)
        w2 = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
        e.append(err(wino_2d(x2, w2), ref_2d(x2, w2)))
        print("%-12s %10.2e %10.2e %10.2e %10.2e" % (f"{cin}->{cout}", *e))


if __name__ == "__main__":
    main()
