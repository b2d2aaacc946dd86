"""What a plain streaming kernel reaches on this box (development aid): the practical ceiling for the HBM-bound kernels.
copy = read N + write N bytes (the depthwise conv's traffic shape: 4 B in, 4 B out per voxel and channel)."""
import torch, time
dev = torch.device("cuda:0")
for mb in (672, 1344, 2688):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device=dev)
    y = torch.empty_like(x)
    res = {}
    for name, fn, bytes_ in (("copy (read+write)", lambda: y.copy_(x), 8 * n), ("read (sum)", lambda: x.sum(), 4 * n),
                             ("write (fill)", lambda: y.fill_(1.5), 4 * n), ("scale (read+write, y = 2x)", lambda: torch.mul(x, 2.0, out=y), 8 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[name] = bytes_ / ms / 1e6
    print(f"{mb} MiB tensors: " + ", ".join(f"{k} {v:.0f} GB/s" for k, v in res.items()), flush=True)
