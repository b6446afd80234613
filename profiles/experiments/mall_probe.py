"""Does the 256 MiB Infinity Cache keep freshly written data?  Time a read (sum) of a buffer right after it was
written, for several sizes, against the same read after 2 GB of other traffic."""
import torch, time
torch.cuda.init()
big = torch.empty(1 << 29, dtype=torch.float32, device="cuda")      # 2 GB
def t(fn, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
for mb in (16, 32, 64, 128, 192, 256, 512):
    x = torch.empty(mb << 18, dtype=torch.float32, device="cuda")
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    res = {}
    for mode in ("warm", "cold"):
        best = 1e9
        for _ in range(5):
            x.fill_(1.0)
            if mode == "cold":
                big.fill_(2.0)
            s0.record(); y = x.sum(); s1.record(); torch.cuda.synchronize()
            best = min(best, s0.elapsed_time(s1))
        res[mode] = mb / 1024 / (best * 1e-3)
    print("%4d MB: read right after write %.0f GB/s, after 2 GB of other writes %.0f GB/s" % (mb, res["warm"], res["cold"]), flush=True)
