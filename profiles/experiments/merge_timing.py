"""Throughput of oip_merge_subimages_be16 on 64 reference frames' worth of sub-images (2.0 GB in, 2.0 GB out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import opticalimageprocessor_amd as oip

ctx = oip.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
V, H, L, Cc = 5 * 64, 8, 256, 1536
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randint(-32768, 32767, (V, H, L, Cc), dtype=torch.int16, device="cuda", generator=g)
out = torch.empty(V * L, H * Cc, dtype=torch.int16, device="cuda")
for rep in range(3):
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(5):
        ctx.merge_subimages_be16(a, out, V, H, L, Cc)
    torch.cuda.synchronize()
ms, n = ctx.profile()["merge_be16_kernel"]
b = a.numel() * 4
print("merge_be16_kernel %.4f ms per launch, %.0f GB/s read+write (%.1f %% of 8 TB/s)" % (ms / n, b / (ms / n * 1e-3) / 1e9, b / (ms / n * 1e-3) / 8e12 * 100))
