"""Per-kernel times of one plain 16000 x 3000 f32 phase correlation (pack loader without up-sampling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import opticalimageprocessor_amd as oip

ctx = oip.Context(0)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
ctx.set_stream(s.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.rand((16000, 3000), device="cuda", generator=g)
b = torch.roll(a, (3, -2), (0, 1)).contiguous()
for rep in range(3):
    ctx.profile_enable(True)
    ctx.profile_reset()
    r = ctx.phase_correlate_f32(a, b, 16000, 3000)
    torch.cuda.synchronize()
print(r)
for k, (ms, cnt) in ctx.profile().items():
    print("%-40s %.4f ms x%d" % (k, ms / max(cnt, 1), cnt))
