"""Exercise the raster I/O staging entry points step by step (prints after every step)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import opticalimageprocessor_amd as oip

def say(*a):
    print(*a, flush=True)

ctx = oip.Context(0)
say("ctx ok, pool threads", oip.load_library().oip_stage_threads())
W, L = 30000, 8192
rng = np.random.default_rng(1)
host = rng.integers(0, 4096, (L, W), dtype=np.uint16)
dev = torch.zeros(L, W, dtype=torch.uint16, device="cuda")
torch.cuda.synchronize()
t = time.time(); ctx.upload_staged(dev, host); ctx.sync(); say("upload_staged %.1f GB/s" % (host.nbytes / (time.time() - t) / 1e9))
back = np.empty_like(host)
t = time.time(); ctx.download_staged(back, dev); say("download_staged %.1f GB/s" % (host.nbytes / (time.time() - t) / 1e9), np.array_equal(back, host))
path = "/tmp/staging_probe.raw"
t = time.time(); ctx.write_device_to_file(dev, host.nbytes, path); say("write_device_to_file %.2f s" % (time.time() - t), os.path.getsize(path) == host.nbytes)
dev2 = torch.zeros_like(dev)
t = time.time(); got = ctx.read_file_to_device(path, dev2); ctx.sync(); say("read_file_to_device %.2f s" % (time.time() - t), got == host.nbytes, bool((dev2.view(torch.int16) == dev.view(torch.int16)).all()))
got, tk = ctx.read_file_to_device(path, dev2, want_ticket=True); ctx.stage_wait(tk); ctx.sync(); say("ticket path ok", tk)
kb = np.stack([np.ones(W), np.zeros(W)], 1)
hb = host.copy()
t = time.time(); ctx.rrc_u16_host(hb, kb); say("rrc_u16_host %.2f Gpix/s" % (hb.size / (time.time() - t) / 1e9), np.array_equal(hb, host))
t = time.time(); ctx.rrc_u16_host(hb, kb); say("rrc_u16_host again %.2f Gpix/s" % (hb.size / (time.time() - t) / 1e9))
os.remove(path)
ctx.close()
say("done")
