"""What bounds the host side of `end_to_end` (bench.py)?  VERDICT r2 item 8.
  1. the link alone: pinned host memory -> HBM and back (torch pinned tensors, 2 GiB, best of 3);
  2. the staging ring on a pageable buffer: wall rate, and where the uploading thread spent its time (oip_stage_stats: seconds
     in pageable -> pinned pool copies vs seconds waiting for a ring slot whose DMA had not finished);
  3. the same with 8 / 16 / 32 copy threads (OIP_HOST_COPY_THREADS is read once per process: child processes);
  4. NUMA: nodes of the box, the CPUs this process may use, the GPU's node."""
import glob, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

def child():
    import opticalimageprocessor_amd as oip
    ctx = oip.Context(0)
    W, L = 30000, 65536                      # 3.9 GB, pageable
    host = np.empty((L, W), np.uint16); host[:] = 1234      # first touch before the clock starts
    dev = torch.zeros(L, W, dtype=torch.uint16, device="cuda")
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        ctx.stage_stats(reset=True)
        t = time.perf_counter(); ctx.upload_staged(dev, host); ctx.sync(); dt = time.perf_counter() - t
        c, w, b, n = ctx.stage_stats()
        if best is None or dt < best[0]:
            best = (dt, c, w)
    dt, c, w = best
    print(json.dumps({"threads": oip.load_library().oip_stage_threads(), "upload_GBs": host.nbytes / dt / 1e9, "wall_s": dt,
                      "copy_s": c, "slot_wait_s": w, "other_s": dt - c - w}), flush=True)
    ctx.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
    sys.exit(0)

n = 1 << 30
pin = torch.empty(n, dtype=torch.int16).pin_memory()
devt = torch.empty(n, dtype=torch.int16, device="cuda")
for name, dst, src in (("H2D pinned", devt, pin), ("D2H pinned", pin, devt)):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("%s: %.1f GB/s (2 GiB)" % (name, 2 * n / best / 1e9), flush=True)
del pin, devt
for thr in ("8", "16", "32"):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, OIP_HOST_COPY_THREADS=thr), capture_output=True, text=True)
    print("staging ring, OIP_HOST_COPY_THREADS=%s: %s" % (thr, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("NUMA nodes: %d; cpus allowed: %d" % (len(nodes), len(os.sched_getaffinity(0))))
for nd in nodes:
    try:
        print("  %s cpus %s" % (os.path.basename(nd), open(nd + "/cpulist").read().strip()))
    except OSError:
        pass
for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node")):
    try:
        print("  %s: numa_node %s" % (card.split("/")[4], open(card).read().strip()))
    except OSError:
        pass
print("allowed cpus:", sorted(os.sched_getaffinity(0)))
