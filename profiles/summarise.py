#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh into the small summaries that are committed:
   profiles/<tag>_<workload>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, kernel rows only)
   profiles/<tag>_bench_<workload>.json          (the bench line)
   profiles/<tag>_pmc_raw.json                   (per kernel: launches, mean FETCH_SIZE / WRITE_SIZE in KiB)
   profiles/traffic.json                         (HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, keyed like bench.py)
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", tag)
prof = os.path.join(out, "summary")          # copied into profiles/ by hand after the run (gpurun returns gpurun_out/ only)
os.makedirs(prof, exist_ok=True)


def bench_name(kernel):
    """library profiler name (bench.py's key) of a demangled kernel symbol"""
    m = re.search(r"fft_pass_ct_kernel<(\d+), *(\d+), *(\d+), *(\d+), *(\d+)", kernel)      # <F, VS, MODE, NT, IOK, radices...>
    if m:
        F, iok = int(m.group(1)), int(m.group(5))
        return "fft_pass_ct_kernel_F%d%s" % (F, {0: "", 1: "_pack", 2: "_peak"}[iok])
    m = re.search(r"fft_first_pass_up_kernel<(\d+)", kernel)
    if m:
        return "fft_pass_ct_kernel_F%d_pack" % int(m.group(1))       # the library profiles it under the pass's name
    if "fft_col128_peak_kernel" in kernel:
        return "fft_pass_ct_kernel_F128_peak"                         # the register-staged peak pass, profiled under the pass's name
    if "align_mss8_kernel" in kernel:
        return "align_mss_kernel"
    if "mss_split_flat_kernel" in kernel:
        return "mss_split_rrc_kernel"                                 # the flat form, profiled under the seam's name
    if "resize_cubic_v_x4_kernel" in kernel:
        return "resize_cubic_v_kernel"                                # profiled under the generic kernel's name
    m = re.search(r"([A-Za-z_0-9]+_kernel)\b", kernel)
    return m.group(1) if m else kernel


for extra, dst in (("host_rrc.json", "%s_host_rrc.json"), ("staging_probe.txt", "%s_staging_probe.txt"),
                   ("bench_prestitch_f16.json", "%s_bench_prestitch_f16.json"), ("bench_w12288.json", "%s_bench_w12288.json")):
    h = os.path.join(out, extra)
    if os.path.exists(h) and os.path.getsize(h):
        shutil.copy(h, os.path.join(prof, dst % tag))
raw, traffic = {}, {}
for w in ("default", "prestitch", "prestitch_fused", "rrc"):
    b = os.path.join(out, "bench_%s.json" % w)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(prof, "%s_bench_%s.json" % (tag, w)))
    b = os.path.join(out, "bench_%s_full.json" % w)                   # the verbose record behind the compact stdout line
    if os.path.exists(b):
        shutil.copy(b, os.path.join(prof, "%s_bench_%s_full.json" % (tag, w)))
    st = glob.glob(os.path.join(out, "trace_%s" % w, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(prof, "%s_%s_kernel_stats.csv" % (tag, w)))
    per = defaultdict(lambda: defaultdict(list))
    rows = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(out, "pmc_%s_%s" % (c, w), "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != c:
                    continue
                name = bench_name(row["Kernel_Name"])
                grid = int(row.get("Grid_Size", 0) or 0)
                if name.endswith("_peak") and grid <= 50 * 256 and "fft_col128_peak_kernel" not in row["Kernel_Name"]:
                    name = "fft_window_F" + name.split("_F")[1].split("_")[0]          # 25-tile window launches
                rows.append((name, grid, c, float(row["Counter_Value"])))
    # the band transforms run the same pass kernels on a quarter of the grid; the library profiles them under
    # "<pass>_band"
    gmax = defaultdict(int)
    for name, grid, c, v in rows:
        gmax[name] = max(gmax[name], grid)
    for name, grid, c, v in rows:
        if name.startswith("fft_pass_ct_kernel") and not name.endswith("_peak") and grid * 3 < gmax[name]:
            name += "_band"                     # (OIP_SPECTRAL_UP=1 runs them at a quarter of the width: "_quarter")
        if name == "fft_pass_ct_kernel_F32":
            name += "_band"
        per[name][c].append(v)
    raw[w], traffic[w] = {}, {}
    for name, d in per.items():
        if not re.fullmatch(r"[A-Za-z_0-9]+", name) or name.startswith("__amd") or name in ("reduce_kernel", "vectorized_elementwise_kernel", "distribution_elementwise_grid_stride_kernel"):
            continue                                                  # torch's own kernels (synthetic data generation)
        f = d.get("FETCH_SIZE", [])
        wr = d.get("WRITE_SIZE", [])
        if not f or not wr:
            continue
        fm, wm = sum(f) / len(f), sum(wr) / len(wr)
        raw[w][name] = {"launches": len(f), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm}
        traffic[w][name] = (2 * fm + wm) * 1024
about = ("HBM bytes per launch from rocprofv3 PMC counters on MI355X (gfx950, ROCm 7.2), collected in separate passes "
         "(--pmc FETCH_SIZE, then --pmc WRITE_SIZE) of the same bench.py command (profiles/collect.sh); traffic = "
         "(2*FETCH_SIZE + WRITE_SIZE)*1024 B: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream "
         "(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.  Kernels whose loads are narrower than 16 B/lane "
         "(align, mss_split, the 2-byte PAN reads of the first FFT pass) are uncalibrated on the read side.  "
         "Raw: profiles/%s_pmc_raw.json" % tag)
# a collect.sh call covers the workloads named in WLS (a gpurun call is limited to 20 minutes, every call gets a fresh box):
# workloads this call did not measure keep what the committed files hold for them
def merged(path, fresh, keep_key=None):
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except ValueError:
            old = {}
    out = dict(old)
    for w, v in fresh.items():
        if v or w not in out:
            out[w] = v
    return out, old

raw_path = os.path.join(root, "profiles", "%s_pmc_raw.json" % tag)
raw, _ = merged(raw_path, raw)
json.dump(raw, open(os.path.join(prof, "%s_pmc_raw.json" % tag), "w"), indent=1, sort_keys=True)
old_t = {}
if os.path.exists(os.path.join(root, "profiles", "traffic.json")):
    old_t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
src = dict(old_t.get("_source", {}))
for w, v in list(traffic.items()):
    if v:
        src[w] = "profiles/%s_pmc_raw.json" % tag
    elif w in old_t:
        traffic[w] = old_t[w]
t = {"_about": about, "_source": src}
t.update(traffic)
json.dump(t, open(os.path.join(prof, "traffic.json"), "w"), indent=1, sort_keys=True)
print("summaries written for", tag, {w: len(v) for w, v in traffic.items()})
