#!/bin/bash
# SQ / TCP / TCC counters of every kernel of one correlation section (10 units at 16000 x 3000), one rocprofv3
# --pmc pass per counter set (no tracing flags beside it).  Usage on the GPU box:
#   bash profiles/experiments/pmc_passes.sh <tag>      -> gpurun_out/<tag>/pmc_passes.json (+ counters_avail.txt)
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-pmc_passes}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_all.txt 2>&1 || true
grep -i -E "UTCL|TLB|TCP_|TCC_HIT|TCC_MISS|TCC_EA0_RD|TCC_REQ|TA_BUSY|TA_ADDR|TD_" $OUT/counters_all.txt | cut -c1-160 | sort -u | head -300 > $OUT/counters_avail.txt
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum"
)
i=0
for S in "${SETS[@]}"; do
  timeout -k 10 200 rocprofv3 --pmc $S --output-format csv -d $OUT/s$i -o pmc -- python3 profiles/experiments/rows_probe.py 0 > $OUT/s$i.out 2> $OUT/s$i.err || { echo "set $i failed"; tail -3 $OUT/s$i.err; }
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"fft_pass_ct_kernel<(\d+), *(\d+), *(\d+), *(\d+), *(\d+)", k)
        if m: name = "fft_pass_F%s_iok%s" % (m.group(1), m.group(5))
        else:
            m = re.search(r"(\w+_kernel)", k)
            name = m.group(1) if m else k[:40]
        grid = int(r.get("Grid_Size", 0) or 0)
        if name.endswith("iok2") and grid <= 50 * 256: name = "fft_window"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} | {"_launches": max(len(v) for v in d.values())} for k, d in acc.items()}
json.dump(res, open(out + "/pmc_passes.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items()):
    print(k, {c: round(v) for c, v in d.items()})
PY
rm -rf $OUT/s*/
echo done
