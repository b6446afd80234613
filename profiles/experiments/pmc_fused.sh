#!/bin/bash
# SQ counters of the fused cross-CCD step's kernels (bench.py --workload prestitch --fused), one rocprofv3 --pmc pass per set.
#   [WL="--workload default"] bash profiles/experiments/pmc_fused.sh <tag> [extra bench flags]   -> gpurun_out/<tag>/pmc_fused.json
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-pmc_fused}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
)
i=0
for S in "${SETS[@]}"; do
  timeout -k 10 200 rocprofv3 --pmc $S --output-format csv -d $OUT/s$i -o pmc -- python3 bench.py ${WL:---workload prestitch --fused} --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs "$@" > $OUT/s$i.out 2> $OUT/s$i.err || { echo "set $i failed"; tail -3 $OUT/s$i.err; }
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        if "F16" in r["Kernel_Name"] or "Lb1" in r["Kernel_Name"]: name += "<1>"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    if not any(x in k for x in ("remap", "rrc", "stitch", "align", "split", "pack_bands")): continue
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["_launches"] = max(len(v) for v in d.values())
json.dump(res, open(out + "/pmc_fused.json", "w"), indent=1, sort_keys=True)
for k, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(k, "launches", d["_launches"], "waves %.0f" % d.get("SQ_WAVES", 0), "VALU active %.1f%%" % (100 * d.get("SQ_ACTIVE_INST_VALU", 0) / wc),
              "wait any %.1f%%" % (100 * d.get("SQ_WAIT_ANY", 0) / wc), "wait inst %.1f%%" % (100 * d.get("SQ_WAIT_INST_ANY", 0) / wc),
              "LDS %.1f%%" % (100 * d.get("SQ_ACTIVE_INST_LDS", 0) / wc), "VALU insts/wave %.0f" % (d.get("SQ_INSTS_VALU", 0) / max(d.get("SQ_WAVES", 1), 1)),
              "SALU/wave %.0f" % (d.get("SQ_INSTS_SALU", 0) / max(d.get("SQ_WAVES", 1), 1)), "SMEM/wave %.0f" % (d.get("SQ_INSTS_SMEM", 0) / max(d.get("SQ_WAVES", 1), 1)))
PY
rm -rf $OUT/s[0-9]
