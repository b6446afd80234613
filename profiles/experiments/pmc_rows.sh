#!/bin/bash
# SQ counters of corr_rows_kernel with and without its prefetch loads (OIP_ROWS_DBG masks 0 / 8 / 24)
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_rows
rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM")
for M in 0 8 24; do
  i=0
  for S in "${SETS[@]}"; do
    timeout -k 10 300 rocprofv3 --pmc $S --output-format csv -d $OUT/m${M}_s$i -o pmc -- python3 profiles/experiments/rows_probe.py $M > /dev/null 2> $OUT/m${M}_s$i.err || { tail -5 $OUT/m${M}_s$i.err; exit 1; }
    i=$((i+1))
  done
done
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/pmc_rows"
for m in (0, 8, 24):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/m%d_s*/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "corr_rows_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mask", m, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
rm -rf $OUT/m*_s*/
