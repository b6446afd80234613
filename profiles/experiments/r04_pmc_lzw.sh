#!/bin/bash
# round 4: SQ counters of the LZW strip kernels on a product-sized image (r04_lzw_kernel_time.py)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lzw
rm -rf $OUT; mkdir -p $OUT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM")
i=0
for S in "${SETS[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $S --output-format csv -d $OUT/s$i -o pmc -- python3 profiles/experiments/r04_lzw_kernel_time.py ${1:-25000} > /dev/null 2> $OUT/s$i.err || tail -5 $OUT/s$i.err
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lzw/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("lzw_strips_kernel", "lzw_decode_kernel"):
            if name in r["Kernel_Name"]:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} | {"_launches": len(next(iter(d.values())))} for k, d in acc.items()}
json.dump(res, open("gpurun_out/r04_pmc_lzw.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
rm -rf $OUT
