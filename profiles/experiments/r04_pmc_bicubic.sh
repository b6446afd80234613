#!/bin/bash
# round 4: vector-instruction counts of the re-priced bicubic kernels (after the changes of DESIGN 4.2, second half)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bicubic
rm -rf $OUT; mkdir -p $OUT
S="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"
i=0
for F in "--workload default" "--workload prestitch" "--workload prestitch --fused" "--workload prestitch --fused --fp16-accumulate"; do
  timeout -k 10 300 rocprofv3 --pmc $S --output-format csv -d $OUT/s$i -o pmc -- python3 bench.py $F --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-cli > /dev/null 2> $OUT/s$i.err || tail -5 $OUT/s$i.err
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_bicubic/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("align_mss8_kernel", "remap_shift8_kernel", "remap_shift8_rrc_kernel<false>", "remap_shift8_rrc_kernel<true>", "remap_shift8_f16_kernel"):
            if name in r["Kernel_Name"]:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} | {"_launches": len(next(iter(d.values())))} for k, d in acc.items()}
json.dump(res, open("gpurun_out/r04_pmc_bicubic.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
rm -rf $OUT
