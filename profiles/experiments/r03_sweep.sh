#!/bin/bash
# Round-3 knob sweep on one box: every line of the output is one process of section_probe.py under one setting.
#   bash profiles/experiments/r03_sweep.sh OUTFILE  "VAR=val VAR2=val" "VAR=val" ...
OUT=$1; shift
: > $OUT
for setting in "" "$@"; do
  echo "== [$setting]" >> $OUT
  env $setting timeout -k 10 120 python3 profiles/experiments/section_probe.py --tag "$setting" >> $OUT 2>&1 || echo "FAILED [$setting]" >> $OUT
done
