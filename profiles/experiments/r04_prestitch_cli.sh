# round 4: wall time of `oip prestitch` on two 30000 x 100000 strips in tmpfs (products: 3 x 6 GB on writer threads, one download
# lane each); the log's own lines tell how long each product took, the script how long the command took
D=/dev/shm/oip_prestitch
rm -rf $D; mkdir -p $D
python - <<'PY'
import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from opticalimageprocessor_amd import synth
W, L, OV = 30000, 100000, 200
kb1, kb2 = synth.lut(W), synth.lut(W, 5)
p1, p2 = synth.ccd_pair(0, L, W, OV, kb1, kb2, device="cuda")
p1.cpu().numpy().tofile("/dev/shm/oip_prestitch/C_PAN-1.RAW"); p2.cpu().numpy().tofile("/dev/shm/oip_prestitch/C_PAN-2.RAW")
for n, kb in (("P1.csv", kb1), ("P2.csv", kb2)):
    with open("/dev/shm/oip_prestitch/" + n, "w") as f:
        f.write("1\n%d\n0\n" % len(kb)); f.write("".join("%.6f , %.4f\n" % (k, b) for k, b in kb))
PY
R=$PWD
cd $D
for i in 1 2; do
  rm -f *.RRC.RAW *.PRESTT.RAW
  s=$(date +%s.%N)
  LOGFILE=$D/oip.log $R/opticalimageprocessor_amd/lib/oip prestitch --width 30000 --pan1 C_PAN-1.RAW --pan2 C_PAN-2.RAW --rrc1 P1.csv --rrc2 P2.csv -s 6 > run$i.log 2>&1
  e=$(date +%s.%N)
  echo "exit code $?"; tail -3 run$i.log
  python3 -c "print('run $i: wall %.3f s' % ($e - $s))"
  grep -E "bytes written|processed & written|bytes read|dx:" run$i.log
  ls -la *.RRC.RAW *.PRESTT.RAW | awk '{print $5, $9}'
done
cd $R; rm -rf $D
