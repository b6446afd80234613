"""End-to-end wall time of DOC/sample-task.sh's five commands (through the file system) vs the fused `oip task`
on two 12288 x 40000 CCD strips (PAN 983 MB each, MSS 246 MB each)."""
import os, subprocess, sys, tempfile, time
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R)
sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import _synth

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OIP = os.path.join(ROOT, "opticalimageprocessor_amd", "lib", "oip")
W, L, OV = 12288, 40000, 200
d = tempfile.mkdtemp(prefix="oiptask")
env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"))
pan1, pan2 = _synth.ccd_pair(L, W, OV, (3, -2), seed=11)
rng = np.random.default_rng(4)

def mss_of(pan, shifts):
    small = pan.astype(np.float32).reshape(L // 4, 4, W // 4, 4).mean(axis=(1, 3))
    return np.concatenate([np.clip(np.rint(np.roll(small, (sy, sx), (0, 1))), 0, 65535).astype(np.uint16) for sx, sy in shifts], axis=1)

for name, a in (("A_PAN-1.RAW", pan1), ("A_PAN-2.RAW", pan2), ("A_MSS-1.RAW", mss_of(pan1, [(1, 0), (0, 1), (-1, 0), (0, -1)])),
                ("A_MSS-2.RAW", mss_of(pan2, [(0, 1), (1, 0), (0, -1), (-1, 0)]))):
    a.tofile(os.path.join(d, name))

def csv(path, kb):
    with open(path, "w") as f:
        f.write("1\n%d\n0\n" % len(kb))
        for k, b in kb:
            f.write("%.6f , %.4f\n" % (k, b))

csv(os.path.join(d, "P1.csv"), _synth.lut(W, 1)); csv(os.path.join(d, "P2.csv"), _synth.lut(W, 2))
for c in (1, 2):
    for b in range(4):
        csv(os.path.join(d, "M%dB%d.csv" % (c, b + 1)), _synth.lut(W // 4, 30 + 4 * c + b))
stt = ["-s", "4", "-l", "8000", "--stt-threshold=-1"]      # timing run: accept every section
ibc = ["--ibc-sections", "2", "--ibc-threshold", "0"]

def run(args):
    r = subprocess.run([OIP] + args, cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

t0 = time.time()
run(["prestitch", "--pan1", "A_PAN-1.RAW", "--pan2", "A_PAN-2.RAW", "--rrc1", "P1.csv", "--rrc2", "P2.csv"] + stt)
run(["stitch", "--image1", "A_PAN-1.RRC.RAW", "--image2", "A_PAN-2.RRC.PRESTT.RAW", "--fold-cols", "200", "-o", "ref-PAN.TIFF"])
for c, s1 in ((1, "A_PAN-1.RRC.RAW"), (2, "A_PAN-2.RRC.PRESTT.RAW")):
    run(["--pan", s1, "--mss", "A_MSS-%d.RAW" % c] + ibc + sum([["--rrc-msb%d" % (b + 1), "M%dB%d.csv" % (c, b + 1)] for b in range(4)], []))
run(["stitch", "--image1", "A_MSS-1.ALIGNED.TIFF", "--image2", "A_MSS-2.ALIGNED.TIFF", "--fold-cols", "50", "-o", "ref-MSS.TIFF"])
t_ref = time.time() - t0
task = ["task", "--pan1", "A_PAN-1.RAW", "--pan2", "A_PAN-2.RAW", "--rrc1", "P1.csv", "--rrc2", "P2.csv", "--mss1", "A_MSS-1.RAW", "--mss2",
        "A_MSS-2.RAW", "--fold-cols-pan", "200", "--fold-cols-mss", "50", "--out-pan", "fused-PAN.TIFF", "--out-mss", "fused-MSS.TIFF"] + stt + ibc
for c in (1, 2):
    for b in range(4):
        task += ["--rrc-mss%d-b%d" % (c, b + 1), "M%dB%d.csv" % (c, b + 1)]
t0 = time.time()
run(task)
t_fused = time.time() - t0
same = all(open(os.path.join(d, a), "rb").read() == open(os.path.join(d, b), "rb").read()
           for a, b in (("ref-PAN.TIFF", "fused-PAN.TIFF"), ("ref-MSS.TIFF", "fused-MSS.TIFF")))
print("five commands %.2f s, fused task %.2f s, products byte-identical: %s" % (t_ref, t_fused, same))
subprocess.run(["rm", "-rf", d])
