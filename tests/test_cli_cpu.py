"""CPU suite, part 4: the `oip` command line -- flag validation and exit codes of main.cpp:92-343
(no GPU is touched before the arguments are accepted)."""
import os
import subprocess

import pytest

OIP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opticalimageprocessor_amd", "lib", "oip")


def run(args, cwd):
    env = dict(os.environ, LOGFILE=os.path.join(cwd, "oip.log"))
    return subprocess.run([OIP] + args, cwd=cwd, env=env, capture_output=True, text=True)


@pytest.fixture()
def files(tmp_path):
    for n in ("a.raw", "b.raw", "c.tiff", "k.csv"):
        (tmp_path / n).write_bytes(b"\0" * 64)
    return str(tmp_path)


def test_cli_is_built():
    assert os.path.exists(OIP), "run __graft_entry__.build()"


def test_version_and_help(files):
    r = run(["--version"], files)
    assert r.stdout.strip() == "1.1" and r.returncode == 255            # CLI::Success + 255 (main.cpp:262-263)
    assert run(["-h"], files).returncode == 255


def test_required_and_validation_errors(files):
    assert run(["stitch", "--image1", "a.raw"], files).returncode == 106                        # RequiredError
    assert run(["stitch", "--image1", "a.raw", "--image2", "b.raw", "-c", "1"], files).returncode == 105   # fold-cols >= 2
    assert run(["prestitch", "--pan1", "a.raw", "--pan2", "missing.raw"], files).returncode == 105          # ExistingFile
    assert run(["prestitch", "--pan1", "a.raw", "--pan2", "b.raw", "--stitch-overlap", "200", "-e", "101"], files).returncode == 105
    assert run(["--pan", "a.raw", "--mss", "b.raw", "--ibc-threshold", "1.0", "--no-rrc4mss"], files).returncode == 105
    assert run(["stitch", "--bogus"], files).returncode == 109                                  # ExtrasError
    assert run(["prestitch", "--pan1", "a.raw", "--pan2", "b.raw", "-s", "x"], files).returncode == 104     # ConversionError


def test_usage_errors_exit_254(files):
    r = run(["--pan", "a.raw", "--mss", "b.raw", "--do-rrc4pan"], files)
    assert r.returncode == 254 and "USAGE ERROR: RRC parameter file of PAN needed." in r.stdout           # main.cpp:290-292
    r = run(["--pan", "a.raw", "--mss", "b.raw", "--rrc-msb1", "k.csv"], files)
    assert r.returncode == 254 and "all MSS Bands" in r.stdout                                             # main.cpp:293-299


def test_runtime_errors_exit_2(files):
    r = run(["stitch", "--image1", "a.raw", "--image2", "c.tiff", "-c", "100"], files)
    assert r.returncode == 2 and "two images should be same type" in r.stdout                              # stitcher.h:31-33
    r = run(["prestitch", "--pan1", "a.raw", "--pan2", "b.raw"], files)
    assert r.returncode == 2 and "too small for SECTION" in r.stdout                                       # stitcher.h:61-63
    # PAN must be 4x the MSS size (preproc.h:565-567)
    r = run(["--pan", "a.raw", "--mss", "b.raw", "--no-rrc4mss"], files)
    assert r.returncode == 2 and "PAN file size does not match MSS file size" in r.stdout


def test_tiff_writer_roundtrip(tmp_path):
    """csrc/oip_tiff.hpp through a tiny driver: classic TIFF, 1 and 4 samples, odd sizes, many strips"""
    import numpy as np
    import _tiff
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include "oip_tiff.hpp"\n#include <cstdlib>\n#include <vector>\nint main(int c, char** v){int w=atoi(v[2]);long h=atol(v[3]);int s=atoi(v[4]);'
                   'std::vector<uint16_t> d((size_t)w*h*s);for(size_t i=0;i<d.size();++i)d[i]=(uint16_t)(i*2654435761u>>7);'
                   'OIPGPU::write_tiff_u16(v[1],d.data(),w,h,s,s==4);return 0;}\n')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "opticalimageprocessor_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    for w, h, spp in [(7, 5, 1), (1001, 333, 4), (3000, 3000, 1), (7500, 700, 4)]:
        out = tmp_path / ("o_%d_%d_%d.tiff" % (w, h, spp))
        subprocess.run([str(exe), str(out), str(w), str(h), str(spp)], check=True)
        img, tags, big = _tiff.read_tiff_u16(str(out))
        n = w * h * spp
        want = ((np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)) >> 7).astype(np.uint16)
        want = want.reshape(h, w, spp) if spp > 1 else want.reshape(h, w)
        if spp == 4:
            want = want[:, :, [2, 1, 0, 3]]
        assert not big and np.array_equal(img, want), (w, h, spp)
        if spp == 1:
            from PIL import Image
            with Image.open(str(out)) as im:
                assert np.array_equal(np.asarray(im), want)


def test_tiff_reader(tmp_path):
    """oip_tiff.hpp reader: its own files, a foreign uncompressed file, and a refused LZW file"""
    import numpy as np
    import _tiff
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "r.cpp"
    src.write_text('#include "oip_tiff.hpp"\n#include <cstdlib>\n#include <vector>\nint main(int c, char** v){int w,s;long h;std::vector<uint16_t> d;'
                   'try{OIPGPU::read_tiff_u16(v[1],&w,&h,&s,&d);}catch(std::exception&e){printf("ERR %s\\n",e.what());return 3;}'
                   'if(c>3){OIPGPU::write_tiff_u16(v[3],d.data(),w,h,s,false);}'
                   'FILE*f=fopen(v[2],"wb");fwrite(d.data(),2,d.size(),f);fclose(f);printf("%d %ld %d\\n",w,h,s);return 0;}\n')
    exe = tmp_path / "r"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "opticalimageprocessor_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    rng = np.random.default_rng(3)
    for shape in [(5, 7), (333, 1001, 4), (900, 7500, 4)]:
        a = rng.integers(0, 65536, shape).astype(np.uint16)
        p = tmp_path / "in.tiff"
        _tiff.write_tiff_u16(str(p), a)
        r = subprocess.run([str(exe), str(p), str(tmp_path / "o.raw"), str(tmp_path / "rt.tiff")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        w, h, s = map(int, r.stdout.split())
        assert (h, w) == shape[:2] and s == (shape[2] if len(shape) == 3 else 1)
        assert np.array_equal(np.fromfile(tmp_path / "o.raw", np.uint16).reshape(shape), a)
        # written again by the C++ writer (many strips for the large case) and read back by both readers
        back, _, _ = _tiff.read_tiff_u16(str(tmp_path / "rt.tiff"))
        assert np.array_equal(back.reshape(shape), a)
        r = subprocess.run([str(exe), str(tmp_path / "rt.tiff"), str(tmp_path / "o2.raw")], capture_output=True, text=True)
        assert r.returncode == 0 and np.array_equal(np.fromfile(tmp_path / "o2.raw", np.uint16).reshape(shape), a)
    # Pillow writes 16-bit gray; uncompressed is accepted, LZW is refused with a clear message
    g = rng.integers(0, 65536, (40, 50)).astype(np.uint16)
    Image.fromarray(g).save(str(tmp_path / "pil.tiff"))
    r = subprocess.run([str(exe), str(tmp_path / "pil.tiff"), str(tmp_path / "o3.raw")], capture_output=True, text=True)
    assert r.returncode == 0 and np.array_equal(np.fromfile(tmp_path / "o3.raw", np.uint16).reshape(40, 50), g)
    Image.fromarray(g).save(str(tmp_path / "lzw.tiff"), compression="tiff_lzw")
    r = subprocess.run([str(exe), str(tmp_path / "lzw.tiff"), str(tmp_path / "o4.raw")], capture_output=True, text=True)
    assert r.returncode == 3 and "compressed TIFF input" in r.stdout
    (tmp_path / "bad.tiff").write_bytes(b"not a tiff at all")
    r = subprocess.run([str(exe), str(tmp_path / "bad.tiff"), str(tmp_path / "o5.raw")], capture_output=True, text=True)
    assert r.returncode == 3


def test_task_subcommand_argument_errors(files):
    """`oip task` (fused flow of DOC/sample-task.sh): argument checking happens before any device use"""
    r = run(["task", "--pan1", "a.raw"], files)
    assert r.returncode == 106 and "is required" in r.stderr
    base = ["task", "--pan1", "a.raw", "--pan2", "b.raw", "--rrc1", "k.csv", "--rrc2", "k.csv", "--mss1", "a.raw", "--mss2", "b.raw",
            "--out-pan", "p.TIFF", "--out-mss", "m.TIFF"]
    for c in (1, 2):
        for b in range(1, 5):
            base += ["--rrc-mss%d-b%d" % (c, b), "k.csv"]
    r = run(base + ["--fold-cols-pan", "1", "--fold-cols-mss", "12"], files)
    assert r.returncode == 105 and "fold column value too small" in r.stderr
    r = run(base + ["--fold-cols-pan", "40", "--fold-cols-mss", "12", "-m", "1,2,3,4"], files)
    assert r.returncode == 107                                       # --band-map needs --GDAL
    r = run(base + ["--fold-cols-pan", "40", "--fold-cols-mss", "12", "--bogus"], files)
    assert r.returncode == 109
