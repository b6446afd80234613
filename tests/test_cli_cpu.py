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
