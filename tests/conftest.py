import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU checker (oracle/).  Built on demand; tests are the only importer besides
    __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """One oip context on cuda:0 through the C ABI.  Fails (does not skip) when the HIP
    library is missing: GPU tests must never pass on a fallback."""
    import torch
    import opticalimageprocessor_amd as oip
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    c = oip.Context(0)
    # one stream for torch and the library: tensors produced by torch kernels (randint, copies)
    # are ordered before the library's kernels that read them, and vice versa
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    c.set_stream(stream)
    yield c
    torch.cuda.synchronize()
    c.close()


@pytest.fixture()
def parity_log(request):
    """Records the measured |GPU - oracle| of a tolerance test next to the bar it was held to (DESIGN.md section 2 quotes
    this file): one JSON line per call in gpurun_out/parity_deltas.jsonl, best effort."""
    import json

    def log(**vals):
        rec = {"test": request.node.name}
        rec.update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in vals.items()})
        print("\nparity: %s" % json.dumps(rec))
        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "parity_deltas.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        except OSError:
            pass
    return log
