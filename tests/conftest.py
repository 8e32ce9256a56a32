import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"          # exists only in the build container, never on the GPU box
SHIMS = os.path.join(ROOT, "tests", "shims")


def pytest_configure(config):
    # the CPU oracle runs on torch's intra-op pool: size it by the cores we may really use (cgroup quota), not by
    # the visible CPU count — on the GPU box that is 16 vs 256 and a 20x difference in oracle time
    try:
        import torch
        from whisper_amd.utils import usable_cores
        torch.set_num_threads(usable_cores())
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "reference: needs the read-only reference checkout at /root/reference")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(os.path.join(REFERENCE, "whisper"))
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")


def write_report(name: str, obj) -> str:
    """Measured parity figures of a GPU test as JSON under gpurun_out/parity/ (merged back from the GPU box; the ones
    to be judged are copied into profiles/).  Never raises: a report is a by-product."""
    import json
    try:
        d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "parity")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name)
        with open(path, "w") as f:
            json.dump(obj, f, indent=1, default=float)
        return path
    except Exception:      # noqa: BLE001
        return ""
