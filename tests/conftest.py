import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def leaf_goldens():
    import numpy as np

    return np.load(os.path.join(GOLDEN, "ref_leaf_goldens.npz"))


# ---- kernel coverage of a GPU session ----------------------------------------------------------------------------------
def _launch_sites():
    """Every label a kernel launch of the library is checked under (epa::check_launch("...") in csrc/*.hip)."""
    import glob
    import re

    labels = set()
    for f in glob.glob(os.path.join(ROOT, "echopype_amd", "csrc", "*.hip")):
        labels.update(re.findall(r'check_launch\("([^"]+)"\)', open(f).read()))
    return labels


def pytest_sessionfinish(session, exitstatus):
    """After a session that launched kernels: which launch sites of the library ran in this process, which never did
    (gpurun_out/kernel_coverage.txt; tests/test_zz_kernel_coverage.py asserts on a whole-suite run)."""
    mod = sys.modules.get("echopype_amd._lib")
    if mod is None:
        return
    seen = set(mod.launched_kernels())
    if not seen:
        return
    sites = _launch_sites()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "kernel_coverage.txt"), "w") as fh:
        fh.write(f"{len(seen & sites)} of {len(sites)} launch sites ran in this pytest process ({len(session.items)} tests)\n")
        fh.write("never launched: " + (", ".join(sorted(sites - seen)) or "none") + "\n")
        fh.write("launched: " + ", ".join(sorted(seen)) + "\n")
