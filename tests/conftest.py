import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import shufflingvideosfortsg_amd  # noqa: E402,F401  (before torch touches the GPU: _runtime_env)
import torch  # noqa: E402
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The CPU oracle's recurrences are thousands of small ops; with torch's default of one thread per hardware thread (128 on the GPU box) every
# one of them pays a 128-way fork / join: the 2-layer BiLSTM at [3, 512, 1024] takes 52.9 s with 128 threads, 1.2 s with 8
# (tools/oracle_threads_probe.py).  The GPU suite spent 9 of its 15 minutes there.
torch.set_num_threads(min(8, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a bare `pytest tests/`
    also works in the CPU-only build container."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Golden:
    """One tests/golden/<name>.npz: arrays as torch tensors, `w.`/`gw.` groups as dicts."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.a = {k: z[k] for k in z.files}

    def t(self, key, **kw):
        return torch.from_numpy(np.array(self.a[key], copy=True, order="C")).to(**kw)

    def group(self, prefix):
        n = len(prefix)
        return {k[n:]: torch.from_numpy(np.array(v, copy=True, order="C")) for k, v in self.a.items() if k.startswith(prefix)}

    @property
    def weights(self):
        return self.group("w.")

    @property
    def wgrads(self):
        return self.group("gw.")


@pytest.fixture
def golden():
    return Golden
