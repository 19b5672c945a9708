import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Safety net: if the in-tree library is missing or older than its sources, (re)build it before any test loads it (hipcc cross-compiles
    gfx950 without a GPU; unchanged translation units are skipped)."""
    from mixermdm_amd import build as B
    if B.needs_build():
        B.build(verbose=False)


def load_golden(name):
    """Return (arrays, weights-by-prefix-getter) for tests/golden/<name>.npz."""
    import torch
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs = {k: d[k] for k in d.files}

    def weights(prefix, dims=()):
        """state_dict under 'w:<prefix>' as torch tensors, with the pe buffers re-created for `dims`."""
        from oracle.layers import pe_table
        W = {k[len("w:" + prefix):]: torch.from_numpy(v) for k, v in arrs.items() if k.startswith("w:" + prefix)}
        for key, D in dims:
            W[key] = pe_table(D)
        return W

    def t(key):
        return torch.from_numpy(np.asarray(arrs[key]))

    return arrs, weights, t


@pytest.fixture(scope="session")
def golden():
    return load_golden
