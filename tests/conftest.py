import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference-generated vectors (oracle/make_golden.py) as (cfg, state_dict, vectors, meta)."""
    import numpy as np
    import torch
    from oracle.tscd_oracle import TSCDConfig

    g = os.path.join(ROOT, "tests", "golden")
    meta = {k: v.item() for k, v in np.load(os.path.join(g, "tscd_tiny_meta.npz")).items()}
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(g, "tscd_tiny_weights.npz")).items()}
    vec = dict(np.load(os.path.join(g, "tscd_tiny_vectors.npz")))
    cfg = TSCDConfig(num_classes=int(meta["num_classes"]), depth=float(meta["depth"]), width=float(meta["width"]),
                     minimal_limit=int(meta["minimal_limit"]), maximal_limit=int(meta["maximal_limit"]))
    return cfg, sd, vec, meta
