import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def traces():
    d = np.load(GOLDEN / "cqi_traces_rbg64.npz")
    return {"cqi": d["cqi"], "mapping": d["mapping"], "hist": d["hist"]}


@pytest.fixture(scope="session")
def rs():
    """The product package with its HIP library built (hipcc cross-compiles without a GPU)."""
    from radiosaber_amd import build
    build.build()
    import radiosaber_amd
    return radiosaber_amd


def synth_cqi(seed, shape, hist):
    """Seeded CQI grids drawn from a histogram over CQI 1..15 (numpy; used for parity inputs)."""
    rng = np.random.default_rng(seed)
    p = np.asarray(hist, np.float64)
    p = p / p.sum()
    return rng.choice(np.arange(1, 16, dtype=np.uint8), size=shape, p=p).astype(np.uint8)
