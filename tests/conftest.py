import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lightning-generative-models_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
_PARITY_LOG = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def parity(request):
    """``parity(what, err, tol)``: assert a measured parity error against its tolerance AND put the measured
    value on record — printed (``pytest -s`` / ``-rP``) and, with ``LGM_PARITY_LOG=<file>``, written as JSON at
    the end of the session (the committed copy is profiles/rNN_parity_errors.json)."""
    def check(what, err, tol):
        err, tol = float(err), float(tol)
        rec = {"test": request.node.name, "what": what, "err": err, "tol": tol}
        _PARITY_LOG.append(rec)
        print(f"[parity] {request.node.name}: {what}: {err:.3e} (tol {tol:.1e})")
        assert err < tol, rec
        return err

    def record(what, **values):
        """Put measured values on record without a bound of their own (the asserting comparison is made elsewhere)."""
        _PARITY_LOG.append({"test": request.node.name, "what": what, **{k: float(v) for k, v in values.items()}})
    check.record = record
    return check


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("LGM_PARITY_LOG")
    if path and _PARITY_LOG:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(_PARITY_LOG, fh, indent=1)
