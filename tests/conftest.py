import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libref.so built from /root/reference")


def pytest_sessionstart(session):
    """On a GPU box the reference-built checkers MUST have travelled with the tree: oracle/_ref/libref.so (libdeflate 1.23,
    zlib, igzip, bgzf_compress) and oracle/_ref/cielbox_ref (the reference CLI).  Nine interop tests would otherwise lose
    their reference half and stay green.  HD_ALLOW_NO_REF=1 runs without them (a box that was handed a tree built
    where /root/reference does not exist)."""
    if not os.path.exists("/dev/kfd") or os.environ.get("HD_ALLOW_NO_REF") == "1":
        return
    missing = [p for p in ("libref.so", "cielbox_ref") if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", p))]
    if missing:
        raise pytest.UsageError("GPU box without oracle/_ref/{%s}: the reference legs of the interop tests would be skipped "
                                "silently.  Build them where /root/reference exists (python -c 'import __graft_entry__ as g; "
                                "g.build()') or set HD_ALLOW_NO_REF=1." % ",".join(missing))
