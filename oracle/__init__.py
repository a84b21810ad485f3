"""CPU oracle for the QUICK W4A16 GEMM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``quick_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker or the
timed CPU baseline -- never as the thing shipped.

Parity status: the reference (SqueezeBits/QUICK) ships no tests or golden
vectors for this path, so the oracle is pinned by fixtures
generated *from the reference's own Python* in the build container
(``tests/golden/gen_golden.py`` imports /root/reference by file path and writes
``tests/golden/*.npz``).  ``tests/test_oracle_golden.py`` checks every function
here against those fixtures.
"""
from .w4a16 import *  # noqa: F401,F403
