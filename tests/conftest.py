import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# p2_circuit_set_zk_key / _seed (fixed blinding key: reproducible zk proofs for the byte-exact comparisons) are test hooks that
# the library refuses unless the process opts in
os.environ.setdefault("P2AES_ALLOW_FIXED_ZK_KEY", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    # the oracle does not scale past one socket's worth of threads (tools/orc_scale.py: 1.9 s per AES-GCM 1 KiB proof on 32
    # threads, 3.1 s on the 128 of the GPU boxes' hosts): cap it -- the GPU suite spends most of its time in the oracle
    L = oracle_lib.lib()
    L.orc_set_num_threads(min(L.orc_num_threads(), 32))
    return oracle_lib
