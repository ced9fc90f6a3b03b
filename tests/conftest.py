import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"), os.path.join(ROOT, "tests"),
          os.path.join(ROOT, "tests", "wave_emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth_mod():
    from rfid import synth
    return synth


@pytest.fixture(scope="session")
def emu_mod():
    import emu
    emu.lib()
    return emu


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import rfid
    ctx = rfid.Context(device=0)
    yield ctx
    ctx.close()
