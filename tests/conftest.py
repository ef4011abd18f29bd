import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(autouse=True)
def _repo_cwd(monkeypatch):
    # scene files name meshes relative to the working directory (src/Mesh.hs:218-221)
    monkeypatch.chdir(ROOT)


@pytest.fixture
def library_env(monkeypatch):
    """library_env(name, value="1"): set a RAYHS_* switch for one test.  The library reads its environment once, when it is
    loaded (never on the launch path), so it is told to read it again -- and once more when the test's environment is restored."""
    from rayhs_amd import lib as L

    def set_(name, value="1"):
        monkeypatch.setenv(name, value)
        L.load().rayhs_config_reload()
    yield set_
    monkeypatch.undo()
    L.load().rayhs_config_reload()


def pytest_sessionstart(session):
    # the C-ABI library and the oracle are build products (git-ignored); build them if this checkout has none
    lib = os.path.join(ROOT, "rayhs_amd", "bin", "librayhs_hip.so")
    orc = os.path.join(ROOT, "oracle", "librayhs_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()
