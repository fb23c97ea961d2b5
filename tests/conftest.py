import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with one addition: libquicked_hip.so parses its QE_* switches once (qe_pool.h: SwitchTable), so a
    test that sets or removes one while the library is loaded has it parsed again (quicked_debug_reload_env) -- and once
    more when the test's changes are undone."""
    from quicked_amd import capi
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set_(name, value, prepend)
        capi.reload_env()

    def delenv(name, raising=True):
        del_(name, raising)
        capi.reload_env()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    capi.reload_env()
