import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import capi
    capi.build()
    return capi.lib()


# ---- tuning knobs of the GPU tests -----------------------------------------------------------------------------------
# The schedule tests name their knobs by the DEVELOPER build's environment variables (CLSIMHIP_KERNEL, CLSIMHIP_SLICES, ...:
# the names tools/scan_env.sh and tools/stress_schedules.py use with a `make DEVELOPER=1` library).  The library under test is
# the DEFAULT build, which reads no tuning from the environment (tests/test_abi.py checks that): this fixture translates what a
# test has put into os.environ into clsimhip_set_tuning calls on every converter made through converter.initializeHIP.
TUNING_ENV = {"CLSIMHIP_K_NEW": "k_new", "CLSIMHIP_K_SEARCH": "k_search", "CLSIMHIP_SLICES": "slices", "CLSIMHIP_K_POP": "k_pop",
              "CLSIMHIP_K_WAIT": "k_wait", "CLSIMHIP_K_AIM": "k_aim", "CLSIMHIP_RESULT_MIN_RECORDS": "result_min_records",
              "CLSIMHIP_POOL_R": "pool_ring", "CLSIMHIP_POOL_MIN_STEPS": "pool_min_steps", "CLSIMHIP_GRID": "grid",
              "CLSIMHIP_NO_FAST": "generic_kernels", "CLSIMHIP_PROX_N": "string_map_cells", "CLSIMHIP_DOM_PROX_N": "dom_map_cells"}


def tuning_from_env(env=None):
    env = os.environ if env is None else env
    tuning = {key: int(env[name]) for name, key in TUNING_ENV.items() if name in env}
    if "CLSIMHIP_KERNEL" in env:
        tuning["kernel"] = "pool" if env["CLSIMHIP_KERNEL"] == "pool" else "classic"
    if "CLSIMHIP_POOL_INDEX_BITS" in env:
        tuning["pool_max_steps"] = (1 << int(env["CLSIMHIP_POOL_INDEX_BITS"])) - 1
    if "CLSIMHIP_NO_NAMED_SEARCH" in env:
        tuning["named_search"] = 0 if env["CLSIMHIP_NO_NAMED_SEARCH"] == "1" else 1
    return tuning


@pytest.fixture(autouse=True)
def _tuning_through_the_c_abi(monkeypatch):
    from clsim_amd import converter as CV
    plain = CV.initializeHIP

    def initialize_with_the_tests_tuning(*args, **kwargs):
        tuning = dict(tuning_from_env())
        tuning.update(kwargs.pop("tuning", None) or {})
        return plain(*args, tuning=tuning, **kwargs)
    monkeypatch.setattr(CV, "initializeHIP", initialize_with_the_tests_tuning)
