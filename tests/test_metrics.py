"""LossMeter (evomotion_amd/metrics.py; evo_motion_networks/src/metrics.cpp:12-75) against the reference's own known answers —
evo_motion_networks/tests/src/test_metrics.cpp:20-25, the only golden values the reference's test suite holds — and the rest of
its behaviour (empty meter, window changes, the fixed six-digit strings, the literal CSV behaviour)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (values, window_size, expected loss()): test_metrics.cpp:20-25
REFERENCE_CASES = [([1.0, 2.0, 1.0, 2.0], 4, 1.5), ([1.0, 2.0, 1.0, 2.0], 2, 1.5), ([1.0, 1.0, 2.0, 2.0], 2, 2.0)]


def _meter_cls():
    # metrics.py has no GPU dependency; import it without the package's ctypes loader
    import importlib.util
    spec = importlib.util.spec_from_file_location("evm_metrics", os.path.join(ROOT, "evomotion_amd", "metrics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.LossMeter


@pytest.mark.parametrize("values,window,expected", REFERENCE_CASES)
def test_reference_known_answers(values, window, expected):
    m = _meter_cls()("test", window)
    for v in values:
        m.add(v)
    assert m.loss() == expected          # ASSERT_EQ in the reference: exact


def test_the_rest_of_the_meter(tmp_path):
    LossMeter = _meter_cls()
    m = LossMeter("steps", 64)
    assert m.loss() == 0.0 and m.to_string() == "steps = 0.000000"        # default value, std::fixed << setprecision(6)
    for v in range(100):
        m.add(v)
    assert len(m.values) == 64 and m.values[0] == 36.0 and m.curr_step == 100
    assert m.loss() == sum(range(36, 100)) / 64.0
    m.set_window_size(2)                                                    # takes effect at the next add (metrics.cpp:17-19)
    assert len(m.values) == 64
    m.add(1.0)
    assert m.values == [99.0, 1.0] and m.loss() == 50.0
    unbounded = LossMeter("all", None)                                      # std::nullopt: no window
    for v in range(1000):
        unbounded.add(1.0)
    assert len(unbounded.values) == 1000 and unbounded.loss() == 1.0
    # fp32 like the reference's float accumulate
    f = LossMeter("f", 4)
    for v in (0.1, 0.2, 0.3):
        f.add(v)
    import numpy as np
    want = np.float32(np.float32(np.float32(np.float32(0.1) + np.float32(0.2)) + np.float32(0.3)) / np.float32(3))
    assert f.loss() == float(want)
    # to_csv, literally metrics.cpp:37-49: the file is reopened (truncated) for the value line
    m.to_csv(str(tmp_path))
    assert open(tmp_path / "steps.csv").read() == "101,50.000000\n"
    m.add(3.0)
    m.to_csv(str(tmp_path))
    assert open(tmp_path / "steps.csv").read() == "102,2.000000\n"


def test_exported_by_the_package(hip_lib):
    import evomotion_amd
    m = evomotion_amd.LossMeter("test", 2)
    for v in (1.0, 1.0, 2.0, 2.0):
        m.add(v)
    assert m.loss() == 2.0
