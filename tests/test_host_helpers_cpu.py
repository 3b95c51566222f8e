"""The host helpers of the C ABI that need no GPU: bitwise compare / copy of x-sized arrays with and without the helper threads
(pk_host_threads), and the policy that decides whether a process keeps them (pockit_amd.runtime.host_helpers: measured on the
spot -- a host whose CPUs are time slices of fewer cores keeps none)."""
import numpy as np
import pytest

from pockit_amd import runtime


@pytest.fixture()
def lib():
    lib = runtime.load_library()
    yield lib
    lib.pk_host_threads(0)
    runtime.host_helpers_stopped()


@pytest.mark.parametrize("helpers", [0, 3])
def test_same_bits_and_copy_bits(lib, helpers):
    assert lib.pk_host_threads(helpers) == 0
    rng = np.random.default_rng(helpers)
    for n in (1, 511, 131_072, 300_001):
        a = rng.standard_normal(n)
        b = np.empty(n)
        assert lib.pk_copy_bits(b.ctypes.data, a.ctypes.data, n) == 0
        assert np.array_equal(a, b) and lib.pk_same_bits(a.ctypes.data, b.ctypes.data, n) == 1
        for where in (0, n // 2, n - 1):
            keep = b[where]
            b[where] = np.nextafter(keep, np.inf)
            assert lib.pk_same_bits(a.ctypes.data, b.ctypes.data, n) == 0
            b[where] = keep
        b[n - 1] = -a[n - 1] if a[n - 1] != 0 else 1.0        # (-0.0 against 0.0 would be a bitwise difference too)
        assert lib.pk_same_bits(a.ctypes.data, b.ctypes.data, n) == 0
    assert lib.pk_host_threads(17) != 0 and lib.pk_host_threads(-1) != 0


def test_helper_policy_is_measured_and_size_gated(lib, monkeypatch):
    monkeypatch.delenv("POCKIT_AMD_HOST_THREADS", raising=False)
    runtime.host_helpers_stopped()
    assert runtime.host_helpers(lib, 96_008) == 0                  # the 12k-node headline: x has 0.77 MB, nobody is started
    assert lib.pk_host_threads_hot() == 0
    k = runtime.host_helpers(lib, 600_012)                         # C5's x (4.8 MB): started only if a pass got a quarter faster
    assert 0 <= k <= 6
    assert runtime.host_helpers(lib, 600_012) == k                 # decided once per process
    monkeypatch.setenv("POCKIT_AMD_HOST_THREADS", "0")
    runtime.host_helpers_stopped()
    lib.pk_host_threads(0)
    assert runtime.host_helpers(lib, 600_012) == 0
