"""tests/_parity.py: the parity bar follows from what the host's libm is. Both modes are exercised here through TRAY_PARITY_MODE, and the probe
itself runs once: on the build image (glibc 2.35, x86-64 with FMA) it must say "bit"."""
import os
import platform

import numpy as np
import pytest

import _parity


@pytest.fixture
def fresh(monkeypatch):
    def set_mode(m):
        if m is None: monkeypatch.delenv("TRAY_PARITY_MODE", raising=False)
        else: monkeypatch.setenv("TRAY_PARITY_MODE", m)
        _parity.reset()
    yield set_mode
    monkeypatch.delenv("TRAY_PARITY_MODE", raising=False)
    _parity.reset()


def records(n=20000, seed=3):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 8), np.float32)
    a[:, :3] = rng.uniform(0, 1, (n, 3)); a[:, 5] = rng.integers(1, 9, n); a[:, 6] = 3 * a[:, 5]
    return a


def last_bits(a, share, seed=5):
    """`share` of the samples with the last bit of one radiance channel changed: what another libm does to a sample on the same path"""
    b = a.copy()
    rng = np.random.default_rng(seed)
    k = rng.random(len(a)) < share
    u = b[:, 0].view(np.uint32)
    u[k] ^= 1
    return b


def test_the_probe_on_this_host(fresh):
    fresh(None)
    ok, detail = _parity.probe()
    print(_parity.glibc_version(), platform.machine(), "probe:", ok, detail)
    assert _parity.mode() == ("bit" if ok else "bars")
    if _parity.glibc_version() == "glibc 2.35" and platform.machine() == "x86_64":
        assert ok, detail   # the build image: the libm the restatement was made from


def test_bit_mode_demands_every_bit(fresh):
    fresh("bit")
    a = records()
    assert _parity.check_samples(a, a.copy(), "identical")[0] == 1.0
    with pytest.raises(AssertionError):
        _parity.check_samples(a, last_bits(a, 0.001), "one sample in a thousand")
    b = a.copy(); b[7, 5] += 1
    with pytest.raises(AssertionError):
        _parity.check_samples(a, b, "one path flipped")
    assert _parity.film_bar(5e-5) == 5e-5


def test_bars_mode_is_the_round_4_bars(fresh):
    fresh("bars")
    a = records()
    share, flipped, r = _parity.check_samples(a, last_bits(a, 0.2), "a fifth of the samples differ in their last bit")
    assert 0.75 < share < 0.85 and flipped == 0 and r < 1e-6
    with pytest.raises(AssertionError):
        _parity.check_samples(a, last_bits(a, 0.5), "half of the samples differ")
    b = a.copy(); b[:100, 5] += 1   # 0.5 % of the samples on another path
    with pytest.raises(AssertionError):
        _parity.check_samples(a, b, "too many flipped paths")
    b = a.copy(); b[:, 0] += 1e-3
    with pytest.raises(AssertionError):
        _parity.check_samples(a, b, "a real error, not last bits")
    assert _parity.film_bar(5e-5) == 1e-4


def test_a_failing_probe_selects_the_bars_loudly(fresh, monkeypatch, capsys):
    fresh(None)
    monkeypatch.setattr(_parity, "probe", lambda: (False, "sinf(0x1p-3): libm 0x1.fd5bap-4, device source 0x1.fd5bcp-4"))
    assert _parity.mode() == "bars"
    assert "PARITY BARS RELAXED" in capsys.readouterr().err
    assert "bars" in _parity.describe()
