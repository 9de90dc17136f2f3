"""sage_amd/csrc/crlog.h — the correctly rounded ln the kernels and the host share — compiled for the host
(tests/hostemu/crlog_emu.cpp) and held to (a) a 60-digit decimal reference, (b) the oracle's independent correctly rounded ln
(libquadmath's 113-bit logq rounded to double) on a million arguments of the kind this path produces, with each phase of the
two-phase evaluation forced; and the platform libm's agreement rate, which is what "bit-identical to the reference on this host"
can mean for a reference that calls `f64::ln`."""
import ctypes as C
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np
import pytest

import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostemu", "crlog_emu.cpp")
LIB = os.path.join(HERE, "hostemu", "libcrlog_emu.so")
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def emu():
    csrc = os.path.join(HERE, "..", "sage_amd", "csrc")
    deps = [SRC, os.path.join(csrc, "crlog.h"), os.path.join(csrc, "crlog_tables.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        # (no -mfma: __builtin_fma then goes through libm's exact fma — the header must not depend on the host having the instruction)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    lib.emu_cr_log.argtypes = [C.c_int, dp, C.c_uint64, dp]
    return lib


def cr_log(lib, x, mode=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib.emu_cr_log(mode, x.ctypes.data_as(dp), len(x), out.ctypes.data_as(dp))
    return out


def domains(rng, n):
    sb = np.exp(rng.normal(6, 2, n)).astype(np.float32)
    sy = np.exp(rng.normal(6, 2, n)).astype(np.float32)
    return {
        # ScoreType::score: ((summed_b + 1) as f64) * ((summed_y + 1) as f64), scoring.rs:183-185
        "hyperscore": (sb + np.float32(1)).astype(np.float64) * (sy + np.float32(1)).astype(np.float64),
        # lambda = matched_peaks / scored_candidates, scoring.rs:499
        "lambda": rng.integers(1, 5000, n) / rng.integers(1, 3000, n),
        # lnfact's arguments: n and 2 pi n, scoring.rs:170-177
        "lnfact": np.concatenate([np.arange(1, 4097, dtype=np.float64), np.arange(1, 4097) * (np.pi * 2.0)]),
        "wide": np.exp(rng.uniform(-700, 700, n)),
        "near one": 1.0 + rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-52, -6, n),
        "cell edges": np.ldexp(1.0 + (rng.integers(0, 128, n) + rng.choice([0.0, 1e-13, 1 - 1e-13, 0.5], n)) / 128.0,
                               rng.integers(-3, 4, n)),
        "subnormal": np.ldexp(rng.uniform(0.5, 1, max(n // 20, 10)), rng.integers(-1074, -1020, max(n // 20, 10))),
    }


def test_against_a_decimal_reference(emu):
    getcontext().prec = 60
    rng = np.random.default_rng(7)
    for name, x in domains(rng, 3000).items():
        x = x[(x > 0) & np.isfinite(x)]
        ref = np.array([float(Decimal(float(v)).ln()) for v in x])  # (Decimal.ln and float() both round correctly)
        for mode in (0, 1):
            assert np.array_equal(cr_log(emu, x, mode), ref), (name, mode)
        y = cr_log(emu, x, 2)  # fast phase only: NaN where it cannot decide — never a wrong value
        assert np.array_equal(y[~np.isnan(y)], ref[~np.isnan(y)]), name


def test_against_the_oracles_quadmath_ln_on_a_million_arguments(emu):
    rng = np.random.default_rng(11)
    undecided = {}
    for name, x in domains(rng, 250_000).items():
        x = x[(x > 0) & np.isfinite(x)]
        ref = oracle_lib.ln_batch(x, 1)
        for mode in (0, 1):
            bad = np.flatnonzero(cr_log(emu, x, mode) != ref)
            assert len(bad) == 0, (name, mode, x[bad[:3]])
        y = cr_log(emu, x, 2)
        und = np.isnan(y)
        assert np.array_equal(y[~und], ref[~und]), name
        undecided[name] = float(und.mean())
    # the hot rescoring kernel carries the fast phase only and sends the spectrum of an undecided logarithm through the retry
    # pass: that must stay rare on the arguments it sees
    assert undecided["hyperscore"] < 1e-5 and undecided["lambda"] < 1e-5, undecided


def test_special_arguments(emu):
    x = np.array([1.0, 0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, 2.0, 0.5, np.finfo(np.float64).max, np.finfo(np.float64).tiny])
    with np.errstate(all="ignore"):
        want = np.log(x)
    for mode in (0, 1):
        y = cr_log(emu, x, mode)
        assert np.array_equal(np.isnan(y), np.isnan(want))
        ok = ~np.isnan(want)
        # (np.log is the platform libm: correctly rounded on these easy arguments)
        assert np.array_equal(y[ok], want[ok]), (mode, y, want)
    assert np.signbit(cr_log(emu, np.array([1.0]))[0]) == False  # noqa: E712  ln(1) = +0


def test_the_platform_libm_is_almost_correctly_rounded():
    """glibc >= 2.28 claims 0.52 ulp for log: on this path's arguments it returns the correctly rounded value all but ~1e-4 of the
    time — which is why a correctly rounded ln is the closest platform-independent contract to "the reference's value", and why
    the GPU suite also compares with the oracle in platform-libm mode (tests/test_gpu_parity.py)."""
    rng = np.random.default_rng(5)
    d = domains(rng, 200_000)
    for name in ("hyperscore", "lambda"):
        x = d[name]
        a, b = oracle_lib.ln_batch(x, 0), oracle_lib.ln_batch(x, 1)
        assert np.mean(a == b) > 0.998, name
        nz = b != 0  # (lambda == 1: ln = 0 on both sides)
        assert np.array_equal(a[~nz], b[~nz]) and np.max(np.abs(a[nz] - b[nz]) / np.abs(b[nz])) < 2.3e-16, name
