"""CPU tests of sage_amd/csrc/core.h — the arithmetic shared by the HIP kernels and the host — compiled for the host
(tests/hostemu/core_emu.cpp) and compared with the oracle or with a plain restatement: Tolerance::bounds, trim_hits' k,
bounded_min_heapify through the compact list, the sorted-window counting shortcuts, select_most_intense_peak."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from sage_amd.api import Tolerance

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostemu", "core_emu.cpp")
LIB = os.path.join(HERE, "hostemu", "libcore_emu.so")
f32p = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def emu():
    deps = [SRC, os.path.join(HERE, "..", "sage_amd", "csrc", "core.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    lib.emu_tol_bounds.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p]
    lib.emu_trim_k.restype = C.c_uint32
    lib.emu_trim_k.argtypes = [C.c_uint64, C.c_uint32]
    lib.emu_max_fragment_charge.restype = C.c_uint32
    lib.emu_max_fragment_charge.argtypes = [C.c_int, C.c_uint32]
    lib.emu_clist_trim.restype = C.c_uint32
    lib.emu_clist_trim.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.emu_count_windows.restype = C.c_uint32
    lib.emu_count_windows.argtypes = [f32p, f32p, C.c_uint32, C.c_float, C.c_int]
    lib.emu_select_peak.restype = C.c_int
    lib.emu_select_peak.argtypes = [f32p, f32p, C.c_uint32, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int]
    return lib


def fp(a):
    return a.ctypes.data_as(f32p)


def test_tolerance_bounds_match_the_oracle(emu):
    rng = np.random.default_rng(1)
    kinds = {"ppm": 0, "pct": 1, "da": 2}
    for _ in range(2000):
        name = rng.choice(list(kinds))
        lo, hi = np.float32(-rng.uniform(0, 50)), np.float32(rng.uniform(0, 50))
        c = np.float32(rng.uniform(50, 6000))
        a, b = C.c_float(), C.c_float()
        emu.emu_tol_bounds(kinds[name], lo, hi, c, C.byref(a), C.byref(b))
        ol, oh = oracle_lib.tol_bounds(Tolerance(name, float(lo), float(hi)), float(c))
        assert np.float32(a.value) == ol and np.float32(b.value) == oh


def test_trim_k_and_fragment_charge(emu):
    for n in (0, 1, 7, 49, 50, 51, 99, 100, 101, 5000):
        for r in (1, 2, 5, 25, 26, 32, 60):
            want = min(max(50, min(2 * r, n)), n)  # 50.clamp(min(2r, len), len), scoring.rs:323-326
            assert emu.emu_trim_k(n, r) == want
    for user in (-1, 0, 1, 2, 3, 7):
        for z in range(1, 8):
            inner = z if user < 0 else user + 1
            assert emu.emu_max_fragment_charge(user, z) == max(2, min(z, inner))  # scoring.rs:239-247


def _bounded_min_heapify(a, k):  # heap.rs:7-60
    a = list(a)
    if len(a) <= k:
        return a

    def sift(idx):
        while True:
            l, r, s = 2 * idx + 1, 2 * idx + 2, idx
            if l < k and a[l] < a[s]:
                s = l
            if r < k and a[r] < a[s]:
                s = r
            if s == idx:
                return
            a[s], a[idx] = a[idx], a[s]
            idx = s

    for i in reversed(range(k // 2)):
        sift(i)
    for i in range(k, len(a)):
        if a[i] > a[0]:
            a[i], a[0] = a[0], a[i]
            sift(0)
    return a


def test_compact_list_trim_equals_bounded_min_heapify(emu):
    """core.h's CList keeps the first kmax logical entries verbatim and only the non-empty later ones; trimming it must
    give exactly what bounded_min_heapify + truncate give on the full vector (layout included)."""
    rng = np.random.default_rng(2)
    EMPTY = 0x0000FFFFFFFF0080
    for trial in range(300):
        n = int(rng.integers(1, 400))
        r = int(rng.choice([1, 2, 5, 30]))
        kmax = max(50, 2 * r)
        dens = rng.choice([0.02, 0.3, 0.9])
        items = []
        for i in range(n):
            if rng.random() < dens:
                items.append((int(rng.integers(1, 6)) << 48) | (i << 16) | (2 << 8) | 128)
            else:
                items.append(EMPTY)
        k = min(max(50, min(2 * r, n)), n)
        want = _bounded_min_heapify(items, k)[:k]
        buf = np.array(items, dtype=np.uint64)
        got_n = emu.emu_clist_trim(buf.ctypes.data_as(C.POINTER(C.c_uint64)), n, kmax, r)
        assert got_n == k and [int(x) for x in buf[:k]] == want, (trial, n, r)


def test_window_counting_shortcuts_agree(emu):
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(0, 160))
        m = np.sort(rng.uniform(100, 2000, n)).astype(np.float32)
        lo = (m + m * np.float32(-10.0) / np.float32(1e6)).astype(np.float32)
        hi = (m + m * np.float32(10.0) / np.float32(1e6)).astype(np.float32)
        for frag in list(rng.uniform(90, 2100, 20).astype(np.float32)) + list(m[:5]) + list(lo[:3]) + list(hi[:3]):
            want = int(np.sum((frag >= lo) & (frag <= hi)))
            for variant in (0, 1, 2):
                assert emu.emu_count_windows(fp(lo), fp(hi), n, np.float32(frag), variant) == want


def test_select_most_intense_peak_variants(emu):
    """spectrum.rs:134-159 restated directly: most intense peak inside the bounds, the last one on ties."""
    rng = np.random.default_rng(4)
    for _ in range(300):
        n = int(rng.integers(0, 150))
        m = np.sort(rng.uniform(100, 1500, n)).astype(np.float32)
        if n > 4:
            m[2] = m[1]  # duplicated masses
        it = rng.choice([1.0, 2.0, 5.0, 5.0, 9.0], n).astype(np.float32)
        for center in list(rng.uniform(90, 1600, 10).astype(np.float32)) + list(m[:6]):
            lo = np.float32(center) + np.float32(center) * np.float32(-20.0) / np.float32(1e6)
            hi = np.float32(center) + np.float32(center) * np.float32(20.0) / np.float32(1e6)
            inside = [i for i in range(n) if lo <= m[i] <= hi]
            want = -1
            best = np.float32(0.0)
            for i in inside:
                if it[i] >= best:
                    best, want = it[i], i
            for lock in (0, 1):
                assert emu.emu_select_peak(fp(m), fp(it), n, np.float32(center), 0, -20.0, 20.0, lock) == want
