"""The scalar polynomial layer against the REFERENCE ITSELF.

Two pins of oracle/unary_polynomial.py (ACM algorithm 30 roots incl. its None => reject-Pade outcome, Brent zero):

* tests/golden/ref_poly.json -- outcomes of the reference's own translation unit (libsanm/unary_polynomial.cpp +
  third_party/BRENT, compiled by oracle/build_ref.py with g++ -O2) on the Pade denominators of every Pade build of
  the cuboid goldens, bob, armadillo_small and human_arap16, plus solve_eqn samples and the KAT of
  tests/pade.cpp:16-62; generator: tests/golden/make_ref_poly.py.  Compared BIT FOR BIT (hex floats).
* live, when oracle/_ref/libref_poly_O2.so exists (authoring container): random polynomials against the library.

The product's host root finder (sanm_amd/csrc/poly.cpp) is held to the same fixtures through the C ABI in
tests/test_device_ops.py::test_host_poly_helpers.
"""
import json
import os

import numpy as np
import pytest

from oracle import build_ref, unary_polynomial as up

HERE = os.path.dirname(os.path.abspath(__file__))


def _fx():
    return json.load(open(os.path.join(HERE, "golden", "ref_poly.json")))


def _f(h):
    return [float.fromhex(v) for v in h]


def _same(a, b):
    return a == b or (a != a and b != b)


def test_fixture_covers_both_outcomes():
    fx = _fx()
    valid = [r["valid"] for r in fx["roots"]]
    assert len(valid) >= 25 and any(valid) and not all(valid)
    assert {r["src"] for r in fx["roots"]} >= {"cuboid_nc", "cuboid_ni", "cuboid_arap", "bob", "armadillo_small",
                                               "human_arap16"}
    assert len(fx["solve_eqn"]) >= 10


def test_roots_valid_flag_and_roots_bit_exact():
    for rec in _fx()["roots"]:
        f = _f(rec["f"])
        got = up.real_roots(f)
        assert (got is not None) == rec["valid"], rec["src"]
        if got is None:
            assert up.roots(f, False) is None or rec["all"] is not None
            continue
        want = _f(rec["real"])
        assert len(got) == len(want) and all(_same(g, w) for g, w in zip(got, want)), rec["src"]
        full = up.roots(f, False)
        assert (full is None) == (rec["all"] is None)
        if full is not None:
            assert len(full) == len(rec["all"])
            for z, (re, im) in zip(full, rec["all"]):
                assert _same(z.real, float.fromhex(re)) and _same(z.imag, float.fromhex(im))


def test_solve_eqn_bit_exact():
    for rec in _fx()["solve_eqn"]:
        args = [float.fromhex(rec[k]) for k in ("xmin", "xmax", "b", "eps")]
        assert up.solve_eqn(_f(rec["f"]), *args) == float.fromhex(rec["x"])


def test_reference_kat_roots_3_and_minus_4():
    # tests/pade.cpp:16-62: q(x) * (x - 3)(x + 4); every root is a root, the real ones are found by both calls
    kat = _fx()["kat"]
    f = _f(kat["f"])
    allr = up.roots(f, False)
    assert len(allr) == len(f) - 1 == 9
    for z, (re, im) in zip(allr, kat["all"]):
        assert z.real == float.fromhex(re) and z.imag == float.fromhex(im)
        s = 0j
        for c in reversed(f):
            s = s * z + c
        assert abs(s.real) < 2e-4 and abs(s.imag) < 2e-4
    real = sorted(up.real_roots(f))
    assert sorted(z.real for z in allr if z.imag == 0) == real == sorted(_f(kat["real"]))
    assert min(abs(r - 3) for r in real) < 1e-9 and min(abs(r + 4) for r in real) < 1e-9


def test_degenerate_inputs():
    # (expected values: the reference library itself on these inputs, oracle/_ref/libref_poly_O2.so)
    assert up.roots([0.0, 1.0], False) == []                  # x: the zero root is stripped, not reported
    assert up.roots([0.0, 0.0, 1.0], False) == []             # x^2 likewise
    assert up.roots([0.0, 0.0, 2.0, 1.0], True) == [-2 + 0j]  # x^2 (x + 2)
    assert up.roots([-6.0, 1.0, 1.0], True) == [-3 + 0j, 2 + 0j]
    assert up.roots([1.0, 0.0, 1.0], True) == []              # x^2 + 1: no real root
    r = up.roots([1.0, 0.0, 1.0], False)
    assert [z.imag for z in r] == [1.0, -1.0] and all(z.real == 0 for z in r)
    r = up.roots([1.0, 0.0], True)                            # zero leading coefficient: -1/0
    assert len(r) == 1 and r[0].real == -np.inf
    r = up.roots([1.0, 2.0, 0.0, 0.0], True)                  # two zero leading coefficients: a root and two NaNs
    assert r[0] == -0.5 + 0j and len(r) == 3 and all(z.real != z.real for z in r[1:])
    with pytest.raises(AssertionError):
        up.roots([1.0], True)


needs_ref = pytest.mark.skipif(not os.path.exists(build_ref.lib_path("O2")),
                               reason="oracle/_ref not built (no /root/reference on this machine)")


@needs_ref
def test_live_against_reference_library():
    ref = build_ref.RefPoly("O2")
    rng = np.random.RandomState(11)
    for trial in range(200):
        deg = rng.randint(1, 22)
        kind = trial % 4
        if kind == 0:
            f = rng.uniform(-1, 1, deg + 1)
        elif kind == 1:
            f = rng.uniform(-1, 1, deg + 1) * 0.3 ** np.arange(deg + 1)
        elif kind == 2:
            f = np.poly(rng.uniform(-3, 3, deg))[::-1].copy()
        else:
            f = rng.uniform(-1, 1, deg + 1) * 10.0 ** rng.uniform(-8, 8, deg + 1)
        if trial % 7 == 0:
            f[0] = 0.0
        if trial % 11 == 0:
            f[-1] = 0.0
        for only_real in (True, False):
            a, b = ref.roots(f, only_real), up.roots(f, only_real)
            assert (a is None) == (b is None)
            if a is not None:
                assert len(a) == len(b)
                assert all(_same(x.real, y.real) and _same(x.imag, y.imag) for x, y in zip(a, b))
    for trial in range(200):
        f = rng.uniform(-1, 1, rng.randint(2, 20))
        f[0] = -abs(f[0]) - 0.01
        hi = 1.0
        while up.eval_poly(f, hi) <= 0:
            hi *= 2
            f[-1] = abs(f[-1]) + 0.1
        assert ref.solve_eqn(f, 0.0, hi) == up.solve_eqn(f, 0.0, hi)
        assert ref.eval(f, 0.37 * hi) == up.eval_poly(f, 0.37 * hi)
    for order in (2, 6, 16, 20):
        assert ref.stable_x_range(order) == up.stable_x_range(order)


@needs_ref
def test_fixture_matches_the_library_it_was_made_from():
    ref = build_ref.RefPoly("O2")
    for rec in _fx()["roots"]:
        r = ref.roots(_f(rec["f"]), True)
        assert (r is not None) == rec["valid"]
