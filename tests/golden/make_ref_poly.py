"""Generate tests/golden/ref_poly.json from the REFERENCE's own unary_polynomial translation unit
(authoring container only: needs /root/reference and g++).

oracle/build_ref.py compiles /root/reference/libsanm/unary_polynomial.cpp (+ utils.cpp, third_party/BRENT) into
oracle/_ref/libref_poly_{O2,native}.so.  This script runs the oracle's continuations of

    the three Pade cuboid goldens (tests/golden/make_golden.py CASES: cuboid_nc, cuboid_ni, cuboid_arap),
    bob, armadillo_small, human_arap16 (data/meshes/*.json = BASELINE configs 2, 3, 5),

records the denominator polynomial m_d of EVERY Pade build along them (pade.cpp:84-90) and every polynomial handed to
solve_eqn by solve_a (pade.cpp:191-201, anm.cpp:186-191), feeds them to the reference library and writes

    roots:     {"src", "f": coefficients (hex floats, low order first),
                "valid": reference roots(f, only_real=true) returned a value (pade.cpp:113-116 rejects the
                          approximant otherwise), "real": its real roots in the order found,
                "all": all roots [[re, im] ...] of roots(f, only_real=false) or null,
                "native_valid": the same flag from the -march=native build (informational: gcc contracts FMAs
                          there, and Bairstow's outcome on these ill-scaled polynomials moves with it)}
    solve_eqn: {"f", "xmin", "xmax", "b", "eps", "x"}
    kat:       the reference's own known-answer polynomial of tests/pade.cpp:16-62 ((x-3)(x+4)*q(x)) with q drawn
               here, roots from the reference library.

The fixtures pin the g++ -O2 variant: strict IEEE double evaluation of the source, reproducible with any compiler
given -ffp-contract=off, which is how oracle/unary_polynomial.py (Python floats) and sanm_amd/csrc/poly.cpp are
evaluated.  Everything is stored as float.hex() so the comparison is bit for bit.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import build_ref, fea as ofea, pade as opade, unary_polynomial as up  # noqa: E402
from sanm_amd import fea as dfea  # noqa: E402  (mesh / config readers only; no device is touched)


def hx(v):
    return float(v).hex()


def main(full=True):
    build_ref.build()
    ref, ref_native = build_ref.RefPoly("O2"), build_ref.RefPoly("native")
    polys, eqns = [], []
    src = [None]

    orig_est = opade.PadeApproximation.estimate_valid_range
    orig_solve = up.solve_eqn

    def est(self, start, eps, limit=0.0):
        if self.d:
            polys.append((src[0], list(self.d)))
        return orig_est(self, start, eps, limit)

    def solve(f, xmin, xmax, b=0.0, eps=1e-6):
        x = orig_solve(f, xmin, xmax, b, eps)
        eqns.append((src[0], list(f), xmin, xmax, b, eps))
        return x

    opade.PadeApproximation.estimate_valid_range = est
    up.solve_eqn = solve
    try:
        ns = {}
        text = open(os.path.join(HERE, "make_golden.py")).read()
        exec(text[text.index("CASES = {"):text.index("def anm_cases")], ns)
        for name in ("cuboid_nc", "cuboid_ni", "cuboid_arap"):
            dims, sp, cfg = ns["CASES"][name]
            src[0] = name
            _, solver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
            ofea.run_anm(solver)
            print(name, "steps", solver.get_nr_iter(), "polys", len(polys), flush=True)
        if full:
            for name in ("bob", "armadillo_small", "human_arap16"):
                src[0] = name
                cfg, mesh = dfea.load_named_config(name)
                omesh = ofea.TetMesh(mesh.V, mesh.tets, mesh.surface_vtx)
                _, solver, _ = ofea.make_gravity_solver(omesh, cfg)
                ofea.run_anm(solver)
                print(name, "steps", solver.get_nr_iter(), "polys", len(polys), flush=True)
    finally:
        opade.PadeApproximation.estimate_valid_range = orig_est
        up.solve_eqn = orig_solve

    out = {"variant": "g++ -O2 (oracle/build_ref.py VARIANTS['O2'])", "roots": [], "solve_eqn": []}
    for s, f in polys:
        r = ref.roots(f, True)
        ra = ref.roots(f, False)
        rn = ref_native.roots(f, True)
        out["roots"].append({"src": s, "f": [hx(v) for v in f], "valid": r is not None,
                             "real": None if r is None else [hx(z.real) for z in r],
                             "all": None if ra is None else [[hx(z.real), hx(z.imag)] for z in ra],
                             "native_valid": rn is not None})
    # a few well-conditioned ones so that the valid branch is pinned on more than the Pade denominators
    rng = np.random.RandomState(30)
    for k in range(12):
        deg = int(rng.randint(2, 20))
        f = list(np.poly(rng.uniform(-3, 3, deg))[::-1] * rng.uniform(0.1, 10))
        r, ra, rn = ref.roots(f, True), ref.roots(f, False), ref_native.roots(f, True)
        out["roots"].append({"src": "random_real_rooted_%d" % k, "f": [hx(v) for v in f], "valid": r is not None,
                             "real": None if r is None else [hx(z.real) for z in r],
                             "all": None if ra is None else [[hx(z.real), hx(z.imag)] for z in ra],
                             "native_valid": rn is not None})
    seen = set()
    for s, f, xmin, xmax, b, eps in eqns:
        key = (s, tuple(f))
        if key in seen:
            continue
        seen.add(key)
        out["solve_eqn"].append({"src": s, "f": [hx(v) for v in f], "xmin": hx(xmin), "xmax": hx(xmax), "b": hx(b),
                                 "eps": hx(eps), "x": hx(ref.solve_eqn(f, xmin, xmax, b, eps))})
    # tests/pade.cpp:16-62: q of degree 7 with q[7] = 2.3, times (x - 3)(x + 4)
    q = rng.uniform(-1, 1, 8)
    q[7] = 2.3
    f = np.convolve(q, [-12.0, 1.0, 1.0])
    ra = ref.roots(list(f), False)
    out["kat"] = {"f": [hx(v) for v in f], "all": [[hx(z.real), hx(z.imag)] for z in ra],
                  "real": [hx(z.real) for z in ref.roots(list(f), True)]}
    json.dump(out, open(os.path.join(HERE, "ref_poly.json"), "w"), indent=0)
    nv = sum(1 for r in out["roots"] if r["valid"])
    print("roots fixtures:", len(out["roots"]), "valid:", nv, "native-valid:",
          sum(1 for r in out["roots"] if r["native_valid"]), "solve_eqn:", len(out["solve_eqn"]))


if __name__ == "__main__":
    main(full="--small" not in sys.argv)
