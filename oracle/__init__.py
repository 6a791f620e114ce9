"""CPU oracle for the SANM ANM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference algorithm
(jia-kai/SANM, ``libsanm/`` + the parts of ``fea/`` that define the hot-path
inputs).  It exists to *check* the HIP product path in ``sanm_amd/``:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it;
* nothing under ``sanm_amd/`` imports it, and the product path fails loudly
  when the HIP library is missing instead of falling back to this code.

Parity pinning (see DESIGN.md "Oracle"): the reference cannot be built in the
authoring container (Eigen and MKL headers are absent), and its own tests are
invariant based, not golden-vector based.  The oracle is pinned against

* the known-answer values the reference holds (``utils/check_single_tet.py``
  rest height, ``tests/pade.cpp`` polynomial roots),
* vectors produced by importing the reference's own Python utilities
  (``utils/test_cofactor.py``, ``utils/test_svdw_grad.py``,
  ``utils/check_single_tet.py``) -- committed under ``tests/golden/`` with the
  generating script,
* the invariants the reference's Catch2 tests assert (``tests/tensor.cpp``,
  ``tests/symbolic.cpp``), restated in ``tests/test_oracle_*.py``.

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
