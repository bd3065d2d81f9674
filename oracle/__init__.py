"""CPU oracle for the DPGO hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy/scipy restatement of the
reference's per-node MM / AMM inner step (``DPGOHash`` / ``DPGOProblem`` /
``DPGOStar::evaluate_*`` / ``TNT`` / ``STPCG``), written against the explicit
sparse matrices the reference assembles.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path (``dpgo_amd``) never does.

Pinning status ("Oracle"):
  * ``oracle.tnt`` (STPCG / TNT) is PINNED: checked against the reference's own
    header-only solver compiled from /root/reference (``oracle/_ref/tnt_ref``)
    and against the known-answer vectors of the reference's unit tests
    (C++/Optimization/tests/IterativeSolvers_unit_test.cpp:79-247,
    TNT_unit_test.cpp:63-187) committed under tests/golden/.
  * Everything that needs Eigen / CHOLMOD / glog / Boost (DPGOHash,
    DPGOProblem, DPGOStar, the SIMD SO(d) projection) is PARITY UNPINNED: the
    reference cannot be built in this image and ships no tests or golden
    vectors for this path.  It is pinned only by algebraic invariants
     and by the mathematical definition of each
    operator (nearest rotation = polar factor, exact SPD solves).
"""
