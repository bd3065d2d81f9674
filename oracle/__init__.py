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
  * ``oracle.problem.project_to_SOdn`` (nearest rotation) is PINNED: checked
    against the reference's own AVX2 kernels (project_to_SOd.cpp:7-33,
    97-196) built by oracle/ref_so3/Makefile from the sources where they lie;
    outputs committed as tests/golden/so_ref.npz.
  * Everything else that needs Eigen / CHOLMOD / glog / Boost to run
    (DPGOHash, DPGOProblem, DPGOStar, DChordal) is PARITY UNPINNED: the
    reference cannot be built in this image and ships no tests or golden
    vectors for this path.  It is pinned only by algebraic invariants
     and by the mathematical definition of each
    operator (exact SPD solves, explicit sparse products).
"""
