"""CPU oracle for the amortised-posterior flow path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of this path lives in third-party packages that
are neither under /root/reference nor installed here (sbi >= 0.22 -> pyknos/nflows
0.14-0.15, ltu-ili HEAD), and the reference's own tests
hold no golden vectors for it.  This package restates the *published* algorithms
of those packages and is pinned only by analytic
known-answer tests (tests/test_oracle_*.py) and by the reference's in-tree call
sites (file:line cited per function).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under synference_amd/ imports it.
"""
