"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatements of the mind_the_gaps log-likelihood hot path
(/root/reference/mind_the_gaps/gpmodelling.py:127-169 and the third-party
celerite>=0.4.2 solver it calls):

* ``oracle.dense``    -- dense-covariance definition (numpy float64 / mpmath)
* ``oracle.celerite`` -- ctypes view of oracle/celerite_ref.c (the celerite
  semiseparable recurrences in plain C)

PARITY STATUS: "parity unpinned" at the lnL boundary for values of kernels with J > 0 (see
the file headers).  What the reference's notebooks print from celerite is reproduced in
tests/test_notebook_known_answer.py: a white-kernel lnL (to 1.3e-8), and celerite's posterior
maxima on two light curves rebuilt from numpy's seeded generator (DRW, Lorentzian), which sit
1e-4 / 2e-2 below this oracle's likelihood maximum.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; nothing under mind_the_gaps_amd/ does.
"""
