"""CPU oracle for the ADMM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy fp64 restatement of the reference's algorithm
(PeterSutor/ADMM-Project: admm.m:496-743 and the getProxOps.m closures).  It is
the checker the HIP engine is compared against; it is never the thing shipped or
measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package (``admm-project_amd/``)
never imports anything from here and fails loudly when its HIP library is missing.

PARITY PIN STATUS: the reference stores no golden vectors or known-answer tests
(testers/*.m are randomised property tests driven by MATLAB's RNG), and neither
MATLAB nor Octave exists in the build container, so the reference cannot be run.
Per-iteration values (x, z, u, residuals) are therefore **parity unpinned** by
stored reference outputs.  What pins this oracle is the reference's own pass
criteria (lassotest.m:143, ladtest.m:149, huberfittest.m:154,
totalvariationtest.m:151, linearsvmtest.m:180, modeltest.m:122/156 closed form,
basispursuittest.m:139, linearprogramtest.m:134), checked in
``tests/test_oracle_pins.py``.
"""
from .admm_ref import admm  # noqa: F401
from .proxops_ref import getproxops  # noqa: F401
