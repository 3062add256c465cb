"""CPU oracle for the MDCV hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain-CPU restatement (torch fp32 + numpy, explicit Python
loops for the integer grid/anchor assignment) of the reference algorithms on
the CVC-YOLOv3 / RektNet training hot path.  It exists to CHECK the HIP
product path; it is never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under
``mit-driverless-cv-traininginfra_amd/`` imports it (a test enforces that).

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(imports /root/reference read-only) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every function here against them.
"""
