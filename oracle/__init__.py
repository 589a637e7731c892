"""CPU oracle for the ICRL rollout+update hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / torch-CPU restatement of the arithmetic of the
reference (shehryar-malik/icrl, files cited per function as ``ref: path:line``
relative to /root/reference).  It exists to *check* the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``icrl_amd/`` imports it, and the product path raises when the
    HIP extension is missing instead of falling back here.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against the reference ITSELF, imported
in the build container under ``oracle/ref_shim`` by ``oracle/gen_golden.py``;
the resulting vectors are committed under ``tests/golden/`` and replayed by
``tests/test_oracle_golden.py`` (torch / numpy versions recorded in each file).
"""
