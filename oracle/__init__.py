"""CPU oracle for the FG-DM sampling hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU fp32 restatement of the reference
algorithm for the hot path named by BASELINE.json's north_star (DDIM / PLMS /
ancestral loop over the SD-v1.x UNet + FG-DM adapter + ControlNet + CFG).
It exists so that the HIP engine in ``fgdm_amd/`` can be checked; it is NOT a
product path and nothing in ``fgdm_amd/`` imports it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py,
``-m "not gpu"``) against golden vectors in ``tests/golden/*.npz`` that were
produced by importing the reference's own modules on CPU with the committed
script ``tools/make_goldens.py`` (the reference itself has no tests / KATs;
see SURVEY.md section 8c).

Each function cites the reference file:line it restates (paths relative to
the reference checkout).
"""
