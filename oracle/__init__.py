"""CPU oracle for the caption-decoding hot path.  TEST INFRASTRUCTURE ONLY.

A restatement (our own code, torch-CPU fp32 / numpy float64) of the reference algorithms on the hot
path named by BASELINE.json:north_star.  Every function cites the reference file:line it follows
(paths relative to the reference checkout).  Parity status: PINNED -- tests/test_oracle_golden.py checks
every function here against vectors produced by running the reference itself
(tests/golden/make_goldens.py, fixtures under tests/golden/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product path (simpleimagecaptionzoo_amd/) never does and fails loudly without its HIP library.
"""
