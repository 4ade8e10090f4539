"""CPU oracle for the incremental-episode hot path of feyzaakyurek/subspace-reg.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the reported CPU baseline.
The product path (``subspace-reg_amd/``) never imports this package and fails
loudly when its HIP library is missing.

The oracle is a NumPy restatement (fp32 by default, fp64 on request) of the
reference's PyTorch algorithm; every function cites the reference file:line it
follows.  Parity is PINNED: ``tools/make_golden.py`` imports the reference
itself (``/root/reference``, CPU, torch 2.10) in the build container and writes
the fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
oracle against every one of them.
"""
