"""CPU oracle for the MI355X hot path (TEST INFRASTRUCTURE, not product code).

Everything under ``oracle/`` is a plain-PyTorch fp32 CPU restatement of the
reference's arithmetic for the hot path named in BASELINE.json.  It exists only
to *check* the HIP path:

  * ``tests/``                      compare HIP results against it,
  * ``__graft_entry__.smoke()``     checks one small invocation against it,
  * ``bench.py``'s ``cpu_baseline`` times it on the host cores.

Nothing in ``lightning-generative-models_amd/`` may import it.  The oracle is
pinned against the real reference by ``oracle/make_golden.py`` (run in the build
container where ``/root/reference`` exists) which stores fixtures in
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the oracle against
those fixtures on every run.
"""
