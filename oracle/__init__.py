"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement (pure PyTorch fp32/fp64 + one plain-C file) of the Any-Stereo
hot path, used ONLY as the checker:

  * ``tests/``                       parity tests (HIP path vs. this oracle, oracle vs. golden vectors)
  * ``__graft_entry__.smoke()``      one tiny invocation checked against the oracle
  * ``bench.py`` ``cpu_baseline``    the oracle timed on the host cores (reported, never shipped)

Nothing under ``any-stereo_amd/`` imports, links or executes anything from this
package; the product path raises if the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors, so the
oracle is pinned against vectors generated in the build container by importing the
reference itself (``tests/golden/make_golden.py``; outputs committed as
``tests/golden/*.npz``) and checked by ``tests/test_oracle_golden.py``.
"""
