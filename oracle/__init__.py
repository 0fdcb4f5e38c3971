"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker -- never as the thing measured as the GPU
number, never as a fallback of the HIP path.

Parity status: PINNED.  Every function here is checked against golden vectors
emitted by importing the reference itself (``tests/golden/make_golden.py``,
run in the build container where ``/root/reference`` exists) -- see
``tests/test_oracle_golden.py``.
"""
