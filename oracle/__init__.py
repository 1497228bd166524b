"""Oracle = TEST INFRASTRUCTURE, not product.

A CPU restatement (PyTorch eager fp32 / NumPy) of the STTODE forward hot path,
pinned against golden vectors generated from the reference itself
(tests/golden/make_golden.py).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package.  Nothing under
``sttode_amd/`` imports it; the product path fails loudly when the HIP library
is missing.
"""
