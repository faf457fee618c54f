"""CPU ORACLE — test infrastructure, NOT product code.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.
`multike_amd/` (the product) never imports it and has no CPU fallback.
"""
