"""CPU oracle of the BESS hot path - TEST INFRASTRUCTURE, not product code.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the checker / timed baseline.
See oracle/kge.py for the restated reference lines and how parity is pinned.
"""
