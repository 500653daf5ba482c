"""ORACLE — CPU restatement of the reference's algorithms for the hot path.

Test infrastructure only: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package; nothing under `sparse2dense_amd/` may.
Every function cites the reference file:line it follows.
"""
