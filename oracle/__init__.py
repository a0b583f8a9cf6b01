"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

CPU restatements of the reference's algorithms used as the parity checker.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; nothing
under `envidr_amd/` does, and the product path raises when its HIP library is missing instead
of falling back to anything here.
"""
