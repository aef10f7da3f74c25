"""CPU oracle of the IDE-3D render path — TEST INFRASTRUCTURE ONLY.

Nothing under `oracle/` is imported by the product (`ide-3d_amd/`).  Allowed users: `tests/`,
`__graft_entry__.smoke()` (as the checker) and the `cpu_baseline` leg of `bench.py`.
See `oracle/ops.py` for what is restated and how it is pinned to the reference.
"""
