"""CPU oracle for the clair-torch hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker.  The product (``clair_torch_amd``) never imports it.
"""
