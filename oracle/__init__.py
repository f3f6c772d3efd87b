"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot-path algorithms (plain C in
``pointops_ref.c`` / ``structural_ref.c``, torch-CPU in ``pdgnet_ref.py``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``pdgn_amd/`` does.
"""
