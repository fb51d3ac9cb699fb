"""CPU oracle for the NeuroViT hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``neurovit_amd``) never does: it
fails loudly when its HIP extension is missing.

Parity pin: the reference ships no tests / golden vectors, so the
oracle is pinned by fixtures generated in the build container by importing the
reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
"""
