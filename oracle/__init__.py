"""ORACLE — test infrastructure only.

CPU restatements of the reference's algorithms for the hot path
(SURVEY.md §8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import anything from here; the product
package ``partdistillation_amd`` never does (tests/test_product_cpu.py::test_product_never_imports_oracle
enforces it).
"""
