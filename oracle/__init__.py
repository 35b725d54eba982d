"""CPU oracle for the EvoDiff ByteNetLMTime sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only
as the checker.  The product path (``evodiff_amd``) never imports this package and fails loudly
when its HIP library is missing.
"""
