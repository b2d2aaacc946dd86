"""CPU oracle for the MICA voxel-grid hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``mica_amd``) never does.  Every function cites the
reference file:line it restates.  Pinned against the reference itself by
``oracle/gen_golden.py`` (run in the build container where /root/reference exists);
the resulting vectors live in ``tests/golden/``.
"""
