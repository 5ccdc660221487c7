"""CPU oracle for the GANQ hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``ganq_amd``) never does and has no CPU fallback.

``oracle.c_oracle``  ctypes/numpy binding of ``ganq_oracle.c`` (canonical-order C restatement).
``oracle.ganq_ref``  torch restatement that keeps the reference's own op sequence
                     (per-column gather + gemv, lstsq/gelsd); this is what bench.py times as
                     the CPU baseline.
"""
