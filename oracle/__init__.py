"""CPU oracle for the SipMask hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (sipmask_amd) never does.  See oracle/ops.py header for
what is pinned -- against the reference's golden vectors, and against outputs of the reference's own Python code run
in the build container (tests/golden/ref_vectors.npz) -- and what remains "parity unpinned".
"""
