"""CPU oracle for the FIND hot path -- TEST INFRASTRUCTURE ONLY.

Everything under oracle/ is a CPU restatement of the reference algorithm used as the *checker*:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
(find_amd/) never imports, links or executes anything from here and has no CPU fallback.

Pinning status:
  * mlp_ref.py rows a1-a4, a6: PINNED -- checked against golden vectors produced by importing the
    reference's own code (tests/golden/make_golden_mlp.py, tests/test_oracle_mlp.py).
  * rows a5, a7-a14, a16 (registration, cameras, rasteriser, shaders, sampling, Chamfer, smoothness):
    PARITY UNPINNED -- the arithmetic lives in PyTorch3D @ 1706eb8216248e54f68cad86f7ea4125c79a3ca4
    (requirements_mac_linux.txt:31), which is neither vendored under /root/reference nor installable
    here.  Those oracles restate PyTorch3D's published algorithms and are anchored on the reference's
    call sites plus analytic known-answer tests (tests/test_oracle_geom.py, tests/test_oracle_raster.py).
"""
