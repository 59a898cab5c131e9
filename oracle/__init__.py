"""CPU oracle for the MI355X StyleGAN2 inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``maua_stylegan2_amd/`` may import this package; the only
permitted callers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``, where it is the checker / the timed CPU baseline and never the product path.

Every function restates (in its own words, fp32 on the host) the algorithm of the reference file:line
cited in its docstring.  Parity is pinned: ``tests/golden/make_golden.py`` imports the reference in the
build container, checks these restatements against it (<= 1e-5 abs) and writes the fixtures that
``tests/test_oracle_golden.py`` re-checks everywhere, including on the GPU box where the reference is
absent.  Exception, stated in DESIGN.md: the librosa/madmom/kornia stages are third-party code that is
not vendored in the reference (requirements.txt:1-10) and not installed here -> "parity unpinned" for
those (oracle/signal_oracle.py header).
"""
