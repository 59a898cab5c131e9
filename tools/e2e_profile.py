#!/usr/bin/env python3
"""cProfile of a WARM generate() (BASELINE config 3 shape, counting sink): where the host spends the 0.3-0.4 s front end.
    python tools/e2e_profile.py            (prints the top entries by own time and by cumulative time)"""
import cProfile
import io
import os
import pstats
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_config3.py"), "--repeat", "2"]
import maua_stylegan2_amd.generate_audiovisual as gav  # noqa: E402

real = gav.generate
calls = {"n": 0}
prof = cProfile.Profile()


def wrapped(*a, **k):
    calls["n"] += 1
    if calls["n"] == 2:  # the warm run
        prof.enable()
        try:
            return real(*a, **k)
        finally:
            prof.disable()
    return real(*a, **k)


gav.generate = wrapped
runpy.run_path(sys.argv[0], run_name="__main__")
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(line[:170] for line in s.getvalue().splitlines()[4:]))
