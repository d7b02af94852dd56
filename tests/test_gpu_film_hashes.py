"""Films as a regression pin: tools/film_hash.py renders 22 scenes (the example, the material zoo with both lens
shapes, media, four mesh scenes, the room, twelve random boxes) at 640x360 x 4 spp by both kernel routes (the tail
kernel alone; class sort + k_bounce, with k_traverse + k_bounce_resume where there are meshes) and hashes the three
film buffers.  tests/golden/film_hashes.txt holds the hashes of the build whose samples the parity tests compare with
the oracle; a kernel change that is meant to cost less without changing a result must reproduce them bit for bit.  A
change that is meant to change results (and passes the parity tests) regenerates the file:
python tools/film_hash.py > tests/golden/film_hashes.txt on a GPU."""
import io
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_films_match_the_pinned_hashes_by_both_kernel_routes():
    import pathtracing_amd as pt
    import film_hash
    out = io.StringIO()
    argv = sys.argv
    sys.argv = ["film_hash.py"]
    try:
        with redirect_stdout(out):
            film_hash.main()
    finally:
        sys.argv = argv
        pt.set_option("tail_records", 1048576)
    got = dict((line.split()[0], line.split()[1:]) for line in out.getvalue().splitlines() if line.strip())
    with open(os.path.join(ROOT, "tests", "golden", "film_hashes.txt")) as f:
        want = dict((line.split()[0], line.split()[1:]) for line in f if line.strip())
    assert set(got) == set(want)
    for name, hashes in want.items():
        assert got[name][0] == got[name][1], f"{name}: the two kernel routes render different films"
        assert got[name] == hashes, f"{name}: film differs from the pinned one"
