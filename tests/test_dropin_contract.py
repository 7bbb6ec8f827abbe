"""The drop-in import contract: every name the reference's nodes import from the four flat modules the
drop-in replaces -- and every match_finder method they call -- resolves with dropin/ first on sys.path, the way
INTEGRATION.md section A installs it. The expectations are data (tests/golden/import_contract.json, made from the text of
the reference's files by tests/golden/make_import_contract.py); nothing of the reference is needed at run time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = json.load(open(os.path.join(ROOT, "tests", "golden", "import_contract.json")))

_PROBE = r"""
import importlib, json, sys
sys.path.insert(0, sys.argv[1])
contract = json.loads(sys.argv[2])
missing = []
for mod, names in contract["imports"].items():
    m = importlib.import_module(mod)
    assert m.__file__.startswith(sys.argv[1]), (mod, m.__file__)
    missing += ["%s.%s" % (mod, n) for n in names if not hasattr(m, n)]
mf = importlib.import_module("match_finder").match_finder
missing += ["match_finder.match_finder.%s" % n for n in contract["matcher_methods"] if not callable(getattr(mf, n, None))]
# the statements the review named, verbatim
from utils import resize_img, draw_circle_on_map_by_coord_and_angles     # position_finder.py:17
from utils import check_exist                                            # image_logger.py:29
from utils import resize_img, rotate_image, line_intersection, isConvex  # match_finder.py:13
print(json.dumps(missing))
"""


def test_every_imported_name_resolves_from_dropin():
    out = subprocess.run([sys.executable, "-c", _PROBE, os.path.join(ROOT, "dropin"), json.dumps(CONTRACT)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    missing = json.loads(out.stdout.strip().splitlines()[-1])
    assert missing == [], "names the reference's nodes use that dropin/ does not provide: %s" % missing


def test_contract_covers_the_files_the_review_named():
    by_file = CONTRACT["by_file"]
    pf = by_file["image_processing/src/image_processing/position_finder.py"]
    assert "draw_circle_on_map_by_coord_and_angles" in pf["utils"] and "match_finder" in pf["match_finder"]
    assert "check_exist" in by_file["image_processing/src/image_processing/image_logger.py"]["utils"]
    assert "rotate_image" in by_file["image_processing/src/image_processing/match_finder.py"]["utils"]


def test_check_exist_keeps_the_reference_quirk(tmp_path, capsys):
    from image_matching_amd.utils import check_exist
    free = str(tmp_path / "free.csv")
    assert check_exist(free) == free                      # utils.py:13-14
    taken = tmp_path / "taken.csv"
    taken.write_text("x")
    os.chdir(tmp_path)
    assert check_exist("taken.csv") is None               # utils.py:10-12: the recursive call's result is dropped
    assert "1_taken.csv" in capsys.readouterr().out       # ... but the probe of the prefixed name was made


def test_rotate_image_bounds_and_quarter_turn():
    import numpy as np
    from image_matching_amd.utils import rotate_image
    a = np.arange(6 * 9, dtype=np.uint8).reshape(6, 9)
    r = rotate_image(a, 90)                                # bounding box of the turned image (utils.py:35-36)
    assert r.shape == (9, 6)
    assert rotate_image(a, 0).shape == a.shape and np.array_equal(rotate_image(a, 0), a)
    c = np.zeros((40, 60, 3), np.uint8)
    c[10:30, 20:40] = (10, 200, 90)
    t = rotate_image(c, 30)
    assert t.shape == (int(40 * np.cos(np.pi / 6) + 60 * 0.5), int(40 * 0.5 + 60 * np.cos(np.pi / 6)), 3)
    assert abs(int(t[..., 1].astype(np.int64).sum()) - int(c[..., 1].astype(np.int64).sum())) < 0.02 * c[..., 1].sum()
