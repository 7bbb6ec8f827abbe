"""The C-ABI library loads and exports every symbol include/imatch.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(imatch_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "image_matching_amd", "libimatch.so"))
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_header():
    from image_matching_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load_library()


def test_no_device_is_loud():
    """On a box without a GPU the product must raise, not fall back to a CPU path."""
    import image_matching_amd as im
    lib = im.load_library()
    if lib.imatch_device_count() > 0:
        return
    try:
        im.Context(0)
    except im.ImatchError:
        return
    raise AssertionError("Context() succeeded without a GPU")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "image_matching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
