"""Writes tests/golden/import_contract.json: what the reference's nodes expect of the four modules the drop-in replaces.

Reads the TEXT of the reference's node files (ast.parse: nothing is imported or executed) and records, as data only,
  imports   module -> names that some node imports from it (`from utils import resize_img, ...`)
  methods   names read as attributes of a match_finder instance (`self.matcher.X`, `match_finder.X` inside
            image_processing.find_kp_dp_scale)
  fields    names read or set on image_processing / roi objects by position_finder.py and mezhcadr.py
  by_file   the same imports per importing file, for the failure message
Run in the build container (needs /root/reference): python tests/golden/make_import_contract.py"""
import ast
import glob
import json
import os

REF = "/root/reference"
MODULES = ("utils", "match_finder", "image_processing", "geodetic_conv")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    imports = {m: set() for m in MODULES}
    by_file = {}
    methods = set()
    files = sorted(glob.glob(os.path.join(REF, "image_processing/src/image_processing/*.py")) +
                   glob.glob(os.path.join(REF, "csv_data_pkg/src/**/*.py"), recursive=True))
    for path in files:
        try:
            tree = ast.parse(open(path, encoding="utf-8", errors="replace").read())
        except SyntaxError:
            continue
        rel = os.path.relpath(path, REF)
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module in MODULES and node.level == 0:
                names = [a.name for a in node.names]
                imports[node.module].update(names)
                by_file.setdefault(rel, {}).setdefault(node.module, []).extend(names)
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name in MODULES:
                        by_file.setdefault(rel, {}).setdefault(a.name, [])
            # self.matcher.X(...) in the two callers; match_finder.X(...) in image_processing.find_kp_dp_scale
            if isinstance(node, ast.Attribute):
                v = node.value
                if isinstance(v, ast.Attribute) and v.attr == "matcher":
                    methods.add(node.attr)
                elif isinstance(v, ast.Name) and v.id == "match_finder" and rel.endswith("image_processing.py"):
                    methods.add(node.attr)
    out = {"generated_from": "text of %d reference files (ast.parse, not imported)" % len(files),
           "imports": {m: sorted(v) for m, v in imports.items()},
           "matcher_methods": sorted(methods),
           "by_file": {k: {m: sorted(set(n)) for m, n in v.items()} for k, v in sorted(by_file.items())}}
    with open(os.path.join(HERE, "import_contract.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(out["imports"], indent=1))
    print("matcher methods:", out["matcher_methods"])


if __name__ == "__main__":
    main()
