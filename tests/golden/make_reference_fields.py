"""Parse the `pub` fields of the sage-core structs INTEGRATION.md's Rust shim touches out of the mounted reference
checkout and write them to tests/golden/reference_fields.json (tests/test_integration_doc.py holds the shim to that
list, and — when the checkout is present — the list to the checkout).

    python tests/golden/make_reference_fields.py [/root/reference]
"""
import json
import os
import re
import sys

STRUCTS = {  # struct -> file under crates/sage/src
    "Scorer": "scoring.rs", "Feature": "scoring.rs", "ProcessedSpectrum": "spectrum.rs", "Precursor": "spectrum.rs",
    "Peptide": "peptide.rs", "IndexedDatabase": "database.rs", "Theoretical": "database.rs",
}
ENUMS = {"Tolerance": "mass.rs", "ScoreType": "scoring.rs", "Kind": "ion_series.rs"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def body_of(text, keyword, name):
    m = re.search(r"\bpub\s+%s\s+%s\b[^{;]*\{" % (keyword, re.escape(name)), text)
    if not m:
        raise KeyError(f"{keyword} {name} not found")
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def struct_fields(text, name):
    """[(field, type)] of the struct's `pub` fields, in declaration order."""
    body = strip_comments(body_of(text, "struct", name))
    body = re.sub(r"#\[[^\]]*\]", "", body)
    out, depth, cur = [], 0, ""
    for ch in body:  # split on top-level commas (types hold `<A, B>` and `(A, B)`)
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    fields = []
    for item in out:
        m = re.match(r"\s*pub\s+(\w+)\s*:\s*(.+?)\s*$", item, flags=re.S)
        if m:
            fields.append([m.group(1), re.sub(r"\s+", " ", m.group(2))])
    return fields


def enum_variants(text, name):
    body = strip_comments(body_of(text, "enum", name))
    body = re.sub(r"#\[[^\]]*\]", "", body)
    return re.findall(r"^\s*(\w+)\s*(?:\([^)]*\))?\s*,", body, flags=re.M)


def parse(root):
    src = os.path.join(root, "crates", "sage", "src")
    out = {"structs": {}, "enums": {}}
    for name, f in STRUCTS.items():
        out["structs"][name] = struct_fields(open(os.path.join(src, f)).read(), name)
    for name, f in ENUMS.items():
        out["enums"][name] = enum_variants(open(os.path.join(src, f)).read(), name)
    return out


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_fields.json"), "w") as fh:
        json.dump(parse(root), fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote reference_fields.json")
