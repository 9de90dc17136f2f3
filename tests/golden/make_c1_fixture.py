#!/usr/bin/env python3
"""Generate tests/golden/c1_fixture.json from the reference's own test fixtures.

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_c1_fixture.py

The reference's only end-to-end known-answer test on Scorer::score
(crates/sage-cli/tests/integration.rs:7-52) reads tests/LQSRPAAPPAPGPGQLTLR.mzML and
tests/Q99536.fasta.  /root/reference does not exist on the GPU box, so the *decoded* inputs
(f32 m/z + intensity arrays, precursor cvParams as the decimal strings the reference parses,
the protein record) are committed here as a small derived fixture.  Decoding follows
crates/sage-cloudpath/src/mzml.rs:109-403 (zlib + base64, f64 arrays narrowed to f32 at
:318-326, selected-ion m/z parsed straight to f32 at :244-248, isolation window
Da(-lower, +upper) at :354-357, scan start time in minutes at :262-272).
"""
import base64, json, os, struct, sys, zlib
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("SAGE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def local(tag):
    return tag.rsplit("}", 1)[-1]


def main():
    mzml = os.path.join(REF, "tests", "LQSRPAAPPAPGPGQLTLR.mzML")
    fasta = os.path.join(REF, "tests", "Q99536.fasta")
    spectra = []
    for _, el in ET.iterparse(mzml, events=("end",)):
        if local(el.tag) != "spectrum":
            continue
        spec = {"id": el.attrib["id"], "cv": {}}
        for cv in el:
            if local(cv.tag) == "cvParam":
                spec["cv"][cv.attrib["accession"]] = cv.attrib.get("value", "")
        for sub in el.iter():
            t = local(sub.tag)
            if t == "scan":
                for cv in sub:
                    if local(cv.tag) == "cvParam" and cv.attrib["accession"] == "MS:1000016":
                        spec["scan_start_time"] = cv.attrib["value"]
                        spec["scan_start_unit"] = cv.attrib["unitAccession"]
            elif t == "isolationWindow":
                for cv in sub:
                    if local(cv.tag) == "cvParam":
                        spec.setdefault("isolation", {})[cv.attrib["accession"]] = cv.attrib["value"]
            elif t == "selectedIon":
                for cv in sub:
                    if local(cv.tag) == "cvParam":
                        spec.setdefault("selected_ion", {})[cv.attrib["accession"]] = cv.attrib.get("value", "")
            elif t == "binaryDataArray":
                acc = [cv.attrib["accession"] for cv in sub if local(cv.tag) == "cvParam"]
                raw = base64.b64decode(next(b for b in sub if local(b.tag) == "binary").text or "")
                if "MS:1000574" in acc:
                    raw = zlib.decompress(raw)
                if "MS:1000523" in acc:
                    arr = np.frombuffer(raw, dtype="<f8").astype(np.float32)
                else:
                    arr = np.frombuffer(raw, dtype="<f4")
                kind = "mz" if "MS:1000514" in acc else ("intensity" if "MS:1000515" in acc else None)
                if kind:
                    spec[kind + "_f32_b64"] = base64.b64encode(arr.astype("<f4").tobytes()).decode()
                    spec["n_" + kind] = int(arr.size)
        spectra.append(spec)
    out = {
        "source": "lazear/sage tests/LQSRPAAPPAPGPGQLTLR.mzML + tests/Q99536.fasta (decoded)",
        "known_answer": {
            "test": "crates/sage-cli/tests/integration.rs:7-52",
            "n_psm": 1,
            "matched_peaks": 21,
        },
        "fasta": open(fasta).read(),
        "config_json": json.load(open(os.path.join(REF, "tests", "config.json"))),
        "spectra": spectra,
    }
    dst = os.path.join(HERE, "c1_fixture.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, "spectra:", len(spectra), "peaks:", spectra[0]["n_mz"])


if __name__ == "__main__":
    main()
