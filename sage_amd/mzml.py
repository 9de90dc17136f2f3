"""mzML input for the search-and-score path (host side).

`read_mzml` follows the reference's reader, crates/sage-cloudpath/src/mzml.rs:109-403, for the fields the path uses:
  * binary arrays: base64, optional zlib (MS:1000574), 32- or 64-bit floats (MS:1000521 / MS:1000523), 64-bit values
    narrowed to f32 element-wise (:318-326);
  * numeric cvParam values are parsed straight to the target type: f32 for m/z, intensities, times (:139-147);
  * selected ion m/z (MS:1000744, ignored when 0), charge (MS:1000041), intensity (MS:1000042); isolation window target as
    the fallback precursor m/z (MS:1000827, :221-229); isolation_window = Da(-lower, +upper) when both offsets are present
    (:354-357); a precursor is kept only if its m/z != 0 (:353);
  * scan start time in minutes (seconds are divided by 60 in f32, :262-272); inverse reduced ion mobility (MS:1002815);
  * a spectrum whose total ion current cvParam is 0 is dropped (:205-213); an ms-level filter drops other levels.
`write_mzml` is ours (the reference has no writer): centroid MS2 spectra with 32-bit zlib arrays like the reference's
test fixture (tests/LQSRPAAPPAPGPGQLTLR.mzML:117-126), used to feed synthetic workloads through the CLI.
"""
import base64
import struct
import zlib
import xml.etree.ElementTree as ET
from decimal import Decimal, InvalidOperation
from fractions import Fraction
from typing import List, Optional

import numpy as np

from .api import RawSpectrum

_F32 = np.float32


def _local(tag: str) -> str:
    return tag.rsplit("}", 1)[-1]


def _f32(text: str) -> float:
    """Rust's str::parse::<f32>(): the decimal string rounded ONCE to the nearest f32 (ties to even).  float(text) rounds
    to f64 first; the neighbours of that result are compared exactly so a second rounding can never pick the wrong f32."""
    if not text:
        return 0.0
    try:
        exact = Fraction(Decimal(text.strip()))
    except (InvalidOperation, ValueError):
        return float(_F32(float(text)))  # inf / nan spellings
    c = _F32(float(text))
    if not np.isfinite(c):
        return float(c)
    best, best_err = c, abs(Fraction(float(c)) - exact)
    for cand in (np.nextafter(c, _F32(-np.inf)), np.nextafter(c, _F32(np.inf))):
        if not np.isfinite(cand):
            continue
        err = abs(Fraction(float(cand)) - exact)
        if err < best_err or (err == best_err and (int(np.array(cand).view(np.uint32)) & 1) == 0):
            best, best_err = cand, err
    return float(best)


def read_mzml(path: str, file_id: int = 0, ms_level: Optional[int] = 2) -> List[RawSpectrum]:
    """MzMLReader::with_file_id_and_level_filter(file_id, ms_level).parse(..) for the MSn spectra of one file."""
    out: List[RawSpectrum] = []
    for _, el in ET.iterparse(path, events=("end",)):
        if _local(el.tag) != "spectrum":
            continue
        level, tic_zero = None, False
        for cv in el:
            if _local(cv.tag) != "cvParam":
                continue
            acc = cv.attrib.get("accession")
            if acc == "MS:1000511":
                level = int(cv.attrib["value"])
            elif acc == "MS:1000285":
                tic_zero = _f32(cv.attrib["value"]) == 0.0
        if tic_zero or (ms_level is not None and level != ms_level):
            el.clear()
            continue
        mz = np.zeros(0, _F32)
        inten = np.zeros(0, _F32)
        scan_start = 0.0
        prec_mz, prec_charge, prec_ims = 0.0, None, None
        iso_lo = iso_hi = None
        have_precursor = False
        for sub in el.iter():
            t = _local(sub.tag)
            if t == "scan":
                for cv in sub:
                    if _local(cv.tag) != "cvParam":
                        continue
                    acc = cv.attrib.get("accession")
                    if acc == "MS:1000016":
                        v = _F32(_f32(cv.attrib["value"]))
                        unit = cv.attrib.get("unitAccession")
                        if unit == "UO:0000010":
                            v = _F32(v / _F32(60.0))
                        elif unit != "UO:0000031":
                            raise ValueError("malformed mzML: scan start time unit")
                        scan_start = float(v)
                    elif acc == "MS:1002815":
                        prec_ims = _f32(cv.attrib["value"])
            elif t == "precursor" and not have_precursor:  # the path reads precursors.first()
                p_mz, p_z, p_lo, p_hi = 0.0, None, None, None
                for cv in sub.iter():
                    if _local(cv.tag) != "cvParam":
                        continue
                    acc, val = cv.attrib.get("accession"), cv.attrib.get("value", "")
                    if acc == "MS:1000827":
                        if p_mz == 0.0:
                            p_mz = _f32(val)
                    elif acc == "MS:1000828":
                        p_lo = _f32(val)
                    elif acc == "MS:1000829":
                        p_hi = _f32(val)
                    elif acc == "MS:1000041":
                        p_z = int(val)
                    elif acc == "MS:1000744":
                        v = _f32(val)
                        if v != 0.0:
                            p_mz = v
                    elif acc == "MS:1002815":
                        prec_ims = _f32(val)
                if p_mz != 0.0:
                    have_precursor = True
                    prec_mz, prec_charge = p_mz, p_z
                    iso_lo, iso_hi = p_lo, p_hi
            elif t == "binaryDataArray":
                accs = [cv.attrib.get("accession") for cv in sub if _local(cv.tag) == "cvParam"]
                text = next((b.text for b in sub if _local(b.tag) == "binary"), None) or ""
                kind = "mz" if "MS:1000514" in accs else "intensity" if "MS:1000515" in accs else None
                if not text or kind is None:
                    continue
                raw = base64.b64decode(text)
                if "MS:1000574" in accs:
                    raw = zlib.decompress(raw)
                if "MS:1000521" in accs:
                    arr = np.frombuffer(raw[:len(raw) // 4 * 4], dtype="<f4").astype(_F32)
                else:
                    arr = np.frombuffer(raw[:len(raw) // 8 * 8], dtype="<f8").astype(_F32)
                if kind == "mz":
                    mz = arr
                else:
                    inten = arr
        iso = (-iso_lo, iso_hi) if (iso_lo is not None and iso_hi is not None) else None
        out.append(RawSpectrum(mz, inten, prec_mz, prec_charge, iso, scan_start, prec_ims, file_id, el.attrib.get("id", "")))
        el.clear()
    return out


def read_mzml_native(path: str, file_id: int = 0, ms_level: Optional[int] = 2, check_searchable: bool = False):
    """The same reader in C++ (csrc/mzml_reader.cpp, sage_hip_mzml_read): one call per file, arrays straight into a RawBatch
    for Scorer.process_upload — no Python object per spectrum.  gzip-compressed files are inflated on the way in.
    check_searchable: refuse what the reference refuses to search (profile-mode MS2 spectra, MS2 spectra without precursor)."""
    import ctypes as C

    from . import _lib as L
    from .api import RawBatch
    lib = L.load()
    h = C.c_void_p()
    L.check(lib.sage_hip_mzml_read(path.encode(), file_id, -1 if ms_level is None else int(ms_level), C.byref(h)))
    try:
        if check_searchable:
            L.check(lib.sage_hip_mzml_check_searchable(h))
        v = L.SageRawBatch()
        L.check(lib.sage_hip_mzml_view(h, C.byref(v)))
        n = int(v.n_spectra)

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

        peak_off = arr(v.peak_off, n + 1, np.uint64)
        npk = int(peak_off[-1])
        ids = [lib.sage_hip_mzml_spectrum_id(h, i).decode() for i in range(n)]
        return RawBatch.from_arrays(ids, peak_off, arr(v.mz, npk, np.float32), arr(v.intensities, npk, np.float32),
                                    arr(v.precursor_mz, n, np.float32), arr(v.precursor_charge, n, np.uint8),
                                    arr(v.isolation_lo, n, np.float32), arr(v.isolation_hi, n, np.float32),
                                    arr(v.scan_start_time, n, np.float32), arr(v.inverse_ion_mobility, n, np.float32),
                                    arr(v.file_id, n, np.uint32))
    finally:
        lib.sage_hip_mzml_free(h)


def _b64(arr: np.ndarray) -> str:
    return base64.b64encode(zlib.compress(np.ascontiguousarray(arr, dtype="<f4").tobytes())).decode()


def write_mzml(path: str, spectra: List[RawSpectrum]) -> None:
    """Centroid MS2 spectra, 32-bit zlib arrays, selected ion m/z / charge, isolation offsets, scan start time (minutes)."""
    with open(path, "w") as f:
        f.write('<?xml version="1.0" encoding="utf-8"?>\n<mzML xmlns="http://psi.hupo.org/ms/mzml" version="1.1.0">\n')
        f.write(f'<run id="synthetic"><spectrumList count="{len(spectra)}">\n')
        for i, s in enumerate(spectra):
            sid = s.id or f"scan={i + 1}"
            f.write(f'<spectrum index="{i}" id="{sid}" defaultArrayLength="{len(s.mz)}">\n')
            f.write('<cvParam cvRef="MS" accession="MS:1000511" name="ms level" value="2"/>\n')
            f.write('<cvParam cvRef="MS" accession="MS:1000127" name="centroid spectrum"/>\n')
            f.write(f'<scanList count="1"><scan><cvParam cvRef="MS" accession="MS:1000016" name="scan start time" '
                    f'value="{np.format_float_positional(np.float32(s.scan_start_time), unique=True)}" unitCvRef="UO" '
                    f'unitAccession="UO:0000031" unitName="minute"/></scan></scanList>\n')
            f.write('<precursorList count="1"><precursor>')
            if s.isolation_window is not None:
                lo, hi = s.isolation_window
                f.write('<isolationWindow>'
                        f'<cvParam cvRef="MS" accession="MS:1000827" name="isolation window target m/z" value="{np.format_float_positional(np.float32(s.precursor_mz), unique=True)}"/>'
                        f'<cvParam cvRef="MS" accession="MS:1000828" name="isolation window lower offset" value="{np.format_float_positional(np.float32(-lo), unique=True)}"/>'
                        f'<cvParam cvRef="MS" accession="MS:1000829" name="isolation window upper offset" value="{np.format_float_positional(np.float32(hi), unique=True)}"/>'
                        '</isolationWindow>')
            f.write('<selectedIonList count="1"><selectedIon>'
                    f'<cvParam cvRef="MS" accession="MS:1000744" name="selected ion m/z" value="{np.format_float_positional(np.float32(s.precursor_mz), unique=True)}"/>')
            if s.precursor_charge:
                f.write(f'<cvParam cvRef="MS" accession="MS:1000041" name="charge state" value="{int(s.precursor_charge)}"/>')
            f.write('</selectedIon></selectedIonList></precursor></precursorList>\n')
            f.write('<binaryDataArrayList count="2">')
            for acc, name, arr in (("MS:1000514", "m/z array", s.mz), ("MS:1000515", "intensity array", s.intensity)):
                f.write('<binaryDataArray><cvParam cvRef="MS" accession="MS:1000521" name="32-bit float"/>'
                        '<cvParam cvRef="MS" accession="MS:1000574" name="zlib compression"/>'
                        f'<cvParam cvRef="MS" accession="{acc}" name="{name}"/><binary>{_b64(arr)}</binary></binaryDataArray>')
            f.write('</binaryDataArrayList>\n</spectrum>\n')
        f.write('</spectrumList></run></mzML>\n')
