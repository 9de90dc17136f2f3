"""mzML reader / writer, results.sage.tsv formatting and the JSON-config CLI.
CPU tests cover the host logic; the end-to-end CLI runs are GPU tests (the search has no CPU fallback)."""
import json
import os

import numpy as np
import pytest

import oracle_lib
from sage_amd import cli, output
from sage_amd.api import DatabaseParameters, RawSpectrum, SpectrumBatch, SpectrumProcessor
from sage_amd.mzml import _f32, read_mzml, write_mzml
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
from test_oracle_golden import load_c1

REF_MZML = "/root/reference/tests/LQSRPAAPPAPGPGQLTLR.mzML"


def c1_raw():
    d, s, mz, it = load_c1()
    iso = (-float(np.float32(float(s["isolation"]["MS:1000828"]))), float(np.float32(float(s["isolation"]["MS:1000829"]))))
    return d, RawSpectrum(mz, it, _f32(s["selected_ion"]["MS:1000744"]), int(s["selected_ion"]["MS:1000041"]), iso,
                          _f32(s["scan_start_time"]), None, 0, s["id"])


def test_f32_parse_is_a_single_rounding():
    assert _f32("643.0344") == float(np.float32(643.0344))
    assert _f32("0.1") == float(np.float32(0.1))
    # 16777217 is exactly between two f32s: ties to even
    assert _f32("16777217") == 16777216.0 and _f32("16777219") == 16777220.0
    # a decimal just above a tie: float() rounds it to the tie in f64 and a second rounding would go DOWN
    assert _f32("16777217.00000000001") == 16777218.0


def test_mzml_round_trip_and_reader_semantics(tmp_path):
    d, raw = c1_raw()
    p = str(tmp_path / "c1.mzML")
    write_mzml(p, [raw])
    back = read_mzml(p, file_id=3)
    assert len(back) == 1
    b = back[0]
    np.testing.assert_array_equal(b.mz, raw.mz)
    np.testing.assert_array_equal(b.intensity, raw.intensity)
    assert np.float32(b.precursor_mz) == np.float32(raw.precursor_mz) and b.precursor_charge == 3
    assert b.isolation_window == raw.isolation_window and b.file_id == 3 and b.id == raw.id
    assert np.float32(b.scan_start_time) == np.float32(raw.scan_start_time)


@pytest.mark.skipif(not os.path.exists(REF_MZML), reason="reference checkout not mounted (GPU box)")
def test_reader_on_the_reference_fixture_matches_the_committed_decode():
    """The reference's own test file, read by read_mzml, equals tests/golden/c1_fixture.json (decoded independently
    by tests/golden/make_c1_fixture.py)."""
    d, raw = c1_raw()
    got = read_mzml(REF_MZML)[0]
    np.testing.assert_array_equal(got.mz, raw.mz)
    np.testing.assert_array_equal(got.intensity, raw.intensity)
    assert got.precursor_mz == raw.precursor_mz and got.precursor_charge == raw.precursor_charge
    assert got.isolation_window == raw.isolation_window and got.id == raw.id
    assert np.float32(got.scan_start_time) == np.float32(raw.scan_start_time)


def test_ryu_number_formatting():
    f32 = [(1.0, "1.0"), (0.1, "0.1"), (1e-5, "0.00001"), (1e-6, "0.000001"), (1e-7, "1e-7"), (1234.5678, "1234.5677"),
           (1e12, "1000000000000.0"), (1e13, "1e13"), (3.4028235e38, "3.4028235e38"), (-2.5, "-2.5"), (0.999, "0.999"),
           (float("inf"), "inf"), (float("nan"), "NaN"), (0.0, "0.0")]
    for v, e in f32:
        assert output.ryu_f32(v) == e, (v, output.ryu_f32(v), e)
    f64 = [(1e16, "1e16"), (1e15, "1000000000000000.0"), (0.3, "0.3"), (1e-5, "0.00001"), (1e-6, "1e-6"),
           (123456.789, "123456.789"), (1.7976931348623157e308, "1.7976931348623157e308"), (5e-324, "5e-324"),
           (float("-inf"), "-inf"), (42.125, "42.125")]
    for v, e in f64:
        assert output.ryu_f64(v) == e, (v, output.ryu_f64(v), e)
    # every formatted value parses back to the same float (shortest round-trip digits)
    rng = np.random.default_rng(7)
    for x in rng.lognormal(0, 8, 200):
        assert np.float32(float(output.ryu_f32(np.float32(x)))) == np.float32(x)
        assert float(output.ryu_f64(x)) == x


def test_search_parameter_defaults():
    """input.rs:355-385"""
    sp = cli.search_parameters({"precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"da": [-0.02, 0.02]}})
    assert (sp["report_psms"], sp["max_peaks"], sp["min_peaks"], sp["min_matched_peaks"]) == (1, 150, 15, 4)
    assert sp["precursor_charge"] == (2, 4) and sp["isotope_errors"] == (0, 0) and sp["deisotope"] is True
    assert not sp["chimera"] and not sp["wide_window"] and sp["max_fragment_charge"] is None
    assert sp["fragment_tol"].kind == "da" and sp["score_type"] == "SageHyperScore"
    with pytest.raises(SystemExit):
        cli.search_parameters({"precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"ppm": [-10, 10]}, "precursor_charge": [4, 2]})


def _read_tsv(path):
    lines = open(path).read().splitlines()
    return lines[0].split("\t"), [l.split("\t") for l in lines[1:]]


@pytest.mark.gpu
def test_cli_on_the_reference_config(tmp_path, gpu_required):
    """`sage tests/config.json` (BASELINE.json configs[0]) end to end through the GPU path: one PSM, LQSRPAAPPAPGPGQLTLR,
    and a results.sage.tsv whose columns / formatting follow runner.rs:687-935.  (The CLI keeps max_peaks = 150 peaks,
    input.rs:366, where the reference's integration test keeps 100 and counts 21 matched peaks; with 150 it is 22.)"""
    d, raw = c1_raw()
    mz = str(tmp_path / "LQSRPAAPPAPGPGQLTLR.mzML")
    fa = str(tmp_path / "Q99536.fasta")
    write_mzml(mz, [raw])
    open(fa, "w").write(d["fasta"])
    cfg = dict(d["config_json"])
    cfg["database"] = dict(cfg["database"], fasta=fa)
    cfg["mzml_paths"] = [mz]
    cfg["annotate_matches"] = True
    cp = str(tmp_path / "config.json")
    json.dump(cfg, open(cp, "w"))
    out = str(tmp_path / "out")
    cli.main([cp, "-o", out, "--write-pin"])
    hdr, rows = _read_tsv(os.path.join(out, "results.sage.tsv"))
    assert hdr == output.HEADERS and len(rows) == 1
    r = dict(zip(hdr, rows[0]))
    assert r["psm_id"] == "1" and r["peptide"] == "LQSRPAAPPAPGPGQLTLR" and r["proteins"] == "sp|Q99536|VAT1_HUMAN"
    assert r["charge"] == "3" and r["rank"] == "1" and r["label"] == "1"
    assert r["filename"] == "LQSRPAAPPAPGPGQLTLR.mzML" and r["scannr"] == raw.id
    # one target, no decoy: the linear model cannot be fitted (linear_discriminant.rs:83-85) -> heuristic discriminant of
    # runner.rs:285-288, posterior_error stays 1.0, q = (1 decoy pseudo-count) / (1 target) = 1.0
    assert r["spectrum_q"] == "1.0" and r["peptide_q"] == "1.0" and r["protein_q"] == "1.0" and r["posterior_error"] == "1.0"
    assert r["delta_rt_model"] == "0.999" and r["protein_group_q"] == "1.0"
    # the numbers are the oracle's, formatted by ryu
    db = oracle_lib.OracleDb.build(d["fasta"], DatabaseParameters.from_json(d["config_json"]["database"]))
    sp = SpectrumProcessor(150, True, 0.0)
    batch = SpectrumBatch.from_spectra([sp.process(raw)])
    of, oc, _, _ = db.score(cli.scorer_params(cli.search_parameters(cfg)), batch)
    assert oc[0] == 1 and r["matched_peaks"] == str(int(of[0, 0]["matched_peaks"])) == "22"
    assert r["hyperscore"] == output.ryu_f64(of[0, 0]["hyperscore"]) and r["expmass"] == output.ryu_f32(of[0, 0]["expmass"])
    assert r["precursor_ppm"] == output.ryu_f32(of[0, 0]["delta_mass"]) and r["poisson"] == output.ryu_f64(of[0, 0]["poisson"])
    heur = np.log1p(np.float32(-of[0, 0]["poisson"])) + of[0, 0]["longest_y_pct"] / np.float32(3.0)
    assert abs(float(r["sage_discriminant_score"]) - float(heur)) <= 1e-5 * abs(float(heur))
    fh, frows = _read_tsv(os.path.join(out, "matched_fragments.sage.tsv"))
    assert fh == output.FRAGMENT_HEADERS and len(frows) == 22 and all(x[0] == "1" for x in frows)
    assert {x[1] for x in frows} <= {"b", "y"}
    # results.sage.pin (runner.rs:938-1135): log-transformed columns, the scan number picked out of the spectrum id
    ph, prows = _read_tsv(os.path.join(out, "results.sage.pin"))
    assert ph == output.PIN_HEADERS and len(prows) == 1
    p = dict(zip(ph, prows[0]))
    assert p["SpecId"] == "1" and p["Label"] == "1" and p["Peptide"] == "LQSRPAAPPAPGPGQLTLR" and p["z=3"] == "1" and p["z=2"] == "0"
    assert p["ScanNr"] == (raw.id.split("scan=")[-1] if "scan=" in raw.id else raw.id)
    assert p["ln(hyperscore)"] == output.ryu_f64(np.log1p(of[0, 0]["hyperscore"])) and p["posterior_error"] == "1.0"
    assert p["sqrt(delta_rt_model)"] == output.ryu_f32(np.sqrt(np.float32(0.999)))


@pytest.mark.gpu
def test_cli_synthetic_two_files(tmp_path, gpu_required):
    fasta = synthetic_fasta(80, seed=31)
    fa = str(tmp_path / "db.fasta")
    open(fa, "w").write(fasta)
    dbj = {"enzyme": {"missed_cleavages": 1, "cleave_at": "KR", "restrict": "P"}, "static_mods": {"C": 57.0215}, "fasta": fa}
    host = DatabaseParameters.from_json(dbj).build(fasta)
    files = []
    for k in range(2):
        p = str(tmp_path / f"run{k}.mzML")
        write_mzml(p, synthetic_spectra(host, 60, seed=40 + k))
        files.append(p)
    cfg = {"database": dbj, "precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"ppm": [-10, 10]}, "report_psms": 2,
           "mzml_paths": files}
    cp = str(tmp_path / "c.json")
    json.dump(cfg, open(cp, "w"))
    out = str(tmp_path / "o")
    cli.main([cp, "--output_directory", out])
    hdr, rows = _read_tsv(os.path.join(out, "results.sage.tsv"))
    assert len(rows) > 60 and sorted(int(r[0]) for r in rows) == list(range(1, len(rows) + 1))  # psm_id counts from 1
    assert {r[hdr.index("filename")] for r in rows} == {"run0.mzML", "run1.mzML"}
    # rows come out in the order spectrum_fdr leaves them: best discriminant first (runner.rs:290), q-values non-decreasing
    disc = [float(r[hdr.index("sage_discriminant_score")]) for r in rows]
    assert disc == sorted(disc, reverse=True)
    q = [float(r[hdr.index("spectrum_q")]) for r in rows]
    assert q == sorted(q) and 0 < q[0] <= 1.0
    # same spectra scored by the oracle: identical hyperscore column (rows matched up by psm_id = creation order)
    sp = SpectrumProcessor(150, True, 0.0)
    orc = oracle_lib.OracleDb.from_product(host)
    want = []
    for k, p in enumerate(files):
        proc = [q for q in (sp.process(r) for r in read_mzml(p, k)) if len(q.masses) >= 15]
        of, oc, _, _ = orc.score(cli.scorer_params(cli.search_parameters(cfg)), SpectrumBatch.from_spectra(proc))
        want += [output.ryu_f64(of[i, r]["hyperscore"]) for i in range(len(proc)) for r in range(int(oc[i]))]
    by_id = sorted(rows, key=lambda r: int(r[0]))
    assert [r[hdr.index("hyperscore")] for r in by_id] == want


@pytest.mark.gpu
def test_cli_prefilter_flow(tmp_path, gpu_required):
    """database.prefilter (runner.rs:104-127, :143-238): chunked quick_score pass, merged survivors, final search — the CLI on
    the GPU against the same flow driven through the oracle."""
    fasta = synthetic_fasta(80, seed=51)
    fa = str(tmp_path / "db.fasta")
    open(fa, "w").write(fasta)
    dbj = {"enzyme": {"missed_cleavages": 1, "cleave_at": "KR", "restrict": "P"}, "static_mods": {"C": 57.0215},
           "variable_mods": {"M": [15.9949]}, "fasta": fa, "prefilter": True, "prefilter_chunk_size": 25}
    dbp = DatabaseParameters.from_json(dbj)
    full = dbp.build(fasta)
    files = []
    for k in range(2):
        p = str(tmp_path / f"run{k}.mzML")
        write_mzml(p, synthetic_spectra(full, 80, seed=60 + k))
        files.append(p)
    cfg = {"database": dbj, "precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"ppm": [-10, 10]}, "report_psms": 2,
           "mzml_paths": files}
    cp = str(tmp_path / "c.json")
    json.dump(cfg, open(cp, "w"))
    out = str(tmp_path / "o")
    logs = []
    summary = cli.run(cfg, files, out, log=logs.append)
    assert any("using 4 db chunks of size 25" in m for m in logs)
    hdr, rows = _read_tsv(os.path.join(out, "results.sage.tsv"))
    # the oracle's flow
    sp = SpectrumProcessor(150, True, 0.0)
    per_file = []
    for k, p in enumerate(files):
        per_file.append(SpectrumBatch.from_spectra([q for q in (sp.process(r) for r in read_mzml(p, k)) if len(q.masses) >= 15]))
    search = cli.search_parameters(cfg)
    pass_params = cli.scorer_params(dict(search, report_psms=search["report_psms"] + 1))
    n_targets = oracle_lib.fasta_num_targets(fasta, dbp)
    chunks, keeps = [], []
    for first in range(0, n_targets, 25):
        oc = oracle_lib.OracleDb.build_chunk(fasta, dbp, first, 25)
        keep = np.zeros(oc.n_peptides, dtype=np.uint8)
        for b in per_file:
            keep |= oc.quick_score(pass_params, b, True)
        chunks.append(oc)
        keeps.append(keep)
    final = oracle_lib.OracleDb.merge_kept(chunks, keeps, dbp)
    assert 0 < final.n_peptides < full.n_peptides
    assert any(f"generated {final.n_peptides} peptides" in m for m in logs)
    strs = final.peptide_strings()
    want = []
    for b in per_file:
        of, oc, _, _ = final.score(cli.scorer_params(search), b)
        want += [(strs[int(of[i, r]["peptide_idx"])], output.ryu_f64(of[i, r]["hyperscore"]))
                 for i in range(b.n) for r in range(int(oc[i]))]
    by_id = sorted(rows, key=lambda r: int(r[0]))
    assert [(r[hdr.index("peptide")], r[hdr.index("hyperscore")]) for r in by_id] == want
    assert summary["psms"] == len(want) > 100


def test_native_writers_match_the_python_rows(tmp_path):
    """sage_hip_write_results (C++: std::to_chars digits in ryu's layout) against output.feature_row / pin_row byte for byte —
    random PSM records incl. awkward floats, with and without the rescoring / model columns.  CPU only."""
    from types import SimpleNamespace

    from sage_amd import _lib as L
    from sage_amd.synthetic import synthetic_features
    host = DatabaseParameters(enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"), static_mods={"C": 57.0215},
                              variable_mods={"M": [15.9949]}).build(synthetic_fasta(30, seed=71), peptides_only=True)
    n = 400
    f, *_ = synthetic_features(n, seed=72)
    rng = np.random.default_rng(73)
    f["peptide_idx"] = rng.integers(0, host.n_peptides, n)
    f["file_id"] = rng.integers(0, 2, n)
    f["charge"] = rng.integers(1, 8, n)
    specials = np.array([0.0, -0.0, 1e-7, 123456.789, 1e13, 9.999999e12, 1e-5, 1e-6, 0.3, 16777216.0, np.inf, -np.inf, np.nan,
                         1.17549435e-38, 3.4028235e38, 0.001, 100.0], dtype=np.float32)
    f["average_ppm"][:len(specials)] = specials
    f["hyperscore"][:len(specials)] = specials.astype(np.float64) * 1.000000001
    f["poisson"][:8] = [-1e-300, -5e-324, -1e16, -1.5e17, -0.1, -1e-5, -9.99999e-6, -123456789012345680.0]
    f["delta_mass"][:4] = [-3.5, 0.0, 1e-8, 250.25]
    spec_ids = [f"controllerType=0 controllerNumber=1 scan={1000 + i}" if i % 3 else f"index={i}" for i in range(n)]
    psm_ids = np.arange(1, n + 1)
    order = rng.permutation(n)
    filenames = ["a.mzML", "b.mzML"]
    post = SimpleNamespace(discriminant_score=rng.normal(0, 3, n).astype(np.float32), posterior_error=-rng.gamma(2, 2, n).astype(np.float32),
                           spectrum_q=rng.uniform(0, 1, n).astype(np.float32), peptide_q=rng.uniform(0, 1, n).astype(np.float32),
                           protein_q=rng.uniform(0, 1, n).astype(np.float32))
    rtp = SimpleNamespace(aligned_rt=rng.uniform(0, 1, n).astype(np.float32), predicted_rt=rng.uniform(0, 1, n).astype(np.float32),
                          delta_rt_model=np.abs(rng.normal(0, 0.3, n)).astype(np.float32), predicted_ims=rng.uniform(0, 2, n).astype(np.float32),
                          delta_ims_model=np.abs(rng.normal(0, 0.1, n)).astype(np.float32), spectrum_q=np.zeros(n, np.float32))
    rtp.delta_rt_model[:3] = [0.0, 2.5, 0.0005]  # the pin column clamps to [0.001, 1.0] before the square root
    for with_post in (True, False):
        def post_of(i):
            if not with_post:
                return None
            d = {k: getattr(post, k)[i] for k in ("discriminant_score", "posterior_error", "spectrum_q", "peptide_q", "protein_q")}
            d.update({k: getattr(rtp, k)[i] for k in ("aligned_rt", "predicted_rt", "delta_rt_model", "predicted_ims", "delta_ims_model")})
            return d
        for fmt, row_fn, header in (("tsv", output.feature_row, output.HEADERS), ("pin", output.pin_row, output.PIN_HEADERS)):
            p = str(tmp_path / f"native_{fmt}_{with_post}.txt")
            output.write_results_native(p, fmt, host, f, order, psm_ids, filenames, spec_ids, [rtp, post] if with_post else None)
            want = ["\t".join(header)] + ["\t".join(row_fn(int(psm_ids[i]), f[i], host, filenames[int(f["file_id"][i])], spec_ids[i], post_of(i)))
                                          for i in order]
            got = open(p).read().split("\n")
            assert got[-1] == "" and got[:-1] == want, next((a, b) for a, b in zip(got, want) if a != b)


def _assert_same_run(batch, spectra, context):
    """RawBatch from the C++ reader vs the list of RawSpectrum from the Python reader"""
    assert batch.n == len(spectra), context
    for i, s in enumerate(spectra):
        g = batch.spectrum(i)
        assert g.id == s.id, (context, i)
        assert np.array_equal(np.asarray(g.mz, np.float32).view(np.uint32), np.asarray(s.mz, np.float32).view(np.uint32)), (context, i)
        assert np.array_equal(np.asarray(g.intensity, np.float32).view(np.uint32), np.asarray(s.intensity, np.float32).view(np.uint32))
        assert np.float32(g.precursor_mz) == np.float32(s.precursor_mz) and (g.precursor_charge or None) == (s.precursor_charge or None)
        assert np.float32(g.scan_start_time) == np.float32(s.scan_start_time), (context, i, g.scan_start_time, s.scan_start_time)
        assert (g.isolation_window is None) == (s.isolation_window is None), (context, i)
        if s.isolation_window is not None:
            assert tuple(np.float32(x) for x in g.isolation_window) == tuple(np.float32(x) for x in s.isolation_window)
        assert (g.inverse_ion_mobility is None) == (s.inverse_ion_mobility is None)
        if s.inverse_ion_mobility is not None:
            assert np.float32(g.inverse_ion_mobility) == np.float32(s.inverse_ion_mobility)
        assert g.file_id == s.file_id


def test_native_mzml_reader_matches_the_python_reader(tmp_path):
    """csrc/mzml_reader.cpp against sage_amd/mzml.py (itself checked against the reference's fixture): synthetic files, a
    hand-written file with the awkward cases, and the reference's own mzML when the checkout is present.  CPU only."""
    import base64
    import zlib
    from sage_amd.mzml import read_mzml_native
    d, raw = c1_raw()
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(synthetic_fasta(40, seed=81))
    spectra = synthetic_spectra(host, 50, seed=82, )
    spectra[3] = RawSpectrum(spectra[3].mz, spectra[3].intensity, spectra[3].precursor_mz, None, (1.5, 2.25), 12.5, None, 0, "scan=4")
    p = str(tmp_path / "syn.mzML")
    write_mzml(p, spectra + [raw])
    _assert_same_run(read_mzml_native(p, 3), read_mzml(p, 3), "synthetic")

    def arr64(a):
        return base64.b64encode(np.asarray(a, dtype="<f8").tobytes()).decode()

    def arr32z(a):
        s = base64.b64encode(zlib.compress(np.asarray(a, dtype="<f4").tobytes())).decode()
        return s[:20] + "\n   " + s[20:]  # line break inside the base64 text

    def spectrum(idx, sid, level, body, mzs, ints, tic=None):
        tic_cv = f'<cvParam cvRef="MS" accession="MS:1000285" name="total ion current" value="{tic}"/>' if tic is not None else ""
        return (f"<spectrum index='{idx}' id=\"{sid}\" defaultArrayLength='{len(mzs)}'>"
                f'<cvParam cvRef="MS" accession="MS:1000511" name="ms level" value="{level}"/>{tic_cv}{body}'
                '<binaryDataArrayList count="2"><binaryDataArray encodedLength="0">'
                '<cvParam cvRef="MS" accession="MS:1000523" name="64-bit float"/><cvParam cvRef="MS" accession="MS:1000576" name="no compression"/>'
                f'<cvParam cvRef="MS" accession="MS:1000514" name="m/z array"/><binary>{arr64(mzs)}</binary></binaryDataArray>'
                '<binaryDataArray><cvParam cvRef="MS" accession="MS:1000521" name="32-bit float"/>'
                '<cvParam cvRef="MS" accession="MS:1000574" name="zlib compression"/>'
                f'<cvParam cvRef="MS" accession="MS:1000515" name="intensity array"/><binary>{arr32z(ints)}</binary></binaryDataArray>'
                "</binaryDataArrayList></spectrum>")

    scan_s = ('<scanList count="1"><scan><cvParam cvRef="MS" accession="MS:1000016" name="scan start time" value="754.321" '
              'unitCvRef="UO" unitAccession="UO:0000010" unitName="second"/>'
              '<cvParam cvRef="MS" accession="MS:1002815" name="inverse reduced ion mobility" value="0.987654321"/></scan></scanList>')
    scan_m = ('<scanList count="1"><scan><cvParam cvRef="MS" accession="MS:1000016" name="scan start time" value="12.0000001" '
              'unitCvRef="UO" unitAccession="UO:0000031" unitName="minute"/></scan></scanList>')
    two_prec = ('<precursorList count="2"><precursor spectrumRef="x"><isolationWindow>'
                '<cvParam cvRef="MS" accession="MS:1000827" name="isolation window target m/z" value="0"/>'
                '</isolationWindow><selectedIonList count="1"><selectedIon>'
                '<cvParam cvRef="MS" accession="MS:1000744" name="selected ion m/z" value="0.0"/></selectedIon></selectedIonList></precursor>'
                '<precursor><isolationWindow>'
                '<cvParam cvRef="MS" accession="MS:1000827" name="isolation window target m/z" value="500.123456789"/>'
                '<cvParam cvRef="MS" accession="MS:1000828" name="isolation window lower offset" value="0.8"/>'
                '<cvParam cvRef="MS" accession="MS:1000829" name="isolation window upper offset" value="1.2"/>'
                '</isolationWindow><selectedIonList count="1"><selectedIon>'
                '<cvParam cvRef="MS" accession="MS:1000744" name="selected ion m/z" value="500.2500001"/>'
                '<cvParam cvRef="MS" accession="MS:1000041" name="charge state" value="3"/>'
                '<cvParam cvRef="MS" accession="MS:1000042" name="peak intensity" value="1e6"/></selectedIon></selectedIonList></precursor>'
                '</precursorList>')
    target_only = ('<precursorList count="1"><precursor><isolationWindow>'
                   '<cvParam cvRef="MS" accession="MS:1000827" name="isolation window target m/z" value="733.3333333"/>'
                   '</isolationWindow></precursor></precursorList>')
    mzs = [100.000001234, 250.5, 999.99999]
    body = "".join([
        spectrum(0, "ms1 scan=1", 1, scan_m, mzs, [1, 2, 3]),
        spectrum(1, "controllerType=0 scan=2 &amp; more", 2, scan_s + two_prec, mzs, [10.5, 0.0, 3e7]),
        spectrum(2, "scan=3", 2, scan_m + target_only, mzs[:2], [5, 6], tic="0"),       # dropped: TIC 0
        spectrum(3, "scan=4", 2, scan_m + target_only, mzs[:2], [5, 6], tic="123.4"),
        spectrum(4, "scan=5", 2, scan_m, [], []),                                         # no precursor, no peaks
    ])
    p2 = str(tmp_path / "hand.mzML")
    open(p2, "w").write('<?xml version="1.0" encoding="utf-8"?>\n<!-- a comment with a <spectrum> inside -->\n'
                        '<indexedmzML><mzML xmlns="http://psi.hupo.org/ms/mzml"><run id="r"><spectrumList count="5">'
                        + body + "</spectrumList></run></mzML></indexedmzML>\n")
    for level in (2, 1, None):
        want = read_mzml(p2, 0, level)
        _assert_same_run(read_mzml_native(p2, 0, level), want, f"hand level={level}")
    ms2 = read_mzml_native(p2, 7, 2)
    assert ms2.ids == ["controllerType=0 scan=2 & more", "scan=4", "scan=5"] and list(ms2.file_id) == [7, 7, 7]
    assert ms2.precursor_charge[0] == 3 and np.float32(ms2.precursor_mz[0]) == np.float32(500.2500001)
    assert np.float32(ms2.isolation_lo[0]) == np.float32(-0.8) and np.float32(ms2.scan_start_time[0]) == np.float32(754.321) / np.float32(60)
    assert np.float32(ms2.precursor_mz[1]) == np.float32(733.3333333) and np.isnan(ms2.isolation_lo[1])
    ref = "/root/reference/tests/LQSRPAAPPAPGPGQLTLR.mzML"
    if os.path.exists(ref):
        _assert_same_run(read_mzml_native(ref, 0, 2), read_mzml(ref, 0, 2), "reference fixture")
    with pytest.raises(Exception):
        read_mzml_native(str(tmp_path / "missing.mzML"))


def test_native_mzml_reader_decodes_in_parallel_with_sequential_semantics(tmp_path, monkeypatch):
    """The reader finds the <spectrum> blocks in one cheap sequential pass and decodes them on all host threads
    (mzml_reader.cpp).  The result must not depend on the thread count, spectra stay in file order, and a malformed file
    reports what a sequential read would have reported: the FIRST bad block, even when a later one is broken too."""
    import base64
    import zlib
    from sage_amd.mzml import read_mzml_native
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(synthetic_fasta(40, seed=83))
    spectra = synthetic_spectra(host, 700, seed=84)
    p = str(tmp_path / "many.mzML")
    write_mzml(p, spectra)
    runs = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("SAGE_HIP_THREADS", threads)
        runs.append(read_mzml_native(p, 0, 2))
    _assert_same_run(runs[0], read_mzml(p, 0, 2), "700 spectra, one thread")
    for r in runs[1:]:
        assert r.ids == runs[0].ids
        for name in ("peak_off", "mz", "intensities", "precursor_mz", "precursor_charge", "scan_start_time"):
            assert np.array_equal(getattr(r, name), getattr(runs[0], name), equal_nan=True), name
    # the block scan itself runs in parallel on large files: force it with small pieces, and make it fall back — a comment that
    # holds the text `<spectrum ` right where a cut would land must send the reader down the sequential scan
    monkeypatch.setenv("SAGE_HIP_THREADS", "4")
    monkeypatch.setenv("SAGE_HIP_MZML_PIECE_KB", "64")
    r = read_mzml_native(p, 0, 2)
    assert r.ids == runs[0].ids and np.array_equal(r.mz, runs[0].mz) and np.array_equal(r.peak_off, runs[0].peak_off)
    text0 = open(p).read()
    marks = [m for m in range(len(text0)) if text0.startswith("<spectrum ", m)]
    at = marks[len(marks) // 2]
    tricky = text0[:at] + "<!-- " + "x" * 70000 + " <spectrum id='not one'> " + "y" * 70000 + " -->" + text0[at:]
    pt = str(tmp_path / "tricky.mzML")
    open(pt, "w").write(tricky)
    r = read_mzml_native(pt, 0, 2)
    assert r.ids == runs[0].ids and np.array_equal(r.mz, runs[0].mz) and np.array_equal(r.intensities, runs[0].intensities)
    monkeypatch.delenv("SAGE_HIP_MZML_PIECE_KB")
    # corrupt the zlib stream of block 300's second array and cut the file inside the last block
    text = open(p).read()
    blocks = text.split("<spectrum ")
    assert len(blocks) > 600
    victim = blocks[300]
    b0 = victim.index("<binary>", victim.index("<binary>") + 1) + len("<binary>")
    b1 = victim.index("</binary>", b0)
    payload = base64.b64decode(victim[b0:b1])
    if "MS:1000574" in victim:  # zlib-compressed arrays: flip bytes in the middle of the stream
        bad = bytearray(payload)
        for k in range(len(bad) // 3, len(bad) // 3 + 6):
            bad[k] ^= 0xFF
        blocks[300] = victim[:b0] + base64.b64encode(bytes(bad)).decode() + victim[b1:]
        broken = "<spectrum ".join(blocks)
        broken = broken[:broken.rindex("</spectrum>") - 40]
        pb = str(tmp_path / "broken.mzML")
        open(pb, "w").write(broken)
        sid = victim[victim.index('id="') + 4:victim.index('"', victim.index('id="') + 4)]
        for threads in ("1", "8"):
            monkeypatch.setenv("SAGE_HIP_THREADS", threads)
            with pytest.raises(Exception) as ei:
                read_mzml_native(pb, 0, 2)
            assert "zlib" in str(ei.value) and sid in str(ei.value), str(ei.value)
    cut = text[:text.rindex("</spectrum>") - 40]
    pc = str(tmp_path / "cut.mzML")
    open(pc, "w").write(cut)
    with pytest.raises(Exception) as ei:
        read_mzml_native(pc, 0, 2)
    assert "unterminated" in str(ei.value)


# ---- inputs at the edge of the boundary: gzip, unsearchable spectra, device lists --------------------------------------------
def test_gzip_inputs_and_device_lists(tmp_path):
    import gzip
    from sage_amd.mzml import read_mzml_native
    fasta = synthetic_fasta(20, seed=61)
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(fasta)
    p = str(tmp_path / "run.mzML")
    write_mzml(p, synthetic_spectra(host, 25, seed=62))
    open(p + ".gz", "wb").write(gzip.compress(open(p, "rb").read()))
    a, b = read_mzml_native(p, 0), read_mzml_native(p + ".gz", 0)  # sage-cloudpath lib.rs:44-90: gz inputs are inflated
    assert a.n == b.n == 25 and a.ids == b.ids and np.array_equal(a.mz, b.mz) and np.array_equal(a.precursor_mz, b.precursor_mz)
    # two concatenated gzip members, then bytes that are no gzip member at all (padding): the members are the file
    half = len(open(p, "rb").read()) // 2
    data = open(p, "rb").read()
    open(p + ".2.gz", "wb").write(gzip.compress(data[:half]) + gzip.compress(data[half:]) + b"\0" * 512)
    c = read_mzml_native(p + ".2.gz", 0)
    assert c.n == 25 and c.ids == a.ids and np.array_equal(c.mz, a.mz)
    # ... but a second member with a damaged body is a corrupt file, not the end of the input (a silently truncated spectrum list)
    from sage_amd._lib import SageHipError
    second = bytearray(gzip.compress(data[half:]))
    second[len(second) // 2] ^= 0xFF
    second[len(second) // 2 + 1] ^= 0xFF
    open(p + ".3.gz", "wb").write(gzip.compress(data[:half]) + bytes(second))
    with pytest.raises(SageHipError, match="corrupt gzip"):
        read_mzml_native(p + ".3.gz", 0)
    fa = str(tmp_path / "db.fasta.gz")
    open(fa, "wb").write(gzip.compress(fasta.encode()))
    assert cli.read_text(fa) == fasta
    assert cli.parse_devices("all", 4) == [0, 1, 2, 3] and cli.parse_devices("0-2,1", 4) == [0, 1, 2, 1] and cli.parse_devices(None, 4) is None
    with pytest.raises(SystemExit):
        cli.parse_devices("0-8", 4)
    for bad in ("-1", "a-b", "0,,1", ""):
        with pytest.raises(SystemExit):
            cli.parse_devices(bad, 4)


def test_unsearchable_spectra_are_refused(tmp_path):
    """The reference panics on profile-mode MS2 spectra (spectrum.rs:280-286; no centroid term == profile) and on MS2 spectra
    without a precursor (scoring.rs:466-468) when it PROCESSES them; the reader accepts them.  Same here: the read succeeds,
    sage_hip_mzml_check_searchable refuses with the reference's message."""
    from sage_amd._lib import SageHipError
    from sage_amd.mzml import read_mzml_native
    fasta = synthetic_fasta(20, seed=63)
    host = DatabaseParameters(static_mods={"C": 57.0215}).build(fasta)
    p = str(tmp_path / "ok.mzML")
    write_mzml(p, synthetic_spectra(host, 5, seed=64))
    text = open(p).read()
    assert read_mzml_native(p, 0, check_searchable=True).n == 5
    prof = str(tmp_path / "profile.mzML")
    open(prof, "w").write(text.replace('accession="MS:1000127" name="centroid spectrum"', 'accession="MS:1000128" name="profile spectrum"', 1))
    assert read_mzml_native(prof, 0).n == 5
    with pytest.raises(SageHipError, match="contains profile data"):
        read_mzml_native(prof, 0, check_searchable=True)
    nop = str(tmp_path / "noprecursor.mzML")
    a, b = text.index("<precursorList"), text.index("</precursorList>") + len("</precursorList>")
    open(nop, "w").write(text[:a] + text[b:])
    with pytest.raises(SageHipError, match="missing MS1 precursor"):
        read_mzml_native(nop, 0, check_searchable=True)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_c_driver(tmp_path):
    import subprocess
    exe = str(tmp_path / "drive_search")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "ctest", "drive_search.c"), "-o", exe, "-L" + os.path.join(ROOT, "sage_amd"),
                           "-lsage_hip", "-Wl,-rpath," + os.path.join(ROOT, "sage_amd")])
    return exe


def test_c_driver_compiles_and_links_against_the_header(tmp_path):
    """include/sage_hip.h is a C header: a C11 translation unit that drives mzML -> results.sage.tsv through it compiles with
    gcc -Wall -Werror and links against libsage_hip.so (no C++ or HIP types leak through the boundary)."""
    from sage_amd import _lib
    _lib.load()  # (builds the library if it is missing)
    assert os.path.exists(_build_c_driver(tmp_path))


@pytest.mark.gpu
def test_c_driver_matches_the_cli(tmp_path, gpu_required):
    """tests/ctest/drive_search.c (plain C, -lsage_hip: FASTA + mzML in, results.sage.tsv out, no Python in the data path)
    against the Python CLI on the same inputs and parameters: the two files are byte-identical."""
    import subprocess
    fasta = synthetic_fasta(120, seed=71)
    fa = str(tmp_path / "db.fasta")
    open(fa, "w").write(fasta)
    dbj = {"enzyme": {"missed_cleavages": 1, "cleave_at": "KR", "restrict": "P"}, "static_mods": {"C": 57.0215}, "fasta": fa}
    host = DatabaseParameters.from_json(dbj).build(fasta)
    mz = str(tmp_path / "run.mzML")
    write_mzml(mz, synthetic_spectra(host, 400, seed=72))
    cfg = {"database": dbj, "precursor_tol": {"ppm": [-20, 20]}, "fragment_tol": {"ppm": [-10, 10]}, "predict_rt": False,
           "mzml_paths": [mz]}
    cp = str(tmp_path / "c.json")
    json.dump(cfg, open(cp, "w"))
    out = str(tmp_path / "o")
    cli.main([cp, "--output_directory", out])
    exe = _build_c_driver(tmp_path)
    c_out = str(tmp_path / "c_results.tsv")
    r = subprocess.run([exe, fa, mz, c_out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(c_out, "rb").read() == open(os.path.join(out, "results.sage.tsv"), "rb").read()
    assert "PSMs" in r.stdout


@pytest.mark.gpu
def test_cli_on_several_devices(tmp_path, gpu_required):
    """--devices: every file's spectra sharded over the workers (one host thread, device database and scorer each), results
    concatenated in input order.  A 1-GPU box names its device twice — two workers, two replicas of the index — and the
    output must equal the single-device run byte for byte, matched fragments included."""
    fasta = synthetic_fasta(80, seed=81)
    fa = str(tmp_path / "db.fasta")
    open(fa, "w").write(fasta)
    dbj = {"enzyme": {"missed_cleavages": 1, "cleave_at": "KR", "restrict": "P"}, "static_mods": {"C": 57.0215}, "fasta": fa}
    host = DatabaseParameters.from_json(dbj).build(fasta)
    files = []
    for k in range(2):
        p = str(tmp_path / f"run{k}.mzML")
        write_mzml(p, synthetic_spectra(host, 90, seed=82 + k))
        files.append(p)
    cfg = {"database": dbj, "precursor_tol": {"ppm": [-10, 10]}, "fragment_tol": {"ppm": [-10, 10]}, "report_psms": 2,
           "annotate_matches": True, "mzml_paths": files}
    cp = str(tmp_path / "c.json")
    json.dump(cfg, open(cp, "w"))
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    cli.main([cp, "--output_directory", one])
    cli.main([cp, "--output_directory", two, "--devices", "0,0,0"])
    for name in ("results.sage.tsv", "matched_fragments.sage.tsv"):
        assert open(os.path.join(one, name), "rb").read() == open(os.path.join(two, name), "rb").read(), name
    with pytest.raises(SystemExit, match="report_psms"):
        cli.run(dict(cfg, report_psms=40000), files, str(tmp_path / "x"))
    # 40 PSMs per spectrum (lists of 80 candidates: wider than a wavefront), on one device and on "three"
    cfg40 = dict(cfg, report_psms=40, annotate_matches=False)
    r1 = cli.run(cfg40, files, str(tmp_path / "k1"))
    r3 = cli.run(cfg40, files, str(tmp_path / "k3"), devices=[0, 0, 0])
    assert open(os.path.join(str(tmp_path / "k1"), "results.sage.tsv"), "rb").read() == open(os.path.join(str(tmp_path / "k3"), "results.sage.tsv"), "rb").read()
