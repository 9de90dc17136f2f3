"""Product host code (index builder, SpectrumProcessor) vs the oracle, bit-exact.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from sage_amd import _lib as L
from sage_amd.api import DatabaseParameters, RawSpectrum, SpectrumProcessor
from sage_amd.synthetic import synthetic_fasta, synthetic_spectra
from test_oracle_golden import c1_batch, load_c1


def test_library_exports_every_declared_symbol():
    lib = L.load()
    import re, os
    hdr = open(os.path.join(os.path.dirname(L.__file__), "..", "include", "sage_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sage_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libsage_hip.so does not export {name}"
    assert sorted(L.EXPORTED_SYMBOLS) == declared
    assert lib.sage_hip_abi_version() == 6
    assert C.sizeof(L.SageScorerParams) == 44 and L.FEATURE_DTYPE.itemsize == 120


def assert_db_equal(prod, orc, check_missed=True):
    a = orc.arrays()
    assert prod.n_peptides == orc.n_peptides and prod.n_fragments == orc.n_fragments
    assert prod.bucket_size == orc.bucket_size
    np.testing.assert_array_equal(prod.pep_mono.view(np.uint32), a["pep_mono"].view(np.uint32))
    np.testing.assert_array_equal(prod.seq_off, a["seq_off"])
    np.testing.assert_array_equal(prod.seq, a["seq"])
    np.testing.assert_array_equal(prod.mods.view(np.uint32), a["mods"].view(np.uint32))
    np.testing.assert_array_equal(np.isnan(prod.nterm), np.isnan(a["nterm"]))
    np.testing.assert_array_equal(np.nan_to_num(prod.nterm), np.nan_to_num(a["nterm"]))
    np.testing.assert_array_equal(np.nan_to_num(prod.cterm), np.nan_to_num(a["cterm"]))
    np.testing.assert_array_equal(prod.decoy, a["decoy"])
    if check_missed:
        np.testing.assert_array_equal(prod.missed_cleavages, a["missed"])
    np.testing.assert_array_equal(prod.fragments["peptide_index"], a["frag_pep"])
    np.testing.assert_array_equal(prod.fragments["fragment_mz"].view(np.uint32), a["frag_mz"].view(np.uint32))
    np.testing.assert_array_equal(prod.min_value.view(np.uint32), a["min_value"].view(np.uint32))


PARAM_SETS = {
    "default": dict(),
    "c1_config": dict(bucket_size=16384, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                      static_mods={"C": 57.0216}),
    "varmods": dict(bucket_size=64, enzyme=dict(missed_cleavages=2, min_len=5, max_len=30, cleave_at="KR", restrict="P"),
                    static_mods={"C": 57.0215}, variable_mods={"M": [15.9949], "[": [42.010565], "^E": [-18.0106]},
                    max_variable_mods=2),
    "abcxyz_nodecoy": dict(bucket_size=100, ion_kinds=["a", "b", "c", "x", "y", "z"], min_ion_index=1,
                           generate_decoys=False, enzyme=dict(missed_cleavages=0, cleave_at="KR", restrict="")),
    "aspn_semi": dict(bucket_size=512, enzyme=dict(missed_cleavages=1, cleave_at="D", restrict="", c_terminal=False,
                                                   semi_enzymatic=True, min_len=6, max_len=12)),
    "nonspecific": dict(bucket_size=256, enzyme=dict(cleave_at="", min_len=7, max_len=8), peptide_min_mass=600.0),
    "terminal_mods": dict(bucket_size=128, static_mods={"^": 229.1629, "K": 229.1629},
                          variable_mods={"$": [0.984], "]": [14.0156], "S": [79.9663, 541.0611]}, max_variable_mods=3,
                          enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P")),
}


@pytest.mark.parametrize("name", sorted(PARAM_SETS))
def test_index_builder_matches_oracle_q99536(name):
    d, *_ = load_c1()
    params = DatabaseParameters(**PARAM_SETS[name])
    prod = params.build(d["fasta"])
    orc = oracle_lib.OracleDb.build(d["fasta"], params)
    assert prod.n_peptides > 0
    assert_db_equal(prod, orc, check_missed=(name != "aspn_semi"))
    strs = orc.peptide_strings()
    for i in range(0, prod.n_peptides, max(1, prod.n_peptides // 50)):
        assert prod.peptide_string(i) == strs[i]
        assert prod.peptide_proteins(i) == orc.peptide_proteins(i) or prod.decoy[i]  # decoys: tag prefix only in product


def test_index_builder_matches_oracle_synthetic_proteome():
    fasta = synthetic_fasta(120, seed=7)
    params = DatabaseParameters(bucket_size=1024, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215}, variable_mods={"M": [15.9949]})
    prod = params.build(fasta)
    orc = oracle_lib.OracleDb.build(fasta, params)
    assert prod.n_peptides > 5000
    assert_db_equal(prod, orc)
    # reference invariants (crates/sage/tests/integration.rs:44-58)
    bs = prod.bucket_size
    for c in range(len(prod.min_value)):
        chunk = prod.fragments[c * bs:(c + 1) * bs]
        assert np.all(np.diff(chunk["peptide_index"].astype(np.int64)) >= 0)
        assert np.all(chunk["fragment_mz"] >= prod.min_value[c])
        if c + 1 < len(prod.min_value):
            assert np.all(chunk["fragment_mz"] <= prod.min_value[c + 1])
    assert np.all(np.diff(prod.pep_mono) >= 0)


def test_fasta_with_decoy_tag_and_reference_digestion_vector():
    # database.rs:595-671
    fasta = "\n        >sp|AAAAA\n        MEWKLEQSMREQALLKAQLTQLK\n        >sp|BBBBB\n        RMEWKLEQSMREQALLKAQLTQLK\n        "
    params = DatabaseParameters(bucket_size=128, enzyme=dict(missed_cleavages=1, min_len=6, max_len=10, cleave_at="KR",
                                                             restrict="P", c_terminal=True, semi_enzymatic=False),
                                peptide_min_mass=150.0, variable_mods={"[": [42.0]}, generate_decoys=False)
    prod = params.build(fasta)
    got = [prod.peptide_string(i) for i in range(prod.n_peptides)]
    assert got == ["EQALLK", "LEQSMR", "AQLTQLK", "MEWKLEQSMR", "[+42]-MEWKLEQSMR"]
    assert [prod.peptide_proteins(i) for i in range(4)] == ["sp|AAAAA;sp|BBBBB"] * 4
    assert prod.peptide_proteins(4) == "sp|AAAAA"
    # decoy-tagged entries are dropped when decoys are generated, kept as decoys otherwise (fasta.rs:30-34, 64-72)
    f2 = ">sp|T1\nMEWKLEQSMREQALLK\n>rev_sp|T1\nKLLAQERMSQELKWEM\n"
    a = DatabaseParameters(generate_decoys=True, peptide_min_mass=150.0).build(f2)
    b = DatabaseParameters(generate_decoys=False, peptide_min_mass=150.0).build(f2)
    oa = oracle_lib.OracleDb.build(f2, DatabaseParameters(generate_decoys=True, peptide_min_mass=150.0))
    ob = oracle_lib.OracleDb.build(f2, DatabaseParameters(generate_decoys=False, peptide_min_mass=150.0))
    assert_db_equal(a, oa)
    assert_db_equal(b, ob)
    assert b.decoy.sum() > 0


def test_spectrum_processor_matches_oracle():
    d, s, mz, it = load_c1()
    for top_n, deiso, min_mz, z in [(100, True, 0.0, 3), (150, True, 0.0, 0), (150, False, 0.0, 2), (20, False, 0.0, 2),
                                    (50, True, 300.0, 2), (1000, True, 0.0, 4)]:
        sp = SpectrumProcessor(top_n, deiso, min_mz)
        out = sp.process(RawSpectrum(mz, it, 643.03, z or None))
        om, oi, tic = oracle_lib.process_ms2(top_n, deiso, min_mz, mz, it, z)
        np.testing.assert_array_equal(out.masses.view(np.uint32), om.view(np.uint32))
        np.testing.assert_array_equal(out.intensities.view(np.uint32), oi.view(np.uint32))
        assert np.float32(out.total_ion_current) == np.float32(tic)
        assert np.all(np.diff(out.masses) >= 0)
    # synthetic spectra with isotope envelopes and exact duplicates
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(0, 400))
        base = np.sort(rng.uniform(150, 1500, n)).astype(np.float32)
        extra = (base[: n // 3] + np.float32(1.00335) / rng.integers(1, 4, n // 3)).astype(np.float32)
        m = np.sort(np.concatenate([base, extra, base[: n // 10]])).astype(np.float32)
        i = rng.lognormal(8, 1.5, len(m)).astype(np.float32)
        i[: len(i) // 7] = i[0] if len(i) else 0
        for deiso in (True, False):
            out = SpectrumProcessor(150, deiso, 0.0).process(RawSpectrum(m, i, 700.0, int(rng.integers(0, 5)) or None))
            om, oi, tic = oracle_lib.process_ms2(150, deiso, 0.0, m, i, out.precursor_charge or 0)
            np.testing.assert_array_equal(out.masses.view(np.uint32), om.view(np.uint32))
            np.testing.assert_array_equal(out.intensities.view(np.uint32), oi.view(np.uint32))
            assert np.float32(out.total_ion_current) == np.float32(tic)


# ---- the `prefilter` flow (sage-cli runner.rs:104-127, :143-238): chunked build, merge of the kept peptides ------------------
def test_prefilter_chunks_and_merge_match_oracle():
    # shared tryptic peptides between proteins of different chunks, and a protein whose reversed peptide is a target elsewhere
    fasta = synthetic_fasta(40, seed=17) + ">sp|DUP1|X\nMKAAAGGGLLLKDDDEEEFFFKWWWYYYR\n>sp|DUP2|Y\nMKAAAGGGLLLKCCCHHHIIIK\n" \
        + ">sp|PAL1|P\nMRACDEFGHIK\n>sp|PAL2|Q\nMRAIHGFEDCK\n"
    params = DatabaseParameters(bucket_size=512, enzyme=dict(missed_cleavages=1, cleave_at="KR", restrict="P"),
                                static_mods={"C": 57.0215}, variable_mods={"M": [15.9949]})
    n_targets = params.num_targets(fasta)
    assert n_targets == oracle_lib.fasta_num_targets(fasta, params) == 44
    chunk = 9
    prod_chunks, orc_chunks, keeps = [], [], []
    rng = np.random.default_rng(5)
    for first in range(0, n_targets, chunk):
        pc = params.build_chunk(fasta, first, chunk)
        oc = oracle_lib.OracleDb.build_chunk(fasta, params, first, chunk)
        assert_db_equal(pc, oc)
        prod_chunks.append(pc)
        orc_chunks.append(oc)
        keeps.append((rng.random(pc.n_peptides) < 0.4).astype(np.uint8))
    # the whole FASTA as one chunk is the ordinary build
    assert_db_equal(params.build_chunk(fasta, 0, n_targets), oracle_lib.OracleDb.build(fasta, params))
    merged = params.merge_kept(prod_chunks, keeps)
    omerged = oracle_lib.OracleDb.merge_kept(orc_chunks, keeps, params)
    assert 0 < merged.n_peptides <= sum(int(k.sum()) for k in keeps)
    assert_db_equal(merged, omerged, check_missed=False)
    strs = omerged.peptide_strings()
    for i in range(merged.n_peptides):
        assert merged.peptide_string(i) == strs[i]
        if not merged.decoy[i]:
            assert merged.peptide_proteins(i) == omerged.peptide_proteins(i)
    # keeping everything reproduces the peptides of the one-shot build, except decoys that are targets of ANOTHER chunk:
    # those are only dropped inside a chunk (database.rs:212), the merge keeps them and clears nothing
    full = params.build(fasta)
    everything = params.merge_kept(prod_chunks, [np.ones(c.n_peptides, np.uint8) for c in prod_chunks])
    full_set = {(full.peptide_string(i), bool(full.decoy[i])) for i in range(full.n_peptides)}
    all_set = {(everything.peptide_string(i), bool(everything.decoy[i])) for i in range(everything.n_peptides)}
    assert full_set <= all_set
    # peptides_only merge: same peptide arrays, no fragments
    po = params.merge_kept(prod_chunks, keeps, peptides_only=True)
    assert po.n_peptides == merged.n_peptides and not po.has_fragments
    np.testing.assert_array_equal(po.pep_mono.view(np.uint32), merged.pep_mono.view(np.uint32))


def test_auto_prefilter_chunk_size_matches_oracle():
    fasta = synthetic_fasta(60, seed=19)
    for kw in (dict(), dict(variable_mods={"M": [15.9949], "S": [79.9663, 541.0611]}, max_variable_mods=3),
               dict(enzyme=dict(missed_cleavages=2, cleave_at="KR", restrict="P"))):
        params = DatabaseParameters(**kw)
        assert params.auto_prefilter_chunk_size(fasta) == oracle_lib.prefilter_chunk_size(fasta, params) == 60
        fixed = DatabaseParameters(prefilter_chunk_size=7, **kw)
        assert fixed.auto_prefilter_chunk_size(fasta) == 7
    # enough estimated modified peptides for more than one chunk: (keys + 1) * 2^max_variable_mods * digests / 2^23
    big = DatabaseParameters(variable_mods={"M": [15.9949]}, max_variable_mods=13)
    got = big.auto_prefilter_chunk_size(fasta)
    assert got == oracle_lib.prefilter_chunk_size(fasta, big) and 0 < got < 60
