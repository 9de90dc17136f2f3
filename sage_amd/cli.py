"""JSON-config command line for the search-and-score path — the drop-in for `sage config.json [mzML ...]`.

    python -m sage_amd.cli config.json [-o OUTPUT_DIRECTORY] [-f FASTA] [--annotate-matches] [mzml ...]

Mirrors sage-cli for this path and nothing else: the JSON schema and defaults of
crates/sage-cli/src/input.rs (Input -> Search, :298-385; `database` = sage-core Builder, database.rs:59-139), the per-file
flow of runner.rs (read mzML -> SpectrumProcessor::process -> keep MS2 with >= min_peaks peaks -> Scorer::score,
:311-325, :398-461) and the `results.sage.tsv` / `matched_fragments.sage.tsv` writers (:687-935).  Everything downstream
of Scorer::score is limited to the LDA rescoring, q-values and picked peptide / protein FDR (runner.rs:536-541, on the
device: rescore.hip) and the optional percolator .pin file; retention-time / mobility prediction, protein grouping, LFQ/TMT,
parquet, cloud IO are out of scope and their columns carry the defaults a Feature is born with (scoring.rs:576-592).  The search itself runs on the GPU through
libsage_hip.so; there is no CPU fallback.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from . import output
from ._lib import FEATURE_DTYPE as L_FEATURE_DTYPE
from .api import (DatabaseParameters, DeviceDatabase, RawBatch, Scorer, ScorerParams, SpectrumBatch, SpectrumProcessor,
                  Tolerance, device_count, predict_rt, rescore)
from .mzml import read_mzml_native


def search_parameters(cfg: dict) -> dict:
    """Input::build (input.rs:298-385): defaults of every key the path reads."""
    pc = cfg.get("precursor_charge") or [2, 4]
    if pc[0] > pc[1]:
        raise SystemExit(f"Precursor charges should be specified [low, high], user provided: [{pc[0]}, {pc[1]}]")
    iso = cfg.get("isotope_errors") or [0, 0]
    return dict(
        precursor_tol=Tolerance.from_json(cfg["precursor_tol"]), fragment_tol=Tolerance.from_json(cfg["fragment_tol"]),
        report_psms=cfg.get("report_psms") or 1, max_peaks=cfg.get("max_peaks") or 150,
        min_peaks=15 if cfg.get("min_peaks") is None else cfg["min_peaks"],
        min_matched_peaks=4 if cfg.get("min_matched_peaks") is None else cfg["min_matched_peaks"],
        max_fragment_charge=cfg.get("max_fragment_charge"), annotate_matches=bool(cfg.get("annotate_matches", False)),
        precursor_charge=(int(pc[0]), int(pc[1])), override_precursor_charge=bool(cfg.get("override_precursor_charge", False)),
        isotope_errors=(int(iso[0]), int(iso[1])), deisotope=True if cfg.get("deisotope") is None else bool(cfg["deisotope"]),
        chimera=bool(cfg.get("chimera", False)), wide_window=bool(cfg.get("wide_window", False)),
        score_type=cfg.get("score_type") or "SageHyperScore",
        predict_rt=True if cfg.get("predict_rt") is None else bool(cfg["predict_rt"]))  # input.rs:372


def scorer_params(sp: dict) -> ScorerParams:
    """The Scorer struct literal of runner.rs:492-508."""
    return ScorerParams(precursor_tol=sp["precursor_tol"], fragment_tol=sp["fragment_tol"],
                        min_matched_peaks=sp["min_matched_peaks"], min_isotope_err=sp["isotope_errors"][0],
                        max_isotope_err=sp["isotope_errors"][1], min_precursor_charge=sp["precursor_charge"][0],
                        max_precursor_charge=sp["precursor_charge"][1],
                        override_precursor_charge=sp["override_precursor_charge"], max_fragment_charge=sp["max_fragment_charge"],
                        chimera=sp["chimera"], report_psms=sp["report_psms"], wide_window=sp["wide_window"],
                        annotate_matches=sp["annotate_matches"], score_type=sp["score_type"])


def _upload(scorer, processor, raw, sp, host_preprocess, positions=False):
    """spectra of one file (RawBatch) -> resident ProcessedSpectrum batch (+ spectrum ids); None when nothing is left to search.
    positions=True: also the positions in `raw` of the spectra the batch holds."""
    if host_preprocess:  # SpectrumProcessor::process on the host (C++), spectra below min_peaks dropped (runner.rs:313)
        processed = [(i, processor.process(raw.spectrum(i))) for i in range(raw.n)]
        processed = [(i, p) for i, p in processed if len(p.masses) >= sp["min_peaks"]]
        if not processed:
            return (None, [], np.zeros(0, np.int64)) if positions else (None, [])
        out = scorer.upload(SpectrumBatch.from_spectra([p for _, p in processed])), [p.id for _, p in processed]
        return out + (np.array([i for i, _ in processed], dtype=np.int64),) if positions else out
    # ... or on the device: raw peaks in, PSMs out; spectra below min_peaks stay in the batch with zero peaks
    dbatch, _ = scorer.process_upload(raw, sp["max_peaks"], sp["deisotope"], 0.0, sp["min_peaks"])
    return (dbatch, list(raw.ids), np.arange(raw.n, dtype=np.int64)) if positions else (dbatch, list(raw.ids))


def prefilter_peptides(dbp, fasta_text, chunk, n_targets, sp, mzml_paths, processor, device, host_preprocess, log):
    """Runner::prefilter_peptides (runner.rs:143-238): search every spectrum against the FASTA one chunk of target proteins
    at a time with Scorer::quick_score, keep the peptides some spectrum picked, merge the survivors (reorder_peptides) and
    leave build_from_peptides to the device.  The scorer of this pass reports one PSM more than the final one (runner.rs:190)."""
    raws = [read_mzml_native(path, file_id=file_id, ms_level=2) for file_id, path in enumerate(mzml_paths)]
    pass_params = scorer_params(dict(sp, report_psms=sp["report_psms"] + 1))
    chunks, keeps = [], []
    for chunk_id, first in enumerate(range(0, n_targets, chunk)):
        t0 = time.time()
        log(f"pre-filtering fasta chunk {chunk_id}")
        cdb = dbp.build_chunk(fasta_text, first, chunk, peptides_only=True)
        keep = np.zeros(cdb.n_peptides, dtype=np.uint8)
        if cdb.n_peptides:
            scorer = Scorer(DeviceDatabase(cdb, device), pass_params)
            n = 0
            for raw in raws:
                if raw.n == 0:
                    continue
                dbatch, _ = _upload(scorer, processor, raw, sp, host_preprocess)
                if dbatch is None:
                    continue
                scorer.quick_score(dbatch, True if dbp.prefilter_low_memory is None else bool(dbp.prefilter_low_memory), keep)
                n += dbatch.n
                dbatch.close()
            dt = (time.time() - t0) * 1000.0
            log(f"- prefilter search:  {int(dt):8d} ms ({int(n * 1000 / (dt + 1))} spectra/s)")  # runner.rs:272-277
        log(f"found {int(keep.sum())} pre-filtered peptides for fasta chunk {chunk_id}")
        chunks.append(cdb)
        keeps.append(keep)
    return dbp.merge_kept(chunks, keeps, peptides_only=True)


def read_text(path: str) -> str:
    """FASTA text; gzip-compressed files (sage-cloudpath lib.rs:44-90 gunzips `*.gz` / `*.gzip`; here: by the 1f 8b magic)."""
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        import gzip
        raw = gzip.decompress(raw)
    return raw.decode()


def parse_devices(spec, n_visible: int):
    """--devices all | 0-7 | 0,2,3 (a device may be named twice: two workers share it, on one copy of the index)"""
    if spec is None:
        return None
    if spec == "all":
        return list(range(n_visible))
    out = []
    try:
        for part in str(spec).split(","):
            if "-" in part:
                a, b = part.split("-")
                out += list(range(int(a), int(b) + 1))
            else:
                out.append(int(part))
    except ValueError:
        raise SystemExit(f"sage_amd.cli: --devices {spec}: expected `all`, a list like 0,2,3 or a range like 0-7")
    bad = [d for d in out if d < 0 or d >= n_visible]
    if bad or not out:
        raise SystemExit(f"sage_amd.cli: --devices {spec}: {n_visible} HIP device(s) visible")
    return out


def search_file(workers, processor, raw, sp, host_preprocess, annotate, pep_mono=None):
    """Scorer::score over every MS2 spectrum of one file (runner.rs:311-325), on all the workers' devices at once: the file's
    spectra are cut into work-balanced shards that are contiguous in PRECURSOR MASS (sharding.plan_mass_shards: index
    replicated, no exchange between devices, and a device walks 1 / N of the mass-sorted index instead of all of it), one host
    thread per device preprocesses, scores and — if asked — annotates its shard, and the shards' results are merged back into
    input order, as `collect()` leaves them.  Returns (features[n, report], counts[n], ids, annotation | None)."""
    import threading

    from .sharding import estimate_work, plan_mass_shards, plan_shards, precursor_sort_mass
    shards = [np.arange(raw.n, dtype=np.int64)]
    if len(workers) > 1:
        # work per spectrum ~ peaks x queries x candidates in the precursor window (precursor mass drifts with retention time, and
        # in an open search the window size spans orders of magnitude): sharding.estimate_work
        params = scorer_params(sp)
        if pep_mono is None:  # (no masses to estimate windows from: contiguous ranges of the file, balanced on peak counts)
            shards = [np.arange(b, e, dtype=np.int64) for b, e in plan_shards(raw.peak_off, len(workers))]
        else:
            # eight blocks of the mass axis per device, every len(workers)-th block each: the window sizes of an open search span
            # orders of magnitude ALONG the mass axis, so within a block the estimate's weights still matter
            weights = estimate_work(raw.peak_off, raw.precursor_mz, raw.precursor_charge, params, pep_mono, raw.isolation_lo, raw.isolation_hi)
            narrow = not params.wide_window and params.precursor_tol.kind != "da"
            shards = plan_mass_shards(precursor_sort_mass(raw.precursor_mz, raw.precursor_charge, params), len(workers),
                                      None if narrow else weights)
    results = [None] * len(workers)
    stage_ms = [(0.0, 0.0, 0.0)] * len(workers)  # per worker: preprocess + upload, score, annotate
    errors = []

    def work(k):
        try:
            idx = shards[k]
            scorer = workers[k][1]
            part = raw if len(idx) == raw.n else raw.subset(idx)
            if part.n == 0:
                return
            t0 = time.time()
            dbatch, ids, kept = _upload(scorer, processor, part, sp, host_preprocess, positions=True)
            if dbatch is None:
                return
            t1 = time.time()
            feats, counts = scorer.score_resident(dbatch)
            feats, counts = feats.copy(), counts.copy()
            t2 = time.time()
            ann = scorer.annotate(dbatch, feats, counts) if annotate else None
            dbatch.close()
            stage_ms[k] = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.time() - t2) * 1e3)
            results[k] = (feats, counts, ids, ann, idx[kept])  # (idx[kept]: positions in the file of the batch's spectra)
        except BaseException as exc:  # noqa: BLE001 — re-raised on the calling thread
            errors.append(exc)

    if len(workers) == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(k,)) for k in range(len(workers))]
        [t.start() for t in ts]
        [t.join() for t in ts]
    if errors:
        raise errors[0]
    parts = [r for r in results if r is not None]
    if not parts:
        return None
    # merge: rows in the order of their spectra's positions in the file (one part: already so)
    feats = np.concatenate([p[0] for p in parts], axis=0)
    counts = np.concatenate([p[1] for p in parts])
    ids = [i for p in parts for i in p[2]]
    where = np.concatenate([p[4] for p in parts])
    perm = np.argsort(where, kind="stable")
    report = feats.shape[1]
    ann = None
    if annotate:  # Fragments: one run per PSM slot, offsets per part — gather the runs in the merged slot order
        lens = np.concatenate([np.diff(p[3][0].astype(np.int64)) for p in parts])  # per slot (rows x report), concatenated parts
        starts = np.concatenate([p[3][0][:-1].astype(np.int64) + b for p, b in
                                 zip(parts, np.cumsum([0] + [int(p[3][0][-1]) for p in parts[:-1]]))])
        slot = (perm[:, None] * report + np.arange(report)[None, :]).reshape(-1)
        new_off = np.zeros(len(slot) + 1, dtype=np.uint64)
        new_off[1:] = np.cumsum(lens[slot])
        gather = np.repeat(starts[slot] - new_off[:-1].astype(np.int64), lens[slot]) + np.arange(int(new_off[-1]))
        ann = (new_off, {k: np.concatenate([p[3][1][k] for p in parts])[gather] for k in parts[0][3][1]})
    feats, counts, ids = feats[perm], counts[perm], [ids[i] for i in perm]
    valid = np.arange(report)[None, :] < counts[:, None]
    feats["spec_index"] = np.where(valid, np.arange(len(counts), dtype=np.uint32)[:, None], feats["spec_index"])  # row of `ids`
    search_file.last_stage_ms = tuple(max(x[i] for x in stage_ms) for i in range(3))  # (the slowest worker's, per stage)
    return feats, counts, ids, ann


def run(cfg: dict, mzml_paths, output_directory: str, device: int = 0, log=print, host_preprocess: bool = False,
        write_pin: bool = False, devices=None) -> dict:
    if device_count() <= 0:
        raise SystemExit("sage_amd.cli: no HIP device visible — libsage_hip has no CPU fallback")
    sp = search_parameters(cfg)
    dbp = DatabaseParameters.from_json(cfg["database"])
    if not dbp.fasta:
        raise SystemExit("`database.fasta` must be set. For more information try '--help'")
    # (any report_psms the library takes: lists of max(50, 2 * report_psms) candidates live in LDS while they fit a compute unit,
    # in a global-memory workspace beyond — capi.hip: enqueue_compute; the prefilter pass scores with report_psms + 1)
    need = sp["report_psms"] + (1 if dbp.prefilter else 0)
    if need > 32767:
        raise SystemExit(f"sage_amd.cli: report_psms = {sp['report_psms']}: this build supports report_psms <= 32767")
    devices = list(devices) if devices else [device]
    device = devices[0]
    t0 = time.time()
    fasta_text = read_text(dbp.fasta)
    params = scorer_params(sp)
    processor = SpectrumProcessor(sp["max_peaks"], sp["deisotope"], 0.0)  # (no TMT reporter cut-off: quant is out of scope)
    host = None
    if dbp.prefilter:  # runner.rs:104-127
        chunk = dbp.auto_prefilter_chunk_size(fasta_text)
        n_targets = dbp.num_targets(fasta_text)
        if chunk < n_targets:
            log(f"using {(n_targets + chunk - 1) // chunk} db chunks of size {chunk}")
            host = prefilter_peptides(dbp, fasta_text, chunk, n_targets, sp, mzml_paths, processor, device, host_preprocess, log)
    if host is None:
        host = dbp.build(fasta_text, peptides_only=True)  # digest / modify / sort / dedup on the host ...
    # ... build_from_peptides on the device (index_build.hip): the index is replicated on every device that searches
    workers, first = [], {}
    for d in devices:  # one copy of the index per distinct device; a device named again gets a second handle on it
        if d in first:
            workers.append((first[d][0], first[d][1].clone()))  # (sage_hip_scorer_clone: own streams and working set)
        else:
            dev_db = DeviceDatabase(host, d)
            first[d] = (dev_db, Scorer(dev_db, params))
            workers.append(first[d])
    log(f"generated {host.n_peptides} peptides and their fragment index in {int((time.time() - t0) * 1000)}ms" +
        (f" on {len(devices)} devices" if len(devices) > 1 else ""))
    os.makedirs(output_directory, exist_ok=True)
    feats_all, meta, frags = [], [], []  # per PSM: (filename, spectrum id); matched-fragment rows
    psm_id = 1  # PSM_COUNTER starts at 1 (scoring.rs:163)
    n_searched = 0
    search_ms = 0.0
    stage_totals = {"file_io_ms": 0.0, "preprocess_upload_ms": 0.0, "score_ms": 0.0, "annotate_ms": 0.0}
    t_run = time.time()
    # The reader works one file ahead of the search (the reference reads and preprocesses its files in parallel batches,
    # runner.rs:450-461): file k + 1 is parsed on the host threads (csrc/mzml_reader.cpp decodes the spectra of a file in
    # parallel, outside the GIL) while the devices score file k.
    from concurrent.futures import ThreadPoolExecutor

    def read_file(file_id, path):
        t0 = time.time()
        raw = read_mzml_native(path, file_id=file_id, ms_level=2, check_searchable=True)
        return raw, (time.time() - t0) * 1000.0

    mzml_paths = list(mzml_paths)
    reader = ThreadPoolExecutor(max_workers=1)
    ahead = reader.submit(read_file, 0, mzml_paths[0]) if mzml_paths else None
    for file_id, path in enumerate(mzml_paths):
        try:
            raw, io_ms = ahead.result()
        except BaseException:
            reader.shutdown(wait=True)
            raise
        ahead = reader.submit(read_file, file_id + 1, mzml_paths[file_id + 1]) if file_id + 1 < len(mzml_paths) else None
        log(f"- file IO: {int(io_ms):8d} ms")
        if raw.n == 0:
            continue
        t0 = time.time()
        found = search_file(workers, processor, raw, sp, host_preprocess, sp["annotate_matches"], host.pep_mono)
        if found is None:
            continue
        feats, counts, ids, ann = found
        n_batch = len(counts)
        dt = (time.time() - t0) * 1000.0
        search_ms += dt
        n_searched += n_batch
        log(f"- search:  {int(dt):8d} ms ({int(n_batch * 1000 / (dt + 1))} spectra/s)")  # runner.rs:327-330
        pre_ms, score_ms, ann_ms = getattr(search_file, "last_stage_ms", (0.0, 0.0, 0.0))
        log(f"    of which preprocessing + upload {pre_ms:.1f} ms, Scorer::score on the device {score_ms:.1f} ms" +
            (f", fragment annotation {ann_ms:.1f} ms" if sp["annotate_matches"] else ""))
        stage_totals["file_io_ms"] += io_ms
        stage_totals["preprocess_upload_ms"] += pre_ms
        stage_totals["score_ms"] += score_ms
        stage_totals["annotate_ms"] += ann_ms
        off, arr = ann if ann is not None else (None, None)
        name = os.path.basename(path)
        # the PSMs of the file in (spectrum, rank) order — the order Scorer::score results are collected in (runner.rs:325)
        spec_of = np.repeat(np.arange(n_batch), counts.astype(np.int64))
        rank_of = np.arange(len(spec_of)) - np.repeat(np.cumsum(counts.astype(np.int64)) - counts, counts.astype(np.int64))
        part = feats[spec_of, rank_of].copy()
        part["file_id"] = file_id
        feats_all.append(part)
        meta += [(psm_id + j, name, ids[i]) for j, i in enumerate(spec_of.tolist())]
        if arr is not None:
            for j, (i, r) in enumerate(zip(spec_of.tolist(), rank_of.tolist())):
                slot = i * params.report_psms + r
                frags.append(output.fragment_rows(psm_id + j, int(off[slot]), int(off[slot + 1]), arr))
        psm_id += len(part)
    reader.shutdown(wait=True)
    # runner.rs:536-541: spectrum_fdr (LDA or heuristic, sort, q-values), picked_peptide, picked_protein — on the device.
    # (protein grouping is outside this path: its columns keep the defaults, see output.py)
    flat = np.concatenate(feats_all) if feats_all else np.zeros(0, dtype=L_FEATURE_DTYPE)
    post = None
    rtp = None
    order = range(len(flat))
    rescore_summary = {}
    if len(flat):
        t0 = time.time()
        model_inputs = {}
        if sp["predict_rt"]:  # runner.rs:513-530: poisson-sorted q-values, global_alignment, retention / mobility models
            off, seq, mono = host.feature_peptides(flat["peptide_idx"])
            rtp = predict_rt(flat, len(mzml_paths), off, seq, mono, device=device)
            for a in rtp.alignments:
                log(f"aligning file #{int(a['file_id'])}: y = {float(a['slope']):.4f}x + {float(a['intercept']):.4f}")
            log(f"aligned retention times across {len(mzml_paths)} files")
            if rtp.rt_fitted:
                log(f"- fit retention time model, rsq = {rtp.rt_r2}")
            if rtp.ims_fitted:
                log(f"- fit mobility model, rsq = {rtp.ims_r2}")
            else:
                log("Mobility model failed to train")
            model_inputs = dict(aligned_rt=rtp.aligned_rt, delta_rt_model=rtp.delta_rt_model, delta_ims_model=rtp.delta_ims_model)
        pk, npk, prk, npr = host.competition_keys(flat["peptide_idx"])
        res = rescore(flat, sp["precursor_tol"], pk, npk, prk, npr, device=device, **model_inputs)
        if not res.lda_fitted:
            log("linear model fitting failed, falling back to heuristic discriminant score")  # runner.rs:285
        post = res
        order = [int(i) for i in res.order]
        log(f"discovered {res.passing_spectrum} target peptide-spectrum matches at 1% FDR")  # runner.rs:576-587
        log(f"discovered {res.passing_peptide} target peptides at 1% FDR")
        log(f"discovered {res.passing_protein} target proteins (supported by proteotypic peptides only) at 1% FDR")
        rescore_summary = {"lda_fitted": res.lda_fitted, "q_spectrum": res.passing_spectrum, "q_peptide": res.passing_peptide,
                           "q_protein": res.passing_protein, "rescore_ms": (time.time() - t0) * 1000.0,
                           "rescore_device_ms": res.device_ms}

    if rescore_summary:
        log(f"- rescoring (RT / IM models, LDA, q-values, picked FDR): {int(rescore_summary['rescore_ms']):8d} ms")
    # writers: C++ (sage_hip_write_results) — byte-identical to output.feature_row / pin_row, which tests/test_cli_io.py checks
    t_write = time.time()
    filenames = [os.path.basename(p) for p in mzml_paths]
    psm_ids = [m[0] for m in meta]
    spec_ids = [m[2] for m in meta]
    results = os.path.join(output_directory, "results.sage.tsv")
    output.write_results_native(results, "tsv", host, flat, list(order), psm_ids, filenames, spec_ids, [rtp, post])
    paths = [results]
    if sp["annotate_matches"]:
        fp = os.path.join(output_directory, "matched_fragments.sage.tsv")
        output.write_fragments(fp, [r for i in order for r in frags[i]])
        paths.append(fp)
    if write_pin:  # runner.rs:655-660
        pp = os.path.join(output_directory, "results.sage.pin")
        output.write_results_native(pp, "pin", host, flat, list(order), psm_ids, filenames, spec_ids, [rtp, post])
        paths.append(pp)
    stage_totals["write_ms"] = (time.time() - t_write) * 1e3
    log(f"- writing: {int(stage_totals['write_ms']):8d} ms")
    stage_totals["total_after_index_ms"] = (time.time() - t_run) * 1e3
    summary = {"version": "sage-hip 0.1 (search-and-score path of sage 0.15.0-beta.2)", "psms": len(flat),
               "spectra_searched": n_searched, "search_ms": search_ms, "stages": stage_totals, "output_paths": paths, **rescore_summary}
    with open(os.path.join(output_directory, "results.json"), "w") as fh:
        json.dump(dict(cfg, output_paths=paths, summary=summary), fh, indent=2, default=str)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(prog="sage_amd.cli", description="GPU search-and-score path of Sage (JSON-config CLI)")
    ap.add_argument("parameters", help="path to configuration parameters (JSON file)")
    ap.add_argument("mzml_paths", nargs="*", help="paths to mzML files to process. Overrides mzML files listed in the configuration file.")
    ap.add_argument("-f", "--fasta", help="path to FASTA database. Overrides the FASTA file specified in the configuration file.")
    ap.add_argument("-o", "--output_directory", help="where to place output files. Overrides the directory specified in the configuration file.")
    ap.add_argument("--annotate-matches", action="store_true", help="write matched fragments output file")
    ap.add_argument("--write-pin", action="store_true", help="write percolator-compatible `.pin` output files")
    # accepted for command-line compatibility with sage (sage-cli/src/main.rs:55-97)
    ap.add_argument("--batch-size", type=int, help="number of files sage loads and searches in parallel; files are searched one "
                                                   "after the other here (every file is one resident GPU batch)")
    ap.add_argument("--parquet", action="store_true", help="not supported by this path (tsv only)")
    ap.add_argument("--write-report", action="store_true", help="not supported by this path")
    ap.add_argument("--disable-telemetry-i-dont-want-to-improve-sage", action="store_true", dest="disable_telemetry",
                    help="accepted; this implementation never sends telemetry")
    ap.add_argument("--stack-size", type=int, help="accepted; no effect")
    ap.add_argument("--device", type=int, default=0, help="the HIP device that searches (default 0)")
    ap.add_argument("--devices", help="search on several devices at once: all | 0-7 | 0,2,3 — every file's spectra are sharded "
                                      "over them, the index is replicated (sage uses every core of the machine; this is the same)")
    ap.add_argument("--host-preprocess", action="store_true", help="run SpectrumProcessor::process on the host instead of the device")
    args = ap.parse_args(argv)
    if args.parquet or args.write_report:
        raise SystemExit("sage_amd.cli: --parquet / --write-report are outside the search-and-score path (tsv / pin output only)")
    if args.batch_size is not None and args.batch_size < 1:
        raise SystemExit("error: invalid value for '--batch-size': must be >= 1")
    cfg = json.load(open(args.parameters))
    if args.fasta:
        cfg.setdefault("database", {})["fasta"] = args.fasta
    if args.annotate_matches:
        cfg["annotate_matches"] = True
    mzml = args.mzml_paths or cfg.get("mzml_paths")
    if not mzml:
        raise SystemExit("'mzml_paths' must be provided!")
    out = args.output_directory or cfg.get("output_directory") or os.getcwd()
    summary = run(cfg, mzml, out, args.device, host_preprocess=args.host_preprocess,
                  write_pin=args.write_pin or bool(cfg.get("write_pin", False)),
                  devices=parse_devices(args.devices, device_count()))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
