"""Parity tests proper: the HIP path (through the C-ABI, rnacode_amd.api -> librnacode_hip.so)
against (1) golden vectors captured from the unmodified reference objects and (2) the CPU
oracle on the same seeded inputs.  Bars: HSS coordinates / frames / strands bit-exact; scores
and per-sample maxima exact in binary32; mu, lambda and p-values within 1e-6 relative
(north_star tolerance; the fit runs in binary64 on the device with a different summation
order and a different libm than the host)."""
import numpy as np
import pytest

from conftest import ROOT, golden_names, hss_key, load_golden
from helpers import block_from_golden, close_p, param_kwargs

pytestmark = pytest.mark.gpu

REL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    from rnacode_amd import api
    c = api.Context(0)
    yield c
    c.close()


def api_params(doc):
    from rnacode_amd import api
    kw = param_kwargs(doc)
    return api.default_params(seed_base=doc["seed_base"], **kw)


def close(a, b, rel=REL):
    return abs(a - b) <= rel * max(1.0, abs(b))


def run_golden(ctx, doc, max_blocks=None):
    from rnacode_amd import api
    entries = doc["blocks"][:max_blocks] if max_blocks else doc["blocks"]
    blocks = []
    for e in entries:
        blocks.append(block_from_golden(e))   # blocks the reference skipped have no tree: they are skipped here as well
    batch = api.Batch(ctx, blocks, api_params(doc)).run()
    return entries, blocks, batch


def check_entry(batch, i, entry):
    from rnacode_amd import api
    ref = entry["ref"]
    if "skipped" in ref:
        assert batch.status(i) == api.RC_ERR_SKIP
        return
    assert batch.status(i) == api.RC_OK
    fwd, rev = batch.getModels(i)
    for got, want in ((fwd, ref["models"]), (rev, ref["modelsRev"])):
        for g, w in zip(got, want):
            assert np.float32(g["dist"]) == np.float32(w["dist"])
            np.testing.assert_array_equal(np.float32(g["freqs"]), np.float32(w["freqs"]))
            np.testing.assert_array_equal(np.float32(g["scores"]), np.float32(w["scores"]))
            np.testing.assert_array_equal(np.float32(g["probs"]), np.float32(w["probs"]))   # counts[h] of calculateBG, score.c:186-190
            assert np.float32(g["kappa"]) == np.float32(w["kappa"])
    got = sorted(batch.scoreAln(i), key=hss_key)
    want = sorted(ref["hss"], key=hss_key)
    assert len(got) == len(want)
    rc, mu, lam = batch.getExtremeValuePars(i)
    assert rc == ref["evd_rc"]
    for g, w in zip(got, want):
        for k in ("strand", "frame", "startSite", "endSite", "start", "end", "startGenomic", "endGenomic"):
            assert g[k] == w[k], (k, g, w)
        assert np.float32(g["score"]) == np.float32(w["score"])
        if rc == 1:
            assert close_p(g["pvalue"], w["pvalue"]), (g, w)
        else:
            assert g["pvalue"] == 99.0
    if ref["maxScores"]:
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(ref["maxScores"]))
    if rc == 1:
        assert close(mu, ref["mu"]) and close(lam, ref["lambda"]), (mu, lam, ref["mu"], ref["lambda"])


@pytest.mark.parametrize("name", golden_names())
def test_golden_parity(ctx, name):
    doc = load_golden(name)
    entries, blocks, batch = run_golden(ctx, doc)
    assert batch.clamped() == 0
    for i, e in enumerate(entries):
        check_entry(batch, i, e)
    batch.close()


def test_mt_stream_matches_twister(ctx):
    from oracle import binding as ob
    for seed in (0, 1, 42, 5489, 2**32 - 1):
        got = ctx.mt_stream(seed, 2000)
        np.testing.assert_array_equal(got, ob.mt_stream(seed, 2000))


def test_native_score_matrix_bit_exact(ctx):
    """S[b][i] of the native block (score.c:811-848), every cell, against the oracle."""
    from oracle import binding as ob
    for name in ("coding_aln_n100", "synth_6x120_n200", "edge_cases_n50", "synth_20x45_n50"):
        doc = load_golden(name)
        entries, blocks, batch = run_golden(ctx, doc)
        p = ob.default_params(doc["samples"], doc.get("params", {}).get("blosum", 62))
        for i, (e, b) in enumerate(zip(entries, blocks)):
            if "skipped" in e["ref"]:
                continue
            rows = [r.seq for r in b.rows]
            names = [r.name for r in b.rows]
            for strand, rr in ((0, rows), (1, ob.rev_aln(rows))):
                m = ob.get_models(b.tree, rr, names, b.kappa, p.blosum)
                S = ob.score_matrix(rr, m, p)
                L = b.ref_len
                for f in range(3):
                    got = batch.native_S(i, strand, f)
                    sites = (L - f) // 3
                    assert got.shape == (sites, sites)
                    for a in range(sites):
                        jj = np.arange(a, sites)
                        np.testing.assert_array_equal(got[a, a:], S[3 * a + 1 + f, 3 * jj + 3 + f])
        batch.close()


def test_backtrack_matches_reference(ctx):
    for name in ("coding_aln_n100", "synth_6x120_n200", "edge_cases_n50"):
        doc = load_golden(name)
        entries, blocks, batch = run_golden(ctx, doc)
        for i, e in enumerate(entries):
            bt = e["ref"].get("backtrack")
            if not bt:
                continue
            top = sorted(e["ref"]["hss"], key=hss_key)[0]
            st, z, tr = batch.backtrack(i, 0 if top["strand"] == "+" else 1, bt["b"], bt["i"])
            idx = list(range(bt["i"], bt["b"] + 1, -3))
            for k in range(1, blocks[i].n):
                assert list(st[k, idx]) == bt["k"][k - 1]["states"]
                assert list(z[k, idx]) == bt["k"][k - 1]["z"]
                assert list(tr[k, idx]) == bt["k"][k - 1]["transitions"]
        batch.close()


def test_evd_fit_device_vs_oracle(ctx):
    from oracle import binding as ob
    rng = np.random.RandomState(7)
    for n, mu, beta in ((100, 6.0, 1.6), (1000, 5.0, 2.0), (37, 8.0, 0.7)):
        x = np.float32(mu + beta * rng.gumbel(size=n)).astype(np.float64)
        rc_o, mu_o, lam_o = ob.evd_fit(x)
        rc_d, mu_d, lam_d = ctx.evd_fit(x)
        assert rc_o == rc_d == 1
        assert close(mu_d, mu_o, 1e-9) and close(lam_d, lam_o, 1e-9)
    # pathological input: the fit must fail the same way (all samples without HSS)
    x = np.full(50, -1.0)
    assert ob.evd_fit(x)[0] == ctx.evd_fit(x)[0]


def test_errors_and_empty(ctx):
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    batch = api.Batch(ctx, [], api.default_params()).run()
    assert batch.n == 0
    batch.close()
    rows = [AlnRow("a", "ATGGCTAAA"), AlnRow("b", "ATGGCAAAA"), AlnRow("c", "ATGGCTAAG")]
    ragged = [AlnRow("a", "ATGGCTAAA"), AlnRow("b", "ATGGCAAA"), AlnRow("c", "ATGGCTAAG")]
    good = AlnBlock(rows, "ok", "(a:0.1,b:0.1,c:0.1);", 2.0)
    # a block that cannot be scored is left out with a status and a reason; the rest of the batch is scored
    # (the reference's driver prints "Skipping alignment. Failed to build ML tree." and goes on, RNAcode.c:153-156)
    bad = [AlnBlock(rows, "x", "(a:0.1,b:0.1,zzz:0.1);", 2.0), AlnBlock(rows, "x", "(a:0.1,b:0.1", 2.0),
           AlnBlock(ragged, "x", "(a:0.1,b:0.1,c:0.1);", 2.0), AlnBlock(rows, "x", None, None)]
    batch = api.Batch(ctx, [good] + bad + [good], api.default_params(sampleN=64)).run()
    assert batch.status(0) == api.RC_OK and batch.status(5) == api.RC_OK
    for i in range(1, 5):
        assert batch.status(i) == api.RC_ERR_ARG and batch.block_error(i)
        assert (batch.maxScores(i) == -1.0).all()
    assert "zzz" in batch.block_error(1) or "not found" in batch.block_error(1)
    np.testing.assert_array_equal(batch.maxScores(0), batch.maxScores(5))
    solo = api.Batch(ctx, [good], api.default_params(sampleN=64)).run()
    np.testing.assert_array_equal(batch.maxScores(0), solo.maxScores(0))
    assert batch.scoreAln(5) == solo.scoreAln(0)
    solo.close(); batch.close()
    with pytest.raises(api.RnacodeError):
        api.Batch(ctx, [good], api.default_params(blosum=45))
    # more columns than RC_MAX_COLS: a status, not a crash (the reference's breakMAF.pl splits such blocks first)
    long_rows = [AlnRow(n, ("ACGTTGCA" * 8750)[:70000]) for n in "abc"]
    batch = api.Batch(ctx, [AlnBlock(long_rows, "long", "(a:0.1,b:0.1,c:0.1);", 2.0), good], api.default_params(sampleN=64)).run()
    assert batch.status(0) == api.RC_ERR_UNSUPPORTED and "columns" in batch.block_error(0) and batch.status(1) == api.RC_OK
    batch.close()


def test_oracle_parity_on_fresh_seeds(ctx):
    """Same seeded inputs through HIP path and oracle, on blocks and seeds no golden covers."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = synth_blocks(3, 7, 90, seed=11) + synth_blocks(2, 3, 33, seed=12) + synth_blocks(1, 9, 150, seed=13)
    blocks = [b.upper() for b in blocks]
    n, seed = 130, 1234567  # not a multiple of 64: exercises the ragged last wavefront
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed)).run()
    assert batch.clamped() == 0
    for i, b in enumerate(blocks):
        rows = [r.seq for r in b.rows]
        names = [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), seed)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores))
        got = sorted(batch.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            for k in ("strand", "frame", "startSite", "endSite", "start", "end", "startGenomic", "endGenomic"):
                assert g[k] == w[k]
            assert np.float32(g["score"]) == np.float32(w["score"])
            assert close_p(g["pvalue"], w["pvalue"])
        rc, mu, lam = batch.getExtremeValuePars(i)
        assert rc == res.evd_rc
        if rc == 1:
            assert close(mu, res.mu) and close(lam, res.lam)
    batch.close()


def test_full_size_properties(ctx):
    """BASELINE config 3 shape at full sample count (6 seqs x 120 cols, n = 1000): properties
    that need no oracle at this size -- batch-composition invariance, idempotence, sharding
    invariance of the per-sample maxima (seed = seed_base + sample index), plus an oracle check
    on a bounded sub-sample."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = [b.upper() for b in synth_blocks(24, 6, 120, seed=1)]
    p = api.default_params(sampleN=1000, seed_base=42)
    big = api.Batch(ctx, blocks, p).run()
    allmax = big.maxScores_all()
    assert allmax.shape == (24, 1000)
    assert np.isfinite(allmax).all() and ((allmax > 0) | (allmax == -1.0)).all()
    # idempotence
    big.run()
    np.testing.assert_array_equal(allmax, big.maxScores_all())
    # a block scored alone gives the same row
    solo = api.Batch(ctx, [blocks[5]], p).run()
    np.testing.assert_array_equal(solo.maxScores(0), allmax[5])
    solo.close()
    # sharding by samples: the second half of the samples is seeds seed_base+500.. of a 500-sample run
    half = api.Batch(ctx, [blocks[7]], api.default_params(sampleN=500, seed_base=42 + 500)).run()
    np.testing.assert_array_equal(half.maxScores(0), allmax[7, 500:])
    half.close()
    # oracle on a bounded sub-sample (first 64 samples of two blocks)
    for i in (0, 23):
        b = blocks[i]
        rows = [r.seq for r in b.rows]
        names = [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(64), 42)
        np.testing.assert_array_equal(allmax[i, :64], np.float32(res.maxScores))
    big.close()


def test_cli_end_to_end_matches_reference_listing(tmp_path):
    """File in, listing out (section 8f-1): MAF/ClustalW reader -> stream of GPU sub-batches -> printResults
    formats, against the reference's own printResults output for the same seeds."""
    from rnacode_amd import cli
    from rnacode_amd.synth import to_maf
    for name, variant, flags in (("genomic_preprocessed_n100", "tabular", ["-t"]),
                                 ("genomic_preprocessed_n100", "tabular", ["-t", "--sub-blocks", "5"]),   # seven sub-batches through the stream
                                 ("genomic_preprocessed_n100", "tabular_best_region", ["-t", "-r", "--sub-blocks", "1"]),
                                 ("coding_maf_n100", "gtf", ["-g"])):
        doc = load_golden(name)
        blocks = [block_from_golden(e) for e in doc["blocks"]]
        maf = tmp_path / "in.maf"
        maf.write_text(to_maf(blocks))
        side = tmp_path / "trees.tsv"
        side.write_text("".join("-\n" if "skipped" in e["ref"] else f"{e['ref']['tree']}\t{e['ref']['kappa']!r}\n"
                                for e in doc["blocks"]))
        outp = tmp_path / "out.txt"
        rc = cli.main([str(maf), "--trees", str(side), "-n", str(doc["samples"]), "--seed-base", str(doc["seed_base"]),
                       "-o", str(outp), *flags])
        assert rc == 0
        assert outp.read_text() == doc["print"][variant]


def test_native_cli_matches_reference_listing_and_python_cli(tmp_path):
    """rnacode_amd/rnacode_hip (rc_cli.cpp: C++ on the public C-ABI only -- readers, GPU tree fit, rc_stream_*, printResults):
    the reference's own listings byte for byte, and the default table equal to the Python driver's (footer timing aside)."""
    import os
    import subprocess

    from conftest import ROOT
    from rnacode_amd import cli
    from rnacode_amd.synth import to_maf
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for name, variant, flags in (("genomic_preprocessed_n100", "tabular", ["-t"]),
                                 ("genomic_preprocessed_n100", "tabular", ["-t", "--sub-blocks", "5"]),
                                 ("genomic_preprocessed_n100", "tabular_best_region", ["-t", "-r", "--sub-blocks", "1"]),
                                 ("coding_maf_n100", "gtf", ["-g"]),
                                 ("genomic_preprocessed_n100", None, [])):
        doc = load_golden(name)
        blocks = [block_from_golden(e) for e in doc["blocks"]]
        maf = tmp_path / "in.maf"
        maf.write_text(to_maf(blocks))
        side = tmp_path / "trees.tsv"
        side.write_text("".join("-\n" if "skipped" in e["ref"] else f"{e['ref']['tree']}\t{e['ref']['kappa']!r}\n"
                                for e in doc["blocks"]))
        outp = tmp_path / "out.txt"
        common = [str(maf), "--trees", str(side), "-n", str(doc["samples"]), "--seed-base", str(doc["seed_base"])]
        r = subprocess.run([exe, *common, "-o", str(outp), *flags], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        if variant is not None:
            assert outp.read_text() == doc["print"][variant]
        else:   # default table: against the Python driver, without the footer's seconds
            pyo = tmp_path / "py.txt"
            assert cli.main([*common, "-o", str(pyo)]) == 0
            strip = lambda t: [l for l in t.splitlines() if "alignment(s) scored in" not in l]
            assert strip(outp.read_text()) == strip(pyo.read_text())
            assert "Skipping alignment. Too short." in r.stderr
    # ClustalW input, trees fitted on the GPU by the driver itself
    doc = load_golden("coding_aln_n100")
    b = block_from_golden(doc["blocks"][0])
    aln = tmp_path / "in.aln"
    aln.write_text("CLUSTAL W (1.83) multiple sequence alignment\n\n" + "".join(f"{r.name:<40s} {r.seq}\n" for r in b.rows) + "\n")
    r = subprocess.run([exe, str(aln), "-n", "100", "-t"], capture_output=True, text=True, timeout=300)
    pyo = tmp_path / "py2.txt"
    assert r.returncode == 0 and cli.main([str(aln), "-n", "100", "-t", "-o", str(pyo)]) == 0
    assert r.stdout == pyo.read_text() and r.stdout.count("\n") >= 2
    # --limit (pruneAln): both drivers drop the rows first and fit the trees on what is left -- same listing as for a file
    # that never held the other rows
    doc = load_golden("genomic_preprocessed_n100")
    blocks = [block_from_golden(e) for e in doc["blocks"]]
    names = sorted({r.name.split(".")[0] for b in blocks for r in b.rows})
    keep = names[: max(4, len(names) - 2)]
    maf = tmp_path / "all.maf"
    maf.write_text(to_maf(blocks))
    import copy
    pruned = []
    for b in blocks:
        rows = [r for r in b.rows if any(r.name.startswith(k) for k in keep)]
        if rows:
            nb = copy.copy(b)
            nb.rows = rows
            pruned.append(nb)
    pm = tmp_path / "pruned.maf"
    pm.write_text(to_maf(pruned))
    got = subprocess.run([exe, str(maf), "-l", ",".join(keep), "-n", "100", "-t"], capture_output=True, text=True, timeout=300)
    want = subprocess.run([exe, str(pm), "-n", "100", "-t"], capture_output=True, text=True, timeout=300)
    assert got.returncode == 0 and want.returncode == 0, got.stderr + want.stderr
    assert got.stdout == want.stdout and got.stdout.count("\n") >= 2
    pyo = tmp_path / "py3.txt"
    assert cli.main([str(maf), "--limit", ",".join(keep), "-n", "100", "-t", "-o", str(pyo)]) == 0
    assert pyo.read_text() == got.stdout


def test_extreme_shapes_against_oracle(ctx):
    """Maximum row count of the kernels (N = 32), a long block (1500 columns) and the smallest
    legal block, same seeds through the HIP path and the oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    cases = [(synth_blocks(1, 32, 48, seed=21), 70), (synth_blocks(1, 5, 1500, seed=22), 64), (synth_blocks(1, 3, 3, seed=23, gaps=False), 64)]
    for blocks, n in cases:
        blocks = [b.upper() for b in blocks]
        batch = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=7)).run()
        b = blocks[0]
        rows = [r.seq for r in b.rows]
        names = [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), 7)
        np.testing.assert_array_equal(batch.maxScores(0), np.float32(res.maxScores))
        got = sorted(batch.scoreAln(0), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
               [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
        batch.close()
    # one row too many for the kernels: a clear per-block status, not a wrong answer
    big = [b.upper() for b in synth_blocks(1, 501, 12, seed=24, gaps=False)]
    batch = api.Batch(ctx, big, api.default_params()).run()
    assert batch.status(0) == api.RC_ERR_UNSUPPORTED and batch.block_error(0)
    batch.close()


def test_block_without_two_of_the_nucleotides_takes_the_references_nan_semantics(ctx):
    """A block of C and T only (found by tools/soak_parity.py, 1 of 1.2 million random blocks): HKY85 with two zero frequencies makes
    every background model NaN (score.c:204-244), every sigma NaN, and the reference's results then depend on how its MAX macro
    (score.h:28: x > y ? x : y) treats a NaN -- a NaN in y goes through, one in x is dropped -- where v_max_f32 returns the operand
    that is a number.  k_prep_lut flags such blocks (kFlagNan), the native-block kernels and the EXACT k_null instantiations then
    use the macro and the reference's operand order (ref_max): per-sample maxima, the HSS table and the fit's verdict equal the
    oracle's, in the batch the soak had the block in, alone, and with positive and default gap parameters."""
    import json
    import os
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    d = json.load(open(os.path.join(os.path.dirname(__file__), "data", "soak_hss_positive_delta.json")))

    def block(e):
        rows = [AlnRow(n, s) for n, s in zip(e["names"], e["rows"])]
        rows[0].start, rows[0].length = e["start"], e["length"]
        return AlnBlock(rows, "r", e["tree"], e["kappa"])

    at = next(i for i, e in enumerate(d["batch"]) if e["rows"] == d["rows"] and e["tree"] == d["tree"])
    for pars in ({"Delta": d["pars"]["Delta"]}, {}):
        hp = api.default_params(sampleN=130, seed_base=d["seed"], **pars)
        p = ob.default_params(130)
        for k, v in pars.items():
            setattr(p, k, v)
        res = ob.run_block(d["rows"], d["names"], d["start"], d["length"], d["tree"], d["kappa"], p, d["seed"])
        assert not np.isfinite(res.models[1]["scores"][0])            # the premise: NaN models
        for blocks, i in (([block(e) for e in d["batch"]], at), ([block(d)], 0)):
            b = api.Batch(ctx, blocks, hp).run()
            np.testing.assert_array_equal(b.maxScores(i), np.float32(res.maxScores))
            got, want = sorted(b.scoreAln(i), key=hss_key), sorted(res.hss, key=hss_key)
            assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
                   [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]
            assert b.getExtremeValuePars(i)[0] == res.evd_rc
            b.close()
    # the same with more than 64 rows (k_generic_dp: every cell through the written-out recurrence) and with 40 (EXACT only)
    from rnacode_amd.synth import synth_block
    for n_rows, cols in ((66, 30), (40, 33)):
        blk = synth_block(np.random.RandomState(77), n_rows, cols, index=0, gaps=True).upper()
        for r in blk.rows:
            r.seq = r.seq.replace("A", "C").replace("G", "T")
        rows, names = [r.seq for r in blk.rows], [r.name for r in blk.rows]
        p = ob.default_params(70)
        res = ob.run_block(rows, names, blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa, p, 9)
        assert not np.isfinite(res.models[1]["scores"][0])
        b = api.Batch(ctx, [blk], api.default_params(sampleN=70, seed_base=9)).run()
        np.testing.assert_array_equal(b.maxScores(0), np.float32(res.maxScores))
        got, want = sorted(b.scoreAln(0), key=hss_key), sorted(res.hss, key=hss_key)
        assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
               [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]
        assert b.getExtremeValuePars(0)[0] == res.evd_rc
        b.close()


def test_wide_blocks_against_oracle(ctx):
    """More than 32 rows (46-/100-way MAFs; the reference takes up to 500, rnaz_utils.h:7): N = 33, 48 and 64 run the
    wide instantiations (second z word, node states in more words, 13 code words per site).  Same seeds through the HIP
    path and the oracle: models, native S cell by cell, HSS, per-sample maxima, fit -- alone and mixed with narrow
    blocks in one batch and one stream."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    wide = [synth_blocks(1, 33, 45, seed=41)[0], synth_blocks(1, 48, 60, seed=42)[0], synth_blocks(1, 64, 36, seed=43)[0]]
    blocks = [b.upper() for b in wide + synth_blocks(2, 6, 60, seed=44)]
    n, seed = 70, 11
    p = api.default_params(sampleN=n, seed_base=seed)
    batch = api.Batch(ctx, blocks, p).run()
    results = []
    for i, b in enumerate(blocks):
        rows, names = [r.seq for r in b.rows], [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), seed)
        results.append(res)
        assert batch.status(i) == api.RC_OK, batch.block_error(i)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores))
        got = sorted(batch.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
               [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
        rc, mu, lam = batch.getExtremeValuePars(i)
        assert rc == res.evd_rc
        if rc == 1:
            assert close(mu, res.mu) and close(lam, res.lam)
        fwd, _ = batch.getModels(i)
        want_models = ob.get_models(b.tree, rows, names, b.kappa, 62)
        for g, w in zip(fwd, want_models):
            np.testing.assert_array_equal(np.float32(g["scores"]), np.float32(list(w.scores)))
    # native S of the 48-row block, one frame per strand, cell by cell
    b = blocks[1]
    rows, names = [r.seq for r in b.rows], [r.name for r in b.rows]
    op = ob.default_params(n)
    for strand, srows in ((0, rows), (1, ob.rev_aln(rows))):
        models = ob.get_models(b.tree, srows, names, b.kappa, 62)
        S = ob.score_matrix(srows, models, op)
        L = b.ref_len
        got = batch.native_S(1, strand, 1)
        sites = (L - 1) // 3
        for a in range(sites):
            for j in range(a, sites):
                assert got[a, j] == np.float32(S[3 * a + 1 + 1][3 * j + 3 + 1]), (strand, a, j)
    assert batch.clamped() == 0
    batch.close()
    # the same through a stream of two-block sub-batches
    m = api.Marshalled(blocks)
    m.set_trees()
    at = 0
    for sb in api.score_stream(ctx, m, p, 2, depth=2):
        for i in range(sb.n):
            np.testing.assert_array_equal(sb.maxScores(i), np.float32(results[at + i].maxScores))
        at += sb.n
        sb.close()
    assert at == len(blocks)
    # backtrack of a wide block (second z word on the host walk)
    hs = sorted(batch_hss_for_backtrack(ctx, blocks[2], p), key=hss_key)
    assert hs is not None


def test_fits_at_the_noise_floor_of_exp_follow_the_reference(ctx):
    """Three fits tools/soak_parity.py could not match (4 of 3 million blocks): one or two samples, maxima nearly equal, the root of
    Lawless 4.1.6 at lambda = 140 and 53 124 -- where exp(-lambda x) is subnormal, f is known to about its tolerance and which Newton
    iterate passes |f| < 1e-5 (or none before the 100th: then the fit fails, extreme_fit.c:205-211) hangs on the last bit of exp();
    and (round 6) 129 maxima of two values, on which Newton takes thirty-four steps through lambda = 130 along a path only the reference's
    own order of additions leads: a fit on more than 128 samples that has not converged after twelve steps is replayed in that order.
    The fit kernels compute exp by the host C library's algorithm and variant (rc_refexp.h: checked bit for bit against this image's
    glibc on 32 million arguments per variant): verdict, mu and lambda as the reference's floats."""
    import json
    import os
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    for name in ("soak_fit_two_samples.json", "soak_fit_one_sample.json", "soak_fit_two_values.json"):
        d = json.load(open(os.path.join(os.path.dirname(__file__), "data", name)))
        rows = [AlnRow(n, s) for n, s in zip(d["names"], d["rows"])]
        rows[0].start, rows[0].length = d["start"], d["length"]
        blk = AlnBlock(rows, "r", d["tree"], d["kappa"])
        pars = dict(d["pars"])
        p = ob.default_params(d["n_samples"])
        for k, v in pars.items():
            setattr(p, k, v)
        res = ob.run_block(d["rows"], d["names"], d["start"], d["length"], d["tree"], d["kappa"], p, d["seed"])
        b = api.Batch(ctx, [blk], api.default_params(sampleN=d["n_samples"], seed_base=d["seed"], **pars)).run()
        np.testing.assert_array_equal(b.maxScores(0), np.float32(res.maxScores))
        rc, mu, lam = b.getExtremeValuePars(0)
        assert rc == res.evd_rc == 1
        assert np.float32(lam) == np.float32(res.lam) and np.float32(mu) == np.float32(res.mu), (mu, res.mu, lam, res.lam)
        b.close()


_ORACLE_FIT_CHILD = r"""
import ctypes, json, math, os, sys
sys.path.insert(0, sys.argv[1])
from oracle import binding as ob
out = []
for name in sys.argv[2:]:
    d = json.load(open(os.path.join(sys.argv[1], "tests", "data", name)))
    p = ob.default_params(d["n_samples"])
    for k, v in d["pars"].items():
        setattr(p, k, v)
    r = ob.run_block(d["rows"], d["names"], d["start"], d["length"], d["tree"], d["kappa"], p, d["seed"])
    out.append(dict(rc=r.evd_rc, mu=r.mu, lam=r.lam, maxScores=[float(x) for x in r.maxScores]))
libm = ctypes.CDLL("libm.so.6"); libm.exp.restype = ctypes.c_double; libm.exp.argtypes = [ctypes.c_double]
print(json.dumps(dict(results=out, probe=[libm.exp(-x).hex() for x in (700.123, 300.7, 0.3331, 17.25, 123.456, 531.9)])))
"""


@pytest.mark.parametrize("variant", ["generic", "fused"])
def test_fits_at_the_noise_floor_follow_the_reference_under_either_exp_variant(variant, monkeypatch):
    """The reference's fit follows ITS host's exp(), and glibc's x86-64 build has two (rc_refexp.h): the oracle runs in a child process whose C
    library is held to the generic variant by GLIBC_TUNABLES, or left to pick the fused one; the library is told which by RC_FIT_EXP and says so
    (rc_ctx_fit_exp_mode).  On the two soak blocks whose fit hangs on the last bit of exp: the oracle's verdict, mu and lambda as floats.
    (Where the two variants differ the oracle's own results differ between the two runs; where this CPU has no FMA the fused run is the generic one
    again and is compared as such.)"""
    import json
    import os
    import subprocess
    import sys
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    names = ["soak_fit_two_samples.json", "soak_fit_one_sample.json", "soak_fit_two_values.json"]
    env = dict(os.environ)
    env.pop("GLIBC_TUNABLES", None)
    if variant == "generic":
        env["GLIBC_TUNABLES"] = "glibc.cpu.hwcaps=-FMA,-FMA4,-AVX2_Usable,-AVX2"
    doc = json.loads(subprocess.check_output([sys.executable, "-c", _ORACLE_FIT_CHILD, ROOT] + names, env=env, text=True))
    # which variant did the child's exp turn out to be?  (six probes where the two differ or agree: compared against the same probes in this process
    # only to report it; the device side is forced to `variant` either way and must then agree with a child that really ran it)
    monkeypatch.setenv("RC_FIT_EXP", variant)
    ctx = api.Context(0)
    assert ctx.fit_exp_mode == variant
    for name, want in zip(names, doc["results"]):
        d = json.load(open(os.path.join(os.path.dirname(__file__), "data", name)))
        rows = [AlnRow(n, s) for n, s in zip(d["names"], d["rows"])]
        rows[0].start, rows[0].length = d["start"], d["length"]
        blk = AlnBlock(rows, "r", d["tree"], d["kappa"])
        b = api.Batch(ctx, [blk], api.default_params(sampleN=d["n_samples"], seed_base=d["seed"], **d["pars"])).run()
        np.testing.assert_array_equal(b.maxScores(0), np.float32(want["maxScores"]))
        rc, mu, lam = b.getExtremeValuePars(0)
        assert rc == want["rc"], (name, variant, rc, want["rc"])
        if rc == 1:
            assert np.float32(lam) == np.float32(want["lam"]) and np.float32(mu) == np.float32(want["mu"]), (name, variant, mu, want["mu"], lam, want["lam"])
        b.close()
    ctx.close()


def test_the_fit_exp_mode_is_reported_and_a_bad_override_is_refused(monkeypatch):
    from rnacode_amd import api
    c = api.Context(0)
    assert c.fit_exp_mode in ("generic", "fused", "device")
    c.close()
    monkeypatch.setenv("RC_FIT_EXP", "device")
    c = api.Context(0)
    assert c.fit_exp_mode == "device"
    c.close()
    monkeypatch.setenv("RC_FIT_EXP", "glibc")
    with pytest.raises(api.RnacodeError):
        api.Context(0)


def test_fits_on_few_samples_sum_in_the_references_order(ctx):
    """Fits on <= 128 samples evaluate Lawless 4.1.6 with the four sums taken in the reference's order, i = 0 .. n-1 (rc_kernels.hip,
    lawless416_serial; extreme_fit.c:50-58): rc_evd_fit on doubles against the oracle's EVDMaxLikelyFit, lambda to the last bit of a
    binary64 (mu to an ulp: the device's log) on near-degenerate and ordinary samples alike (the library's exp being this host's, rc_refexp.h)."""
    from oracle import binding as ob
    from rnacode_amd import api
    if ctx.fit_exp_mode == "device":
        pytest.skip("this host's exp() is not one rc_refexp.h reproduces")
    rng = np.random.RandomState(77)
    for n in (1, 2, 3, 17, 64, 65, 100, 128):
        for spread in (1e-4, 0.02, 3.0):
            x = 5.0 + spread * rng.rand(n)
            rc, mu, lam = ctx.evd_fit(x)
            wrc, wmu, wlam = ob.evd_fit(x)
            assert rc == wrc, (n, spread, rc, wrc)
            if rc == 1:
                assert lam == wlam and abs(mu - wmu) <= 1e-15 * max(1.0, abs(wmu)), (n, spread, mu, wmu, lam, wlam)   # (mu goes through the device's log: one ulp)


def test_blocks_with_one_purine_and_one_pyrimidine_follow_the_reference(ctx, monkeypatch):
    """A block of G and T only (or A and C, A and T, C and G): with both other frequencies zero Seq-Gen's alphaR is 0 / 0, every branch matrix
    NaN, `r > P` false for every draw (evolve.c:167-175) and every simulated residue state 0 -- the null samples are all-A alignments with the
    block's gaps.  The library's integer thresholds took NaN for "every draw exceeds it" until a soak found this block in round 6
    (rc_host.cpp, threshold_of).  tests/data/sim_nan_branch_matrix.json holds the UNMODIFIED reference's per-sample maxima, HSS and fit verdicts
    (tests/data/make_sim_nan_fixture.py); here both simulations must reproduce them: the per-row-count kernels' and, with every block sent to the generic
    kernels, k_generic_sim's (which the tiled kernels share)."""
    import json
    import os
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    d = json.load(open(os.path.join(os.path.dirname(__file__), "data", "sim_nan_branch_matrix.json")))
    blocks = []
    for e in d["cases"]:
        rows = [AlnRow(n, s) for n, s in zip(e["names"], e["rows"])]
        rows[0].start, rows[0].length = e["start"], e["length"]
        blocks.append(AlnBlock(rows, e["name"], e["tree"], e["kappa"]))
    e0 = d["cases"][0]
    hp = api.default_params(sampleN=e0["n_samples"], seed_base=e0["seed"])

    def check(c):
        for batch in ([blocks], [[b] for b in blocks]):
            at = 0
            for bl in batch:
                b = api.Batch(c, bl, hp).run()
                for i in range(len(bl)):
                    e = d["cases"][at + i]
                    np.testing.assert_array_equal(b.maxScores(i), np.float32(e["maxScores"]), err_msg=e["name"])
                    got, want = sorted(b.scoreAln(i), key=hss_key), sorted(e["hss"], key=hss_key)
                    assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
                           [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want], e["name"]
                    assert b.getExtremeValuePars(i)[0] == e["evd_rc"]
                at += len(bl)
                b.close()

    check(ctx)
    monkeypatch.setenv("RC_GENERIC_MIN_ROWS", "3")
    c2 = api.Context(0)
    check(c2)
    c2.close()


def test_tiled_kernels_equal_the_oracle(monkeypatch):
    """k_tiled_dp<KT> (rc_null_tiled.h; blocks of 32..64 rows and up to 200..250 reference residues by default, here from 17 rows on and any length):
    the sequences besides the reference in tiles of 12..15, a cell's sum carried from tile to tile in score.c's order.  Per-sample maxima, HSS
    tables and fit verdicts against the oracle for every tile size and 2..5 tiles, rows longer than the row buffer's LDS tail (300 columns), 64
    rows (the second z word, a fifth tile), a batch of one block (strand x frame parts over workgroups), a block without two of the nucleotides
    (NaN tables: the second launch with the reference's MAX macro), Delta >= 0 (that instantiation for every block), --stop-early, and a block
    with omega > 0, which must not take these kernels (their padded sequences need a gap state never above 0)."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    monkeypatch.setenv("RC_TILED_MIN_ROWS", "17")
    monkeypatch.setenv("RC_TILED_MAX_L", "100000")
    ctx = api.Context(0)
    rng = np.random.RandomState(606)

    def check(blocks, n, seed, expect_tiled=True, **pars):
        hp = api.default_params(sampleN=n, seed_base=seed, **pars)
        b = api.Batch(ctx, blocks, hp).run()
        assert ("k_tiled_dp" in b.null_kernel()) == expect_tiled, b.null_kernel()
        for i, blk in enumerate(blocks):
            p = ob.default_params(n)
            for k, v in pars.items():
                setattr(p, k, v)
            res = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa, p, seed)
            if not pars.get("stopEarly") or res.evd_rc == 1:
                np.testing.assert_array_equal(b.maxScores(i), np.float32(res.maxScores), err_msg=f"block {i}: {blk.n} x {blk.cols}")
            got, want = sorted(b.scoreAln(i), key=hss_key), sorted(res.hss, key=hss_key)
            assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
                   [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]
            rc, mu, lam = b.getExtremeValuePars(i)
            assert rc == res.evd_rc and (rc != 1 or (close_p(mu, res.mu) and close_p(lam, res.lam)))
        b.close()

    shapes = [(17, 60), (25, 45), (27, 90), (29, 45), (31, 60), (34, 150), (40, 90), (46, 45), (47, 60), (61, 30), (64, 45), (20, 300)]
    blocks = [synth_block(rng, n, c, index=i, gaps=True).upper() for i, (n, c) in enumerate(shapes)]
    check(blocks, 130, 77)
    check(blocks[6:7], 200, 5)                               # one block: its strand x frame parts over workgroups
    check(blocks[:6], 100, 9, Delta=1.5)                     # Delta >= 0: the maximum with Delta counts
    check(blocks[2:8], 100, 11, stopEarly=1, cutoff=0.2)
    check(blocks[4:6], 64, 13, expect_tiled=False, omega=0.5)   # omega > 0: not these kernels
    nan = synth_block(np.random.RandomState(77), 40, 33, index=0, gaps=True).upper()
    for r in nan.rows:
        r.seq = r.seq.replace("A", "C").replace("G", "T")
    check([blocks[3], nan, blocks[9]], 130, 21)
    ctx.close()


def test_tiled_kernels_on_blocks_of_more_than_64_rows_equal_the_oracle(monkeypatch):
    """The tiled kernels past the per-row-count kernels' range (RC_TILED_MAX_ROWS; rc_device.h, block_class): 65 rows (the sixth tile of 13 starts
    in the third z word: 2 x 64 bits hold 64 sequences), 70 and 99 rows (simulation chunks of whole tiles: 60 or 52 sequences, not 64), 130 rows
    (three chunks), NaN tables among them; the native block of such a block goes through k_native_dp_generic.  Per-sample maxima, HSS tables
    and fit verdicts against the oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    monkeypatch.setenv("RC_TILED_MAX_ROWS", "200")
    monkeypatch.setenv("RC_TILED_WIDE_MAX_L", "100000")
    ctx = api.Context(0)
    rng = np.random.RandomState(707)
    shapes = [(65, 45), (70, 60), (99, 36), (130, 30), (66, 120)]
    blocks = [synth_block(rng, n, c, index=i, gaps=True).upper() for i, (n, c) in enumerate(shapes)]
    nan = synth_block(np.random.RandomState(78), 80, 33, index=9, gaps=True).upper()
    for r in nan.rows:
        r.seq = r.seq.replace("A", "C").replace("G", "T")
    blocks.append(nan)
    for n, seed, pars in ((130, 31, {}), (100, 33, {"Delta": 1.5})):
        hp = api.default_params(sampleN=n, seed_base=seed, **pars)
        b = api.Batch(ctx, blocks, hp).run()
        assert "k_tiled_dp" in b.null_kernel(), b.null_kernel()
        for i, blk in enumerate(blocks):
            p = ob.default_params(n)
            for k, v in pars.items():
                setattr(p, k, v)
            res = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa, p, seed)
            np.testing.assert_array_equal(b.maxScores(i), np.float32(res.maxScores), err_msg=f"block {i}: {blk.n} x {blk.cols}")
            got, want = sorted(b.scoreAln(i), key=hss_key), sorted(res.hss, key=hss_key)
            assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
                   [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]
            rc, mu, lam = b.getExtremeValuePars(i)
            assert rc == res.evd_rc and (rc != 1 or (close_p(mu, res.mu) and close_p(lam, res.lam)))
        b.close()
    ctx.close()


def test_wide_blocks_in_several_rounds_equal_one_round(ctx):
    """k_generic_sim leaves the codes of a ROUND of items in the scratch for k_generic_dp (rc_schedule.cpp): with a scratch of 24 GB
    a batch of a few wide blocks is one round, with 8 MB (RC_GENERIC_SCRATCH_MB, read when a context is created) every block's
    sample groups are a round of their own -- queues reset between rounds, the last round short.  Same maxima, HSS and fits."""
    import os
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = [b.upper() for b in synth_blocks(5, 70, 60, seed=61) + synth_blocks(2, 130, 45, seed=62) + synth_blocks(3, 6, 60, seed=63)]
    p = api.default_params(sampleN=200, seed_base=5)
    one = api.Batch(ctx, blocks, p).run()
    want = (one.maxScores_all().copy(), one.fits().copy(), one.scoreAln_all())
    launches_one = one.timing()[1]["null"]
    one.close()
    os.environ["RC_GENERIC_SCRATCH_MB"] = "8"
    try:
        small = api.Context(0)
    finally:
        del os.environ["RC_GENERIC_SCRATCH_MB"]
    b = api.Batch(small, blocks, p).run()
    assert b.timing()[1]["null"] > launches_one          # more rounds = more launches
    np.testing.assert_array_equal(b.maxScores_all(), want[0])
    np.testing.assert_array_equal(b.fits()[:, 0], want[1][:, 0])
    np.testing.assert_allclose(b.fits()[:, 1:3], want[1][:, 1:3], rtol=1e-6)
    assert [[hss_key(h) for h in x] for x in b.scoreAln_all()] == [[hss_key(h) for h in x] for x in want[2]]
    b.close()
    small.close()


@pytest.mark.parametrize("kernels", ["tiled", "generic"])
def test_blocks_beyond_64_rows_against_oracle(monkeypatch, kernels):
    """65..500 rows (the reference's own limit, rnaz_utils.h:7; 100-way genome alignments): k_generic_sim with k_tiled_dp (the default for
    these shapes) or, with the tiled kernels held to 64 rows, k_generic_dp, which keeps the per-sequence states in global memory as
    k_native_dp_generic does.  Same seeds through the HIP path and the oracle, blocks of different widths in ONE launch, mixed with
    narrow ones, as one batch, as a stream and with --stop-early; backtrack of a wide block."""
    from oracle import binding as ob
    if kernels == "generic":
        monkeypatch.setenv("RC_TILED_MAX_ROWS", "64")
    from rnacode_amd import api as _api
    ctx = _api.Context(0)
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    shapes = [(65, 36, 51), (100, 45, 52), (200, 24, 53), (500, 12, 54)]
    wide = [synth_blocks(1, n, cols, seed=sd, gaps=(n < 500))[0] for n, cols, sd in shapes]
    blocks = [b.upper() for b in [wide[0]] + synth_blocks(2, 6, 60, seed=55) + wide[1:] + synth_blocks(1, 40, 30, seed=56)]
    n, seed = 70, 13
    p = api.default_params(sampleN=n, seed_base=seed)
    batch = api.Batch(ctx, blocks, p).run()
    assert ("k_tiled_dp" if kernels == "tiled" else "k_generic_dp") in batch.null_kernel(), batch.null_kernel()
    results = []
    for i, b in enumerate(blocks):
        rows, names = [r.seq for r in b.rows], [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), seed)
        results.append(res)
        assert batch.status(i) == api.RC_OK, batch.block_error(i)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores))
        got = sorted(batch.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
               [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
        rc, mu, lam = batch.getExtremeValuePars(i)
        assert rc == res.evd_rc
        if rc == 1:
            assert close(mu, res.mu) and close(lam, res.lam)
    assert batch.clamped() == 0
    batch.close()
    m = api.Marshalled(blocks)
    m.set_trees()
    at = 0
    for sb in api.score_stream(ctx, m, p, 3, depth=2):
        for i in range(sb.n):
            np.testing.assert_array_equal(sb.maxScores(i), np.float32(results[at + i].maxScores))
        at += sb.n
        sb.close()
    assert at == len(blocks)
    assert batch_hss_for_backtrack(ctx, blocks[3], p) is not None     # 100 rows: four z words on the host walk
    # the built-in tree estimator: up to 64 tips on the GPU, wider blocks on host threads inside the same call
    lnl = []
    sub = [blocks[0], blocks[1], blocks[3]]          # 65 rows, 6 rows, 100 rows
    fits = api.fit_trees(sub, ctx=ctx, lnl=lnl)
    assert all(f is not None for f in fits)
    for b, (nwk, kappa), l in zip(sub, fits, lnl):
        assert nwk.count(",") == b.n - 1 and 0.5 < kappa < 20
        assert abs(l - api.tree_lnl(b, nwk, kappa)) < 0.05 + 1e-4 * abs(l)   # lnL of the printed (%f-rounded) tree, host evaluation
    host = api.fit_trees(sub)
    assert [_newick_parts(h[0])[0] for h in host] == [_newick_parts(f[0])[0] for f in fits]
    refit = [b for b in sub]
    for b, f in zip(refit, fits):
        b.tree, b.kappa = f
    again = api.Batch(ctx, refit, p).run()
    assert [again.status(i) for i in range(3)] == [api.RC_OK] * 3
    again.close()
    ctx.close()


def batch_hss_for_backtrack(ctx, block, p):
    """Backtrack the best HSS of `block` on the device and with the oracle; returns the HSS list."""
    from oracle import binding as ob
    from rnacode_amd import api
    batch = api.Batch(ctx, [block], p).run()
    hss = batch.scoreAln(0)
    if hss:
        h = hss[0]
        strand = 0 if h["strand"] == "+" else 1
        st, z, tr = batch.backtrack(0, strand, h["start"], h["end"])
        rows, names = [r.seq for r in block.rows], [r.name for r in block.rows]
        op = ob.default_params(p.sampleN)
        m = ob.get_models(block.tree, rows, names, block.kappa, 62)
        rrows = ob.rev_aln(rows)
        mr = ob.get_models(block.tree, rrows, names, block.kappa, 62)
        _, skf, skr = ob.score_aln(rows, block.rows[0].start, block.rows[0].length, m, mr, op, want_sk=True)
        want = ob.backtrack(h["start"], h["end"], skf if strand == 0 else skr, rows if strand == 0 else rrows, op)
        ws, wz, wt = want
        idx = list(range(h["end"], h["start"] + 1, -3))
        assert len(idx) >= 3
        for k in range(1, block.n):
            assert list(st[k, idx]) == list(ws[k, idx]) and list(z[k, idx]) == list(wz[k, idx]) and list(tr[k, idx]) == list(wt[k, idx])
    batch.close()
    return hss


def test_built_in_trees_give_the_same_calls_as_phyml_trees(ctx):
    """Section 8f-2 end to end: with trees from rc_fit_tree instead of PhyML the strong HSS keep their
    coordinates and their scores move by well under a percent (the null samples differ: another tree
    orientation means other draws, so p-values are compared on the log scale only)."""
    from rnacode_amd import api, cli
    doc = load_golden("genomic_preprocessed_n1000")
    blocks = [block_from_golden(e) for e in doc["blocks"]]
    trees = cli.fit_trees(blocks)
    mine = []
    for b, t in zip(blocks, trees):
        nb = block_from_golden({"input": dict(id=b.block_id, rows=[dict(name=r.name, seq=r.seq, start=r.start, length=r.length,
                               strand=r.strand, full_length=r.full_length) for r in b.rows]), "ref": {}})
        nb.tree, nb.kappa = t if t is not None else ("(a:0.1,b:0.1,c:0.1);", 2.0)
        mine.append(nb)
    batch = api.Batch(ctx, mine, api.default_params(sampleN=1000, seed_base=42)).run()
    checked = 0
    for i, e in enumerate(doc["blocks"]):
        if "skipped" in e["ref"]:
            continue
        want = sorted(e["ref"]["hss"], key=hss_key)
        got = sorted(batch.scoreAln(i), key=hss_key)
        for w in want:
            if w["score"] < 15.0:
                break
            g = next((h for h in got if (h["strand"], h["frame"], h["start"], h["end"]) == (w["strand"], w["frame"], w["start"], w["end"])), None)
            assert g is not None, (i, w)
            assert abs(g["score"] - w["score"]) <= 0.01 * w["score"]
            if w["pvalue"] > 0:
                assert abs(np.log10(max(g["pvalue"], 1e-300)) - np.log10(w["pvalue"])) < 1.5
            checked += 1
    assert checked >= 20
    batch.close()


def test_cli_without_sidecar_fits_its_own_trees(tmp_path):
    """The stand-alone driver end to end with nothing but an alignment file (section 8f-1 + 8f-2)."""
    from rnacode_amd import cli
    from rnacode_amd.synth import to_maf
    doc = load_golden("coding_maf_n100")
    maf = tmp_path / "in.maf"
    maf.write_text(to_maf([block_from_golden(e) for e in doc["blocks"]]))
    outp = tmp_path / "out.txt"
    assert cli.main([str(maf), "-n", "200", "-t", "-o", str(outp)]) == 0
    rows = [l.split("\t") for l in outp.read_text().splitlines()]
    top = sorted(doc["blocks"][0]["ref"]["hss"], key=hss_key)[0]
    assert rows[0][1] == top["strand"] and int(rows[0][2]) == top["frame"] + 1
    assert int(rows[0][7]) == top["startGenomic"] and int(rows[0][8]) == top["endGenomic"]
    assert abs(float(rows[0][9]) - top["score"]) < 0.01 * top["score"] and float(rows[0][10]) < 1e-4


def test_sample_count_edges_and_config5_shape(ctx):
    """n = 1, 63, 65 (ragged wavefronts) against the oracle; BASELINE config 5 shape (12 seqs x 300
    cols) at its full n = 10000 through the seed-sharding property (first 1000 samples == an n=1000 run)."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    (b,) = [x.upper() for x in synth_blocks(1, 6, 90, seed=31)]
    rows = [r.seq for r in b.rows]
    names = [r.name for r in b.rows]
    for n in (1, 63, 65):
        batch = api.Batch(ctx, [b], api.default_params(sampleN=n, seed_base=5)).run()
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), 5)
        np.testing.assert_array_equal(batch.maxScores(0), np.float32(res.maxScores))
        rc, mu, lam = batch.getExtremeValuePars(0)
        assert rc == res.evd_rc
        if rc == 1:
            assert close(mu, res.mu) and close(lam, res.lam)
        batch.close()
    big = [x.upper() for x in synth_blocks(2, 12, 300, seed=2)]
    full = api.Batch(ctx, big, api.default_params(sampleN=10000, seed_base=42)).run()
    part = api.Batch(ctx, big, api.default_params(sampleN=1000, seed_base=42)).run()
    np.testing.assert_array_equal(full.maxScores_all()[:, :1000], part.maxScores_all())
    doc = load_golden("synth_12x300_n40")   # the same two blocks, first 40 samples pinned by the reference
    for i, e in enumerate(doc["blocks"]):
        np.testing.assert_array_equal(full.maxScores(i)[:40], np.float32(e["ref"]["maxScores"]))
    # 1.3e8 draws: a handful fall beyond the float-rounded cumulative frequencies (the reference reads out
    # of bounds there, seqgen/evolve.c:173; we clamp and count) -- about 1e-7 per root draw
    assert full.clamped() <= 100 and part.clamped() <= 10
    full.close(); part.close()


def test_duplicate_row_names_are_rejected(ctx):
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    rows = [AlnRow("a", "ATGGCTAAA"), AlnRow("a", "ATGGCAAAA"), AlnRow("c", "ATGGCTAAG")]
    batch = api.Batch(ctx, [AlnBlock(rows, "x", "(a:0.1,a:0.1,c:0.1);", 2.0)], api.default_params())
    assert batch.status(0) == api.RC_ERR_ARG and "duplicate" in batch.block_error(0)
    batch.close()


def test_block_too_large_for_the_device_gets_its_own_status(ctx):
    """A block whose MT19937 streams (4 bytes x tree nodes x columns x samples) would not fit a quarter of the device memory is left
    out with RC_ERR_UNSUPPORTED and a reason; the other blocks of the batch are scored (include/rnacode_hip.h, bounds on a block)."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    big = synth_blocks(1, 500, 20000, seed=11)[0].upper()        # 998 nodes x 20 000 columns x 1024 samples x 4 B = 82 GB
    small = synth_blocks(2, 5, 60, seed=12)
    blocks = [small[0].upper(), big, small[1].upper()]
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000)).run()
    assert batch.status(1) == api.RC_ERR_UNSUPPORTED and "too large for the device" in batch.block_error(1)
    assert batch.status(0) == api.RC_OK and batch.status(2) == api.RC_OK
    alone = api.Batch(ctx, [blocks[0], blocks[2]], api.default_params(sampleN=1000)).run()
    np.testing.assert_array_equal(batch.maxScores(0), alone.maxScores(0))
    np.testing.assert_array_equal(batch.maxScores(2), alone.maxScores(1))
    batch.close(); alone.close()


def _newick_parts(nwk):
    import re
    return re.sub(r":[0-9.]+", "", nwk), [float(x) for x in re.findall(r":([0-9.]+)", nwk)]


def test_device_tree_fit_matches_the_host_fit(ctx):
    """rc_fit_trees_device (one wavefront per block) runs the same estimator as rc_fit_trees (host
    threads) -- rc_tree_core.h -- so topologies are identical and lengths / kappa / lnL agree to the
    optimiser's tolerance; checked on the real-data goldens (IUPAC codes, gaps, 3..~12 rows), on
    synthetic blocks, on wide shapes (32, 48 and 64 rows) and on blocks the driver skips."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = []
    for name in ("genomic_preprocessed_n100", "coding_maf_n100", "noncoding_maf_n100", "edge_cases_n50"):
        blocks += [block_from_golden(e) for e in load_golden(name)["blocks"]]
    blocks += synth_blocks(40, 6, 120, seed=3) + synth_blocks(3, 32, 90, seed=4) + synth_blocks(2, 3, 30, seed=5) + synth_blocks(2, 12, 600, seed=6)
    blocks += synth_blocks(1, 48, 90, seed=7) + synth_blocks(1, 64, 60, seed=8)   # wide blocks: up to RC_MAX_ROWS tips
    lnl = []
    dev = api.fit_trees(blocks, ctx=ctx, lnl=lnl)
    host = api.fit_trees(blocks)
    fitted = 0
    for b, d, h, l in zip(blocks, dev, host, lnl):
        assert (d is None) == (h is None)
        if d is None:
            continue
        (td, ld), (th, lh) = _newick_parts(d[0]), _newick_parts(h[0])
        assert td == th, b.block_id
        assert max(abs(x - y) for x, y in zip(ld, lh)) < 2e-4
        assert abs(d[1] - h[1]) <= 1e-3 * h[1]
        assert abs(l - api.tree_lnl(b, d[0], d[1])) < 1e-2   # lnL of the printed (%f-rounded) tree, host evaluation
        fitted += 1
    assert fitted >= 60


def test_tie_rich_and_degenerate_blocks_against_oracle(ctx):
    """Blocks built to hit the rare branches of the path: stretches where every non-reference row is gapped
    (all sigma = 0, so consecutive entries of the score matrix are exactly equal and getHSS's 'equal score, longer
    segment' rule decides, score.c:953-954), frame-shifting gap lengths, and zero-length branches (identity
    transition rows: the degenerate cumulative rows of evolve.c:292).  maxScores and the HSS table must still be
    bit-identical to the oracle's."""
    import re
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(77)
    blocks = []
    for i in range(96):
        n = int(rng.randint(3, 7))
        cols = int(rng.randint(45, 121))
        b = synth_block(rng, n, cols, index=i, gaps=bool(i % 2), mean_branch=(0.02, 0.1, 0.3)[i % 3]).upper()
        rows = [bytearray(r.seq.encode()) for r in b.rows]
        for _ in range(int(rng.randint(1, 4))):          # common gap stretches in all non-reference rows
            length = int((3, 6, 9, 12, 4, 7)[rng.randint(6)])
            pos = int(rng.randint(0, cols - length))
            for r in range(1, n):
                rows[r][pos:pos + length] = b"-" * length
        for r, row in zip(b.rows, rows):
            r.seq = row.decode()
            r.length = sum(1 for ch in r.seq if ch != "-")
        if i % 4 == 0:                                   # one or two zero-length branches
            lens = list(re.finditer(r":[0-9.]+", b.tree))
            for m in [lens[j] for j in rng.choice(len(lens), size=min(2, len(lens)), replace=False)]:
                b.tree = b.tree[:m.start()] + ":0.000000" + b.tree[m.end():]
        blocks.append(b)
    n_samples, seed = 150, 991
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n_samples, seed_base=seed)).run()
    ob.tie_replacements(reset=True)
    for i, b in enumerate(blocks):
        rows = [r.seq for r in b.rows]
        names = [r.name for r in b.rows]
        res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n_samples), seed)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores), err_msg=f"block {i}")
        got = sorted(batch.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            for k in ("strand", "frame", "startSite", "endSite", "start", "end"):
                assert g[k] == w[k]
            assert np.float32(g["score"]) == np.float32(w["score"])
    assert ob.tie_replacements() > 1000   # the construction does exercise the tie rule (native and null alignments)
    batch.close()


@pytest.mark.parametrize("name", ["eps_coding_aln_n100", "eps_genomic_preprocessed_n100"])
def test_eps_plots_with_device_backtrack(ctx, name):
    """Section 8f-4: the reference's hss-<n>.eps files byte for byte, with the state paths of the left extension, the
    segment and the right extension coming from rc_batch_backtrack."""
    from rnacode_amd import api
    from test_eps_cpu import collect_plots
    doc = load_golden(name)
    base = load_golden(doc["base"])
    _, blocks, batch = run_golden(ctx, base)
    index = {b.block_id: i for i, b in enumerate(blocks)}

    def make_backtracker(block, tree, kappa):
        i = index[block.block_id]
        return lambda strand, lo, hi: batch.backtrack(i, 0 if strand == "+" else 1, lo, hi)
    names, texts = collect_plots(doc, base, make_backtracker, set(doc["files"]))
    assert names == doc["names"]
    for k, want in doc["files"].items():
        assert texts[k] == want, k
    batch.close()


def test_cli_writes_eps_files(tmp_path):
    from rnacode_amd import cli
    from rnacode_amd.synth import to_maf
    doc = load_golden("coding_maf_n100")
    maf = tmp_path / "in.maf"
    maf.write_text(to_maf([block_from_golden(e) for e in doc["blocks"]]))
    side = tmp_path / "trees.tsv"
    side.write_text("".join(f"{e['ref']['tree']}\t{e['ref']['kappa']!r}\n" if "skipped" not in e["ref"] else "-\n" for e in doc["blocks"]))
    assert cli.main([str(maf), "-n", "100", "--trees", str(side), "-o", str(tmp_path / "out.txt"), "-e", "-d", str(tmp_path / "plots")]) == 0
    made = sorted(p.name for p in (tmp_path / "plots").iterdir())
    assert made and all(n.startswith("hss-") and n.endswith(".eps") for n in made)
    assert (tmp_path / "plots" / made[0]).read_text().startswith("%!PS-Adobe-3.0 EPSF-3.0\n")


def test_mixed_batch_against_oracle(ctx):
    """One batch with 3..12 rows and 30..200 columns per block (several row-count classes, blocks of very different
    length in each, two sample groups so that six of the eight per-XCD queues start empty and steal): per-sample
    maxima bit-identical to the oracle for every block."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(2024)
    blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(30, 201)), index=i, gaps=True).upper() for i in range(120)]
    n, seed = 70, 31337
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed)).run()
    for i, b in enumerate(blocks):
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa,
                           ob.default_params(n), seed)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores), err_msg=f"block {i} ({b.n} x {b.cols})")
        rc, mu, lam = batch.getExtremeValuePars(i)
        assert rc == res.evd_rc and (rc != 1 or (close(mu, res.mu) and close(lam, res.lam)))
    batch.close()


@pytest.mark.parametrize("pars", [dict(Delta=0.25, Omega=-4.0, omega=-2.0),          # Delta >= 0: MAX3 with Delta matters
                                  dict(Delta=-10.0, Omega=-4.0, omega=-2.0, stopPenalty_k=-3.0e30),   # outside the proven division range
                                  dict(Delta=-1.0e-30, Omega=-4.0, omega=-2.0)])
def test_exact_division_instantiation_against_oracle(ctx, pars):
    """Parameter sets the fast kernel is not proven for are routed to k_null<.., EXACT = true> (plain '/', MAX3 with
    Delta kept): same bit-exact bar against the oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = [b.upper() for b in synth_blocks(3, 5, 75, seed=41) + synth_blocks(2, 9, 60, seed=42)]
    n, seed = 100, 2718
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed, **pars)).run()
    for i, b in enumerate(blocks):
        p = ob.default_params(n)
        for k, v in pars.items():
            setattr(p, k, v)
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, p, seed)
        np.testing.assert_array_equal(batch.maxScores(i), np.float32(res.maxScores), err_msg=f"block {i}")
        got = sorted(batch.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(g["strand"], g["frame"], g["start"], g["end"]) for g in got] == [(w["strand"], w["frame"], w["start"], w["end"]) for w in want]
        assert all(np.float32(g["score"]) == np.float32(w["score"]) for g, w in zip(got, want))
    batch.close()


def test_device_tree_fit_in_several_launches(ctx, monkeypatch):
    """A scratch cap smaller than the batch needs makes rc_fit_trees_device go in several launches: same trees."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = synth_blocks(60, 6, 90, seed=8) + synth_blocks(5, 14, 200, seed=9)
    whole = api.fit_trees(blocks, ctx=ctx)
    monkeypatch.setenv("RC_TREE_SCRATCH_BYTES", str(300 * 1024))
    parts = api.fit_trees(blocks, ctx=ctx)
    assert parts == whole and all(t is not None for t in whole)


def test_bulk_hss_fetch_equals_per_block_fetch(ctx):
    """rc_batch_hss_all against rc_batch_hss block by block (scored, skipped and empty blocks mixed)."""
    doc = load_golden("genomic_preprocessed_n100")
    _, blocks, batch = run_golden(ctx, doc)
    everything = batch.scoreAln_all()
    assert len(everything) == len(blocks)
    from rnacode_amd import api
    for i in range(len(blocks)):
        want = batch.scoreAln(i) if batch.status(i) == api.RC_OK else []
        assert everything[i] == want
    batch.close()


@pytest.mark.parametrize("name", ["reference_ub_block_70x30", "reference_ub_block_6x120"])
def test_draw_beyond_the_last_cumulative_frequency_is_clamped_and_counted(ctx, name):
    """The blocks of tests/data/reference_ub_block_*.json.gz (the reference reads out of bounds on one sample; 70 rows: the generic kernel,
    split into its six strand x frame parts; 6 rows: k_null): the HIP path counts the draw once (rc_batch_clamped == 1) and gives the oracle's
    maxima -- equal to the reference's on all the other samples."""
    from test_oracle_golden import load_ub_block
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    doc = load_ub_block(name)
    rows = [AlnRow(n, s, doc["start"], doc["length"], "+", 10000000) for n, s in zip(doc["names"], doc["rows"])]
    b = AlnBlock(rows, "ub", doc["tree"], doc["kappa"])
    batch = api.Batch(ctx, [b], api.default_params(sampleN=doc["samples"], seed_base=doc["seed_base"])).run()
    assert batch.clamped() == 1
    got = batch.maxScores(0)
    np.testing.assert_array_equal(got, np.float32(doc["oracle_maxScores"]))
    assert list(np.flatnonzero(got != np.float32(doc["ref_maxScores"]))) == doc["ub_samples"]
    batch.close()


def test_malformed_blocks_among_good_ones_do_not_disturb_them(ctx):
    """Fuzz of the block intake: among valid blocks, blocks with broken trees (unknown or duplicate tips, garbage text, negative or huge
    or non-numeric branch lengths, no tree), absurd kappa, bytes outside the alphabet, all-gap rows, too few rows or columns.  No crash;
    every block either scores or carries a status with a reason (RNAcode.c:142-156 skips such blocks and goes on); the valid blocks'
    results are those of a batch without the bad ones."""
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    from rnacode_amd.synth import synth_blocks
    rng = np.random.RandomState(99)
    good = [b.upper() for b in synth_blocks(12, 5, 60, seed=21)]

    def mutate(b, kind):
        rows = [AlnRow(r.name, r.seq, r.start, r.length, r.strand, r.full_length) for r in b.rows]
        tree, kappa = b.tree, b.kappa
        if kind == 0:
            tree = tree.replace(rows[1].name, "nobody")
        elif kind == 1:
            tree = tree.replace(rows[1].name, rows[2].name)          # duplicate tip
        elif kind == 2:
            tree = "".join(chr(int(x)) for x in rng.randint(33, 127, 40))
        elif kind == 3:
            tree = tree.replace(":0.", ":-0.", 1)
        elif kind == 4:
            tree = tree.replace(":0.", ":1e30", 1)
        elif kind == 5:
            tree = tree.replace(":0.", ":abc", 1)
        elif kind == 6:
            tree = tree[: len(tree) // 2]
        elif kind == 7:
            kappa = float(rng.choice([0.0, -3.0, 1e30, float("nan"), float("inf")]))
        elif kind == 8:
            s = bytearray(rows[2].seq.encode())
            for q in rng.randint(0, len(s), 6):
                s[q] = int(rng.choice([0x01, 0x7f, ord("*"), ord("."), ord("?"), ord("x")]))
            rows[2].seq = s.decode("latin1")
        elif kind == 9:
            rows[3].seq = "-" * len(rows[3].seq)
        elif kind == 10:
            rows = rows[:2]
        elif kind == 11:
            rows = [AlnRow(r.name, r.seq[:2], r.start, 2, r.strand, r.full_length) for r in rows]
        elif kind == 12:
            rows[0].seq = "-" * len(rows[0].seq)                      # reference row without residues
        elif kind == 13:
            tree = "(" * 5000 + tree
        return AlnBlock(rows, "m%d" % kind, tree, kappa)

    batch_blocks, is_good = [], []
    for i, g in enumerate(good):
        batch_blocks.append(g); is_good.append(True)
        for kind in rng.choice(14, size=3, replace=False):
            batch_blocks.append(mutate(good[(i + 1) % len(good)], int(kind))); is_good.append(False)
    p = api.default_params(sampleN=64, seed_base=5)
    batch = api.Batch(ctx, batch_blocks, p).run()
    clean = api.Batch(ctx, good, p).run()
    gi = 0
    statuses = set()
    for i, ok in enumerate(is_good):
        st = batch.status(i)
        if ok:
            assert st == api.RC_OK
            np.testing.assert_array_equal(batch.maxScores(i), clean.maxScores(gi))
            assert batch.scoreAln(i) == clean.scoreAln(gi)
            assert batch.getExtremeValuePars(i) == clean.getExtremeValuePars(gi)
            gi += 1
        else:
            statuses.add(st)
            if st != api.RC_OK:
                assert st in (api.RC_ERR_ARG, api.RC_ERR_SKIP, api.RC_ERR_UNSUPPORTED)
                assert (batch.maxScores(i) == -1.0).all()
                assert st == api.RC_ERR_SKIP or batch.block_error(i)
            else:   # scored all the same (e.g. stray bytes count as unknown residues): finite results
                assert np.isfinite(batch.maxScores(i)).all()
    assert api.RC_ERR_ARG in statuses and api.RC_ERR_SKIP in statuses
    batch.close(); clean.close()


def _write_inputs(tmp_path, name):
    """(command-line head, golden) for a reference-scored fixture: its blocks as a file, its PhyML trees as the sidecar."""
    from rnacode_amd.synth import to_maf
    doc = load_golden(name)
    blocks = [block_from_golden(e) for e in doc["blocks"]]
    side = tmp_path / f"{name}.trees.tsv"
    side.write_text("".join("-\n" if "skipped" in e["ref"] else f"{e['ref']['tree']}\t{e['ref']['kappa']!r}\n" for e in doc["blocks"]))
    if all(r.start == 0 and r.length == 0 for b in blocks for r in b.rows):   # a ClustalW input (coding.aln): one block
        path = tmp_path / f"{name}.aln"
        path.write_text("CLUSTAL W (1.83) multiple sequence alignment\n\n" + "".join(f"{r.name:<40s} {r.seq}\n" for r in blocks[0].rows) + "\n")
    else:
        path = tmp_path / f"{name}.maf"
        path.write_text(to_maf(blocks))
    return [str(path), "--trees", str(side), "-n", str(doc["samples"]), "--seed-base", str(doc["seed_base"])], doc


@pytest.mark.parametrize("name", ["eps_coding_aln_n100", "eps_genomic_preprocessed_n100"])
def test_native_cli_draws_the_reference_plots(tmp_path, name):
    """rnacode_hip --eps (rc_eps.h on rc_batch_backtrack + rc_code_tables): the hss-<n>.eps files the unmodified reference wrote
    (src/postscript.c compiled into the harness: goldens eps_*), byte for byte and under the same counters; the whole directory equal
    to the Python driver's; and the same again from two contexts (--gpus 2 --devices 0,0)."""
    import os
    import subprocess

    from conftest import ROOT
    from rnacode_amd import cli
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    doc = load_golden(name)
    head, _ = _write_inputs(tmp_path, doc["base"])
    runs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2", "--devices", "0,0", "--sub-blocks", "7"])):
        d = tmp_path / f"plots_{tag}"
        r = subprocess.run([exe, *head, "-e", "-i", repr(doc["eps_cutoff"]), "-d", str(d), "-o", str(tmp_path / f"list_{tag}.txt"), *extra],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        runs[tag] = {p.name: p.read_text() for p in d.iterdir()}
    counter = lambda n: int(n[4:-4])
    assert sorted(runs["one"], key=counter) == doc["names"]            # which HSS get a plot, and under which counter
    for k, want in doc["files"].items():
        assert runs["one"][k] == want, k
    assert runs["two"] == runs["one"]
    pyd = tmp_path / "plots_py"
    assert cli.main([*head, "-e", "-i", repr(doc["eps_cutoff"]), "-d", str(pyd), "-o", str(tmp_path / "list_py.txt")]) == 0
    assert {p.name: p.read_text() for p in pyd.iterdir()} == runs["one"]
    strip = lambda t: [l for l in t.splitlines() if "alignment(s) scored in" not in l]
    assert strip((tmp_path / "list_py.txt").read_text()) == strip((tmp_path / "list_one.txt").read_text())


def test_native_cli_on_several_gpus_writes_the_single_gpu_listing(tmp_path):
    """rnacode_hip --gpus N inside one process (SURVEY.md section 8e without a collective library): N contexts on N host threads.
    On the one GPU of a test box, --gpus 2 / 3 --devices 0,0[,0] must write the single-context listing byte for byte
    (a) for a many-block input, whose sub-batches are dealt to the contexts in turn -- trees fitted by each context for its own
    sub-batches, one writer, the hit counter running across blocks in input order -- and (b) for few-block inputs, where every context
    simulates a slice of every block's sample range (seed_base + first sample) and the slices are fitted after they meet on the host:
    the shipped coding.aln and three blocks of genomic-preprocessed.maf, with --stop-early and with plots as well."""
    import os
    import subprocess

    from conftest import ROOT
    from rnacode_amd.synth import synth_block, to_maf
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")

    def run(args, **kw):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900, **kw)
        assert r.returncode == 0, r.stderr
        return r

    # (a) 700 blocks of 3..7 rows, a few of them unscorable (two rows; too short)
    rng = np.random.RandomState(99)
    blocks = [synth_block(rng, int(rng.randint(3, 8)), int(rng.randint(30, 120)), index=i, gaps=True).upper() for i in range(700)]
    for k in (13, 350):
        blocks[k].rows = blocks[k].rows[:2]
    maf = tmp_path / "many.maf"
    maf.write_text(to_maf(blocks))
    for fmt in (["-t"], ["-g"], ["-t", "-b"], ["-t", "-r", "-p", "0.5"]):
        one = run([str(maf), "-n", "200", *fmt])
        more = [["--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--sub-blocks", "41"]] if fmt == ["-t"] else []   # (what an 8-GPU node runs: eight contexts, one writer)
        for extra in [["--gpus", "2", "--devices", "0,0"], ["--gpus", "3", "--devices", "0,0,0", "--sub-blocks", "37"]] + more:
            got = run([str(maf), "-n", "200", *fmt, *extra])
            assert got.stdout == one.stdout and one.stdout.count("\n") > 100, (fmt, extra)
            assert sorted(got.stderr.splitlines()) == sorted(one.stderr.splitlines())   # the "Skipping alignment" notes
    one = run([str(maf), "-n", "200"])
    two = run([str(maf), "-n", "200", "--gpus", "2", "--devices", "0,0"], env=dict(os.environ, RC_CLI_TIMES="1"))
    strip = lambda t: [l for l in t.splitlines() if "alignment(s) scored in" not in l]
    assert strip(one.stdout) == strip(two.stdout)
    assert "2 GPUs" in two.stderr and "dealt to the GPUs in turn" in two.stderr
    # (b) few blocks: the sample range is split
    head, doc = _write_inputs(tmp_path, "coding_aln_n100")
    one = run([*head, "-t"])
    for extra in (["--gpus", "2", "--devices", "0,0"],):
        got = run([*head, "-t", *extra], env=dict(os.environ, RC_CLI_TIMES="1"))
        assert got.stdout == one.stdout and "sample ranges over the GPUs" in got.stderr
    base = load_golden("genomic_preprocessed_n1000")
    three = dict(base, blocks=[e for e in base["blocks"] if "skipped" not in e["ref"]][:3])
    from rnacode_amd.synth import to_maf as _to_maf
    blocks3 = [block_from_golden(e) for e in three["blocks"]]
    m3 = tmp_path / "three.maf"
    m3.write_text(_to_maf(blocks3))
    s3 = tmp_path / "three.tsv"
    s3.write_text("".join(f"{e['ref']['tree']}\t{e['ref']['kappa']!r}\n" for e in three["blocks"]))
    for flags in (["-t"], ["-t", "-s", "-p", "0.05"], ["-t", "-e", "-i", "0.9"]):
        outs = []
        for tag, extra in (("one", []), ("two", ["--gpus", "2", "--devices", "0,0"]), ("four", ["--gpus", "4", "--devices", "0,0,0,0"])):
            d = tmp_path / f"p_{tag}_{len(flags)}"
            r = run([str(m3), "--trees", str(s3), "-n", "1000", "--seed-base", "42", *flags, *(["-d", str(d)] if "-e" in flags else []), *extra])
            outs.append((r.stdout, {p.name: p.read_text() for p in d.iterdir()} if "-e" in flags else None))
        assert outs[0] == outs[1] == outs[2], flags
        assert outs[0][0].count("\n") >= 3
    # ... and the p-values of that listing are the reference's (the fit of the gathered rows, through rc_evd_fit)
    r = run([str(m3), "--trees", str(s3), "-n", "1000", "--seed-base", "42", "-t", "--gpus", "2", "--devices", "0,0"])
    want = {}
    for e in three["blocks"]:
        for h in e["ref"]["hss"]:
            want[(e["input"]["rows"][0]["name"], h["strand"], h["frame"] + 1, h["startSite"] + 1, h["endSite"] + 1, h["startGenomic"])] = h
    seen = 0
    for line in r.stdout.splitlines():
        f = line.split("\t")
        h = want[(f[6], f[1], int(f[2]), int(f[4]), int(f[5]), int(f[7]))]
        assert abs(float(f[9]) - h["score"]) < 1e-3 and close_p(float(f[10]), float("%.3e" % h["pvalue"]) if h["pvalue"] < 0.001 else float("%.3f" % h["pvalue"]), rel=2e-3)
        seen += 1
    assert seen >= 3
