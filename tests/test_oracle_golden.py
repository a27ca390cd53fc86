"""The CPU restatement (oracle/) against golden vectors produced by the unmodified reference
objects (oracle/make_goldens.py).  This is what pins the oracle; the GPU parity tests then
compare the HIP path with the oracle and with the same goldens."""
import numpy as np
import pytest

from conftest import golden_names, hss_key, load_golden
from helpers import block_from_golden, close_p, param_kwargs
from oracle import binding as ob

FAST = [n for n in golden_names() if n != "genomic_preprocessed_n1000"]


def oracle_params(doc):
    kw = param_kwargs(doc)
    p = ob.default_params(kw["sampleN"], kw.get("blosum", 62), kw.get("stopEarly", 0), kw.get("cutoff", 1.0))
    for k in ("Delta", "Omega", "omega", "stopPenalty_k"):
        if k in kw:
            setattr(p, k, kw[k])
    return p


def check_block(entry, doc):
    ref = entry["ref"]
    if "skipped" in ref:
        b = block_from_golden(entry)
        assert b.n <= 2 or b.ref_len < 3
        return
    b = block_from_golden(entry)
    rows = [r.seq for r in b.rows]
    names = [r.name for r in b.rows]
    p = oracle_params(doc)
    res = ob.run_block(rows, names, b.rows[0].start, b.rows[0].length, ref["tree"], ref["kappa"], p, doc["seed_base"])
    assert res.clamped == 0
    # background models: bit-exact float32
    for got, want in ((res.models, ref["models"]), (res.modelsRev, ref["modelsRev"])):
        for g, w in zip(got, want):
            assert np.float32(g["dist"]) == np.float32(w["dist"])
            np.testing.assert_array_equal(np.float32(g["freqs"]), np.float32(w["freqs"]))
            np.testing.assert_array_equal(np.float32(g["scores"]), np.float32(w["scores"]))
    # HSS table: coordinates/frames bit-exact, scores and p-values exact float32
    got = sorted(res.hss, key=hss_key)
    want = sorted(ref["hss"], key=hss_key)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for k in ("strand", "frame", "startSite", "endSite", "start", "end", "startGenomic", "endGenomic"):
            assert g[k] == w[k], (k, g, w)
        assert np.float32(g["score"]) == np.float32(w["score"])
        if ref["evd_rc"] == 1:
            assert close_p(g["pvalue"], w["pvalue"])
        else:
            assert g["pvalue"] == 99.0
    assert res.evd_rc == ref["evd_rc"]
    if ref["maxScores"]:
        np.testing.assert_array_equal(np.float32(res.maxScores), np.float32(ref["maxScores"]))
    if ref["evd_rc"] == 1:
        assert np.float32(res.mu) == np.float32(ref["mu"])
        assert np.float32(res.lam) == np.float32(ref["lambda"])


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference(name):
    doc = load_golden(name)
    for entry in doc["blocks"]:
        check_block(entry, doc)


def test_oracle_matches_reference_n1000_subset():
    doc = load_golden("genomic_preprocessed_n1000")
    for entry in doc["blocks"][:3]:
        check_block(entry, doc)


def test_sk_digest_and_backtrack():
    """Sk (pairwise DP) digest and the backtrack path of the best HSS, against the reference."""
    for name in ("coding_aln_n100", "synth_6x120_n200", "edge_cases_n50"):
        doc = load_golden(name)
        p = oracle_params(doc)
        for entry in doc["blocks"]:
            ref = entry["ref"]
            if "skipped" in ref:
                continue
            b = block_from_golden(entry)
            rows = [r.seq for r in b.rows]
            names = [r.name for r in b.rows]
            m = ob.get_models(ref["tree"], rows, names, ref["kappa"], p.blosum)
            rrows = ob.rev_aln(rows)
            mr = ob.get_models(ref["tree"], rrows, names, ref["kappa"], p.blosum)
            hss, skf, skr = ob.score_aln(rows, b.rows[0].start, b.rows[0].length, m, mr, p, want_sk=True)
            L = b.ref_len
            for sk, key in ((skf, "sk_digest_fwd"), (skr, "sk_digest_rev")):
                s = w = 0.0
                for k in range(1, b.n):
                    for x in range(3):
                        for bb in range(1, L + 1):
                            ii = np.arange(bb + 2, L + 1, 3)
                            v = sk[k, x, bb, ii].astype(np.float64)
                            # same accumulation order as the harness digest
                            for vi, i in zip(v, ii):
                                s += vi
                                w += vi * float((k * 3 + x) * 131 + bb * 7 + i)
                assert s == ref[key][0] and w == ref[key][1]
            if "backtrack" in ref:
                bt = ref["backtrack"]
                top = sorted(ref["hss"], key=hss_key)[0]
                sk, rr = (skf, rows) if top["strand"] == "+" else (skr, rrows)
                st, z, tr = ob.backtrack(bt["b"], bt["i"], sk, rr, p)
                idx = list(range(bt["i"], bt["b"] + 1, -3))
                for k in range(1, b.n):
                    assert list(st[k, idx]) == bt["k"][k - 1]["states"]
                    assert list(z[k, idx]) == bt["k"][k - 1]["z"]
                    assert list(tr[k, idx]) == bt["k"][k - 1]["transitions"]


UB_BLOCKS = ["reference_ub_block_70x30", "reference_ub_block_6x120"]


def load_ub_block(name):
    import gzip
    import json
    import os
    from conftest import ROOT
    with gzip.open(os.path.join(ROOT, "tests", "data", name + ".json.gz"), "rb") as fh:
        return json.loads(fh.read().decode())


@pytest.mark.parametrize("name", UB_BLOCKS)
def test_reference_out_of_bounds_draw_is_the_only_divergence(name):
    """tests/data/reference_ub_block_*.json.gz: blocks on which the unmodified reference draws beyond the last cumulative frequency
    (seqgen/evolve.c:173: SetState returns 4, printed as '\\0', the branches below read out of bounds; at the root this corrupts the heap
    and with other sample counts glibc aborts the reference).  The oracle clamps to state 3 and counts: it reports
    exactly one such draw and agrees with the reference on every other sample."""
    doc = load_ub_block(name)
    res = ob.run_block(doc["rows"], doc["names"], doc["start"], doc["length"], doc["tree"], doc["kappa"], ob.default_params(doc["samples"]), doc["seed_base"])
    assert res.clamped == 1
    got, ref = np.float32(res.maxScores), np.float32(doc["ref_maxScores"])
    assert list(np.flatnonzero(got != ref)) == doc["ub_samples"]
    np.testing.assert_array_equal(got, np.float32(doc["oracle_maxScores"]))


def test_one_purine_and_one_pyrimidine_blocks_against_the_reference():
    """tests/data/sim_nan_branch_matrix.json (made by tests/data/make_sim_nan_fixture.py from the unmodified reference): blocks whose two absent
    nucleotides leave Seq-Gen with NaN branch matrices -- every simulated residue state 0.  The oracle restates evolve.c:167-175 literally."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "data", "sim_nan_branch_matrix.json")))
    for e in d["cases"]:
        res = ob.run_block(e["rows"], e["names"], e["start"], e["length"], e["tree"], e["kappa"], ob.default_params(e["n_samples"]), e["seed"])
        np.testing.assert_array_equal(np.float32(res.maxScores), np.float32(e["maxScores"]), err_msg=e["name"])
        assert res.evd_rc == e["evd_rc"] and len(res.hss) == len(e["hss"])
        sim, _ = ob.simulate_null(e["tree"], e["rows"], e["names"], res.models[0]["freqs"], e["kappa"], e["seed"])
        assert set("".join(sim)) <= {"A", "-"}, e["name"]
