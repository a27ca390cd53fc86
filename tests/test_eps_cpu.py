"""EPS plots against the files the reference's colorAln wrote (src/postscript.c, compiled in
place into the harness; goldens eps_*.json.gz).  Here the backtracked state paths come from the oracle, so the test
pins rnacode_amd/eps.py and report.print_results' plot selection on the CPU; tests/test_gpu_parity.py runs the same
comparison with rc_batch_backtrack."""
import io

import pytest

from conftest import load_golden
from helpers import block_from_golden
from oracle import binding as ob
from rnacode_amd import eps, report


def oracle_backtracker(block, tree, kappa):
    p = ob.default_params(100)
    rows = [r.seq for r in block.rows]
    names = [r.name for r in block.rows]
    m = ob.get_models(tree, rows, names, kappa, p.blosum)
    rrows = ob.rev_aln(rows)
    mr = ob.get_models(tree, rrows, names, kappa, p.blosum)
    _, skf, skr = ob.score_aln(rows, block.rows[0].start, block.rows[0].length, m, mr, p, want_sk=True)

    def bt(strand, b, i):
        return ob.backtrack(b, i, skf if strand == "+" else skr, rows if strand == "+" else rrows, p)
    return bt


def collect_plots(doc, base, make_backtracker, wanted):
    """Drive print_results like the reference's driver and render the plots whose names are in `wanted`."""
    st = report.ReportState()
    names, texts = [], {}
    for e in base["blocks"]:
        if "skipped" in e["ref"]:
            continue
        block = block_from_golden(e)
        bt = [None]

        def hook(counter, h, block=block, e=e, bt=bt):
            name = f"hss-{counter}.eps"
            names.append(name)
            if name in wanted:
                if bt[0] is None:
                    bt[0] = make_backtracker(block, e["ref"]["tree"], e["ref"]["kappa"])
                texts[name] = eps.color_aln(block, h, bt[0])
        report.print_results(io.StringIO(), 0, e["ref"]["hss"], e["input"]["rows"][0]["name"], st, eps=hook,
                             eps_cutoff=doc["eps_cutoff"])
    return names, texts


@pytest.mark.parametrize("name", ["eps_coding_aln_n100", "eps_genomic_preprocessed_n100"])
def test_eps_matches_reference_coloraln(name):
    doc = load_golden(name)
    base = load_golden(doc["base"])
    names, texts = collect_plots(doc, base, oracle_backtracker, set(doc["files"]))
    assert names == doc["names"]            # which HSS get a plot, and under which counter
    for k, want in doc["files"].items():
        got = texts[k]
        if got != want:                     # first differing line, for the failure message
            for ln, (a, b) in enumerate(zip(got.splitlines(), want.splitlines())):
                assert a == b, f"{k} line {ln + 1}"
        assert got == want
