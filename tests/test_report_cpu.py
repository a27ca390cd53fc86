"""Output formats against the reference's own printResults
(src/misc.c:392-552), captured by the harness into the goldens' "print" entries."""
import io

import pytest

from conftest import load_golden
from rnacode_amd import report

VARIANTS = {
    "list": dict(fmt=0),
    "gtf": dict(fmt=1),
    "tabular": dict(fmt=2),
    "tabular_p0.05": dict(fmt=2, cutoff=0.05),
    "list_best_only": dict(fmt=0, best_only=True),
    "tabular_best_region": dict(fmt=2, best_region=True),
}


@pytest.mark.parametrize("name", ["coding_aln_n100", "noncoding_aln_n100", "coding_maf_n100", "noncoding_maf_n100",
                                  "genomic_preprocessed_n100"])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_listing_matches_reference_printresults(name, variant):
    doc = load_golden(name)
    out = io.StringIO()
    st = report.ReportState()
    for e in doc["blocks"]:
        if "skipped" in e["ref"]:
            continue
        report.print_results(out, hss=e["ref"]["hss"], ref_name=e["input"]["rows"][0]["name"], state=st, **VARIANTS[variant])
    assert out.getvalue() == doc["print"][variant]
