"""MAF pre-splitting against the reference's scripts/breakMAF.pl: its own example pair
(examples/genomic.maf -> examples/genomic-preprocessed.maf) and synthetic long blocks with other parameters,
byte for byte; and the split blocks go through the reader unchanged."""
import io

from conftest import load_golden
from rnacode_amd import breakmaf


def test_break_maf_matches_the_perl_script():
    for case in load_golden("breakmaf_cases")["cases"]:
        kw = {}
        if case["args"]:
            a = case["args"]
            kw = dict(max_length=int(a[a.index("--maxLength") + 1]), desired_length=int(a[a.index("--desiredLength") + 1]))
        assert breakmaf.break_maf(case["input"], **kw) == case["output"], case["name"]


def test_split_blocks_read_back(tmp_path):
    from rnacode_amd.alnio import read_alignment_file
    case = load_golden("breakmaf_cases")["cases"][0]
    p = tmp_path / "pre.maf"
    p.write_text(breakmaf.break_maf(case["input"]))
    blocks = list(read_alignment_file(str(p)))
    assert len(blocks) == 34 and max(b.cols for b in blocks) <= 400
    q = tmp_path / "orig.maf"
    q.write_text(case["input"])
    orig = list(read_alignment_file(str(q)))
    # columns are conserved: the pieces of a block concatenate to the block
    assert sum(b.cols for b in blocks) == sum(b.cols for b in orig)
