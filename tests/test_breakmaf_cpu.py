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


def test_pieces_are_consistent_with_their_coordinates():
    """Property check on random blocks: every piece has at most ceil(len / int(len / desired)) columns, the pieces of a
    row concatenate to the row, and the coordinates follow the residues -- up to the script's own quirk, kept on
    purpose: a piece that begins inside a gap run of a row gets the position of the residue BEFORE the gap as its
    start (alnCol2genomePos counts letters up to and including the column, breakMAF.pl:303-310), so its start is one
    less and its length one more than the residues it holds."""
    import numpy as np
    from rnacode_amd.alnio import read_alignment_file
    from rnacode_amd.synth import synth_block, to_maf
    import math
    import os
    import tempfile
    rng = np.random.RandomState(5)
    for trial in range(12):
        cols = int(rng.randint(150, 1300))
        blk = synth_block(rng, int(rng.randint(3, 7)), cols, index=trial, gaps=True)
        # long gap runs as well, including at the start of a row
        rows = [bytearray(r.seq.encode()) for r in blk.rows]
        for r in range(1, blk.n):
            p = int(rng.randint(0, cols - 40)); rows[r][p:p + 37] = b"-" * 37
        rows[1][0:9] = b"-" * 9
        for r, row in zip(blk.rows, rows):
            r.seq = row.decode(); r.length = sum(ch != "-" for ch in r.seq)
        mx, want = int(rng.randint(100, 500)), int(rng.randint(60, 100))
        out = breakmaf.break_maf(to_maf([blk]), mx, want)
        with tempfile.NamedTemporaryFile("w", suffix=".maf", delete=False) as fh:
            fh.write(out)
        try:
            pieces = list(read_alignment_file(fh.name))
        finally:
            os.unlink(fh.name)
        if cols > mx:
            assert max(p.cols for p in pieces) <= math.ceil(cols / int(cols / want))
        else:
            assert len(pieces) == 1
        for r in range(blk.n):
            assert "".join(p.rows[r].seq for p in pieces) == blk.rows[r].seq
            pos = blk.rows[r].start
            for p in pieces:
                n_res = sum(ch != "-" for ch in p.rows[r].seq)
                quirk = 1 if (p.rows[r].seq[0] == "-" and pos > blk.rows[r].start) else 0
                if n_res:
                    assert p.rows[r].length == n_res + quirk
                    assert p.rows[r].start == pos - quirk
                pos += n_res
