"""Pre-splitting of long MAF blocks (the reference's scripts/breakMAF.pl).

    python -m rnacode_amd.breakmaf [--maxLength 400] [--desiredLength 200] < input.maf > output.maf

Blocks longer than maxLength columns are cut into ceil(length / int(length / desiredLength))-column
chunks (breakMAF.pl:36-62) with the start/length of every row recomputed from its gap pattern
(sliceAlnByColumn / alnCol2genomePos, breakMAF.pl:232-311); every block is re-written the way the
script's formatAln does it ("a score=0", source names cut to "org.chrom", srcSize 0).  The scorer's
batch path wants this for genome-wide input: it keeps blocks inside the row lengths the sigma-code
staging in LDS is sized for.  Output is byte-identical to the Perl script's (tests/test_breakmaf_cpu.py:
examples/genomic.maf -> examples/genomic-preprocessed.maf), including its quirk: a piece that begins inside a
gap run of a row starts at the residue before the gap, one position early and one residue long."""
from __future__ import annotations

import argparse
import math
import re
import sys
from typing import Dict, List, Optional


def _read_first_block(chunk: str) -> Optional[List[Dict]]:
    """readMAF (breakMAF.pl:74-123) restricted to what the caller uses: the first alignment of the chunk."""
    rows: List[Dict] = []
    lines = chunk.split("\n")
    for i, line in enumerate(lines):
        if re.match(r"\s?#", line) or re.match(r"\s?a", line):
            continue
        if re.match(r"\s?s", line):
            f = line.split()
            name, start, length, strand, seq = f[1], int(f[2]), int(f[3]), f[4], f[6]
            parts = name.split(".")
            rows.append(dict(name=name, org=parts[0], chrom=parts[1] if len(parts) > 1 else "", start=start,
                             end=start + length, seq=seq, strand=strand))
        if re.match(r"\s?$", line) and rows:
            return rows
        if i + 1 >= len(lines) and rows:
            return rows
    return rows or None


def _col_to_pos(seq: str, start: int, col: int) -> int:
    """alnCol2genomePos, breakMAF.pl:285-311."""
    seq = seq.replace(".", "-")
    if re.fullmatch(r"-+", seq):
        return start
    leading = len(seq) - len(seq.lstrip("-"))
    if col < leading:
        return start
    pos = start - 1
    for i in range(leading, col + 1):
        if seq[i:i + 1] != "-":
            pos += 1
    return pos


def _slice(rows: List[Dict], lo: int, hi: int) -> List[Dict]:
    """sliceAlnByColumn, breakMAF.pl:244-271."""
    lo = max(lo, 0)
    hi = min(hi, len(rows[0]["seq"]))
    out = []
    for r in rows:
        n = dict(r)
        n["start"] = _col_to_pos(r["seq"], r["start"], lo)
        n["end"] = _col_to_pos(r["seq"], r["start"], hi - 1) + 1
        n["seq"] = r["seq"][lo:hi]
        out.append(n)
    return out


def _format(rows: List[Dict]) -> str:
    """formatAln(..., 'maf'), breakMAF.pl:217-223."""
    out = ["a score=0\n"]
    for r in rows:
        out.append("s %s.%s %d %d %s 0 %s\n" % (r["org"], r["chrom"], r["start"], r["end"] - r["start"], r["strand"], r["seq"]))
    return "".join(out)


def break_maf(text: str, max_length: int = 400, desired_length: int = 200) -> str:
    out: List[str] = []
    for chunk in text.split("a score="):          # $/ = 'a score=' (breakMAF.pl:25)
        rows = _read_first_block("a score=" + chunk)
        if not rows:
            continue
        length = len(rows[0]["seq"])
        if length > max_length:
            n = int(length / desired_length)
            step = math.ceil(length / n)
            lo = 0
            while True:
                hi = min(lo + step, length)
                out.append(_format(_slice(rows, lo, hi)) + "\n")
                lo = hi
                if lo == length:
                    break
        else:
            out.append(_format(rows) + "\n")
    return "".join(out)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="breakmaf", description="break long MAF blocks (scripts/breakMAF.pl)")
    ap.add_argument("--maxLength", type=int, default=400, help="Break all blocks longer than that (default: 400 columns)")
    ap.add_argument("--desiredLength", type=int, default=200, help="Try to create blocks of this size (default: 200 columns)")
    a = ap.parse_args(argv)
    sys.stdout.write(break_maf(sys.stdin.read(), a.maxLength, a.desiredLength))
    return 0


if __name__ == "__main__":
    sys.exit(main())
