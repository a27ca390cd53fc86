"""Alignment block I/O for the scoring path.

Readers for the two input formats RNAcode accepts (reference: src/rnaz_utils.c:44-234,
`read_clustal` / `read_maf`) and the plain-text block format used to hand blocks to the
reference harness (oracle/ref_harness.c).  Only what the scoring path needs is kept:
row names, aligned sequences, and the reference row's genomic start/length/strand.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Iterator, List, Optional


@dataclass
class AlnRow:
    name: str
    seq: str
    start: int = 0
    length: int = 0
    strand: str = "?"
    full_length: int = 0


@dataclass
class AlnBlock:
    """One alignment block; row 0 is the reference sequence (rnaz_utils.h:12-20)."""

    rows: List[AlnRow]
    block_id: str = ""
    tree: Optional[str] = None  # Newick with branch lengths (sidecar or fitted)
    kappa: Optional[float] = None

    @property
    def n(self) -> int:
        return len(self.rows)

    @property
    def cols(self) -> int:
        return len(self.rows[0].seq) if self.rows else 0

    @property
    def ref_len(self) -> int:
        """Number of non-gap characters in row 0 (misc.c:272-289, getSeqLength)."""
        return sum(1 for c in self.rows[0].seq if c != "-")

    def upper(self) -> "AlnBlock":
        """main() upper-cases every row before anything else (RNAcode.c:121-128)."""
        rows = [AlnRow(r.name, r.seq.upper(), r.start, r.length, r.strand, r.full_length) for r in self.rows]
        return AlnBlock(rows, self.block_id, self.tree, self.kappa)


def read_maf(lines: Iterable[str]) -> Iterator[AlnBlock]:
    """Blocks of a MAF stream.  's' lines need exactly 7 fields; 'i', 'e', 'q' and comment
    lines are skipped; a block ends at the next 'a' line (rnaz_utils.c:132-234)."""
    rows: List[AlnRow] = []
    idx = 0
    for line in lines:
        f = line.split()
        if not f or f[0].startswith("#"):
            continue
        if f[0] in ("i", "e", "q"):
            continue
        if f[0] == "s":
            if len(f) != 7:
                raise ValueError("Invalid MAF format (number of fields in 's' line not correct)")
            strand = f[4][0]
            if strand not in "+-":
                raise ValueError("Invalid MAF format (strand)")
            rows.append(AlnRow(f[1], f[6], int(f[2]), int(f[3]), strand, int(f[5])))
            continue
        if f[0] == "a":
            if rows:
                _check_equal(rows)
                yield AlnBlock(rows, block_id=str(idx))
                idx += 1
            rows = []
    if rows:
        _check_equal(rows)
        yield AlnBlock(rows, block_id=str(idx))


def read_clustal(lines: Iterable[str]) -> Iterator[AlnBlock]:
    """A ClustalW file holds one block; rows get start=length=0, strand '?'
    (rnaz_utils.c:44-117)."""
    names: List[str] = []
    seqs: List[str] = []
    nn = 0
    for raw in lines:
        line = raw.rstrip("\n").rstrip("\r")
        if line.startswith("CLUSTAL"):
            if names:
                break
            continue
        if len(line) < 4 or line[0].isspace():
            nn = 0
            continue
        f = line.split()
        if len(f) < 2:
            continue
        name, seq = f[0][:99], f[1]
        if nn == len(names):
            names.append(name)
            seqs.append(seq)
        else:
            if names[nn] != name:
                raise ValueError("Inconsistent sequence names in CLUSTAL file")
            seqs[nn] += seq
        nn += 1
    if names:
        rows = [AlnRow(n, s, 0, 0, "?", 0) for n, s in zip(names, seqs)]
        _check_equal(rows)
        yield AlnBlock(rows, block_id="0")


def read_alignment_file(path: str) -> List[AlnBlock]:
    with open(path) as fh:
        lines = fh.readlines()
    first = next((l for l in lines if l.strip()), "")
    if first.startswith("CLUSTAL"):
        return list(read_clustal(lines))
    return list(read_maf(lines))


def _check_equal(rows: List[AlnRow]) -> None:
    n = len(rows[0].seq)
    if any(len(r.seq) != n for r in rows):
        raise ValueError("Sequences are of unequal length.")


def write_harness_blocks(blocks: Iterable[AlnBlock], path: str) -> None:
    """Plain-text block format read by oracle/ref_harness.c."""
    with open(path, "w") as fh:
        for b in blocks:
            fh.write(f"B {b.block_id}\n")
            for r in b.rows:
                fh.write(f"S {r.name} {r.start} {r.length} {r.strand} {r.full_length} {r.seq}\n")
            if b.tree is not None:
                fh.write(f"T {b.tree}\n")
                if b.kappa is not None:
                    fh.write(f"K {b.kappa!r}\n")
            fh.write("E\n")
