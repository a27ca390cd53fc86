"""Deterministic synthetic alignment blocks.

The reference ships no generator; BASELINE.json's configs 3-5 are "Synthetic MAF: B blocks x
cols x N seqs".  Per block: a random unrooted binary tree over N tips with branch lengths
~ Exp(mean 0.1) written with 6 decimals (PhyML prints %f, phyml/utilities.c:921), kappa ~
U(1.5, 4), pi ~ Dirichlet(10,10,10,10); a root sequence of `cols` sites drawn from pi is evolved
under HKY85 down the tree; then 2-4 gap runs of length {1,2,3,3,6} are cut into random rows
(row 0 included) so that all three z classes of the DP occur.  Rows are named sp<r>.chr1,
start = 1000 * block index, strand '+'.  The generating tree and kappa travel as a sidecar
(AlnBlock.tree / .kappa) so that the hot path can be measured without a tree fitter.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .alnio import AlnBlock, AlnRow

_NT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _hky_matrix(pi: np.ndarray, kappa: float, t: float) -> np.ndarray:
    """HKY85 transition probabilities for branch length t (expected substitutions/site)."""
    a, c, g, tt = pi
    r, y = a + g, c + tt
    beta = 1.0 / (2.0 * (r * y + kappa * (a * g + c * tt)))
    e1 = np.exp(-beta * t)
    e2r = np.exp(-beta * t * (1.0 + r * (kappa - 1.0)))
    e2y = np.exp(-beta * t * (1.0 + y * (kappa - 1.0)))
    P = np.empty((4, 4))
    grp = (r, y, r, y)
    e2 = (e2r, e2y, e2r, e2y)
    for i in range(4):
        for j in range(4):
            if i == j:
                P[i, j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 + (grp[j] - pi[j]) / grp[j] * e2[j]
            elif (i % 2) == (j % 2):  # transition (A<->G, C<->T)
                P[i, j] = pi[j] + pi[j] * (1.0 / grp[j] - 1.0) * e1 - pi[j] / grp[j] * e2[j]
            else:
                P[i, j] = pi[j] * (1.0 - e1)
    return P


def _random_tree(rng: np.random.RandomState, n: int):
    """Random unrooted binary topology: children lists of internal nodes; node 0 is the
    trifurcating root.  Returns (children dict, tips list)."""
    # nodes are ints; tips are ('t', r)
    children = {0: [("t", 0), ("t", 1), ("t", 2)]}
    nxt = 1
    for r in range(3, n):
        # pick a random edge (parent, slot) and split it
        edges = [(p, s) for p, ch in children.items() for s in range(len(ch))]
        p, s = edges[rng.randint(len(edges))]
        old = children[p][s]
        children[nxt] = [old, ("t", r)] if rng.randint(2) else [("t", r), old]
        children[p][s] = nxt
        nxt += 1
    return children


def synth_block(rng: np.random.RandomState, n: int, cols: int, index: int = 0, gaps: bool = True,
                mean_branch: float = 0.1) -> AlnBlock:
    children = _random_tree(rng, n)
    kappa = float(np.float32(rng.uniform(1.5, 4.0)))
    pi = rng.dirichlet([10.0] * 4)
    names = [f"sp{r}.chr1" for r in range(n)]
    seqs = [None] * n

    def fmt_len() -> float:
        return float(f"{max(rng.exponential(mean_branch), 1e-6):.6f}")

    root_seq = np.searchsorted(np.cumsum(pi), rng.random_sample(cols)).clip(0, 3)

    def descend(node, parent_seq) -> str:
        t = fmt_len()
        P = np.cumsum(_hky_matrix(pi, kappa, t), axis=1)
        u = rng.random_sample(cols)
        seq = (u[:, None] > P[parent_seq]).sum(axis=1).clip(0, 3)
        if isinstance(node, tuple):
            seqs[node[1]] = seq
            return f"{names[node[1]]}:{t:.6f}"
        inner = ",".join(descend(ch, seq) for ch in children[node])
        return f"({inner}):{t:.6f}"

    newick = "(" + ",".join(descend(ch, root_seq) for ch in children[0]) + ");"
    chars = [bytearray(_NT[s].tobytes()) for s in seqs]
    if gaps:
        for _ in range(rng.randint(2, 5)):
            length = (1, 2, 3, 3, 6)[rng.randint(5)]
            row = rng.randint(n)
            pos = rng.randint(0, max(1, cols - length))
            for c in range(pos, min(cols, pos + length)):
                chars[row][c] = ord("-")
    rows = []
    for r in range(n):
        s = chars[r].decode()
        rows.append(AlnRow(names[r], s, 1000 * index, sum(1 for ch in s if ch != "-"), "+", 10_000_000))
    return AlnBlock(rows, block_id=f"synth{index}", tree=newick, kappa=kappa)


def synth_blocks(n_blocks: int, n: int, cols: int, seed: int = 1, gaps: bool = True) -> List[AlnBlock]:
    rng = np.random.RandomState(seed)
    return [synth_block(rng, n, cols, i, gaps) for i in range(n_blocks)]


def to_maf(blocks: List[AlnBlock]) -> str:
    out = ["##maf version=1"]
    for b in blocks:
        out.append("a score=0")
        for r in b.rows:
            out.append(f"s {r.name} {r.start} {r.length} {r.strand} {r.full_length} {r.seq}")
        out.append("")
    return "\n".join(out) + "\n"
