"""Colored alignment plots of high-scoring segments in EPS (the reference's --eps output).

Mirrors colorAln / colorHSS of src/postscript.c:38-680 byte for byte: the alignment in blocks of 60
columns with names, coordinates, ruler and conservation bars, then the segment -- extended left and
right to the next stop codon of the reference sequence (extendRegion, misc.c:555-626) -- colored codon
by codon from the backtracked state path (rc_batch_backtrack, score.c:558-797): synonymous /
conservative / radical substitutions, stops, gaps and out-of-frame stretches.

The layout arithmetic of the reference is in C float; the only place where that matters for the printed
"%.1f" values is the conservation bar height, computed here in numpy float32 like there.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np

from . import api
from .alnio import AlnBlock

# setParameters(), postscript.c:19-36
_COLUMN_WIDTH = 60
_FONT_W = 6.0
_FONT_H = 6.5
_LINE = _FONT_H + 2
_BLOCK = 3.5 * _FONT_H
_CONS = _FONT_H * 0.5
_SS = 12.0
_RULER = 2.0
_NAME_STEP = 3 * _FONT_W
_NUM_STEP = _FONT_W
_MAX_CONS_BAR = 2.5 * _FONT_H
_START_Y = 2.0
_NAMES_X = _FONT_W

_HEADER = (
    "%!PS-Adobe-3.0 EPSF-3.0\n"
    "%%BoundingBox: {} {} {} {}\n"
    "%%EndComments\n"
    "%Created by RNAcode; visit wash.github.com/rnacode\n"
    "% draws box in color given by hue and saturation\n"
    "/box {{ % x1 y1 x2 y2 hue saturation\n"
    "  gsave\n"
    "  dup 0.3 mul 1 exch sub sethsbcolor\n"
    "  exch 3 index sub exch 2 index sub rectfill\n"
    "  grestore\n"
    "}} def\n"
    "% draws a box in current color\n"
    "/box2 {{ % x1 y1 x2 y2\n"
    "  exch 3 index sub exch 2 index sub rectfill\n"
    "}} def\n"
    "/string {{ % (Text) x y\n"
    " 6 add\n"
    " moveto\n"
    "  show\n"
    "}} def\n"
    "0 {} translate\n"
    "1 -1 scale\n"
    "/Courier findfont\n"
    "[10 0 0 -10 0 0] makefont setfont\n"
)

_COLORS = (   # colorMatrix, postscript.c:355-359: [0] red (radical), [1] green (synonymous)
    ("0.0 0.0", "0.0 0.2", "0.0 0.4", "0.0 0.6", "0.0 0.8", "0.0 1"),
    ("0.32 0.1", "0.32 0.2", "0.32 0.4", "0.32 0.6", "0.32 0.8", "0.32 1"),
)
_AA = "ARNDCQEGHILKMFPSTWYV"   # decodeAA, code.c:236-266
_CONS_ORDER = "_ACGUTXKI"      # Law_and_Order, librna/pair_mat.h:4

Backtrack = Callable[[str, int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]]


def _nt(c: str) -> int:   # ntMap: score.c:41, RNAcode.c:94-98 -- everything that is not C, G, T, U is 0
    return {"C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3, "U": 3, "u": 3}.get(c, 0)


def _seq_length(seq: str) -> int:
    return sum(1 for ch in seq if ch != "-")


def _pos2col_map(seq: str) -> List[int]:
    """map[l] = 1-based column of the l-th residue (pos2col, misc.c:250-270); map[0] unused."""
    out = [0]
    for c, ch in enumerate(seq):
        if ch != "-":
            out.append(c + 1)
    return out


def _get_block(x: int, seq_0: str, seq_k: str, map_0: Sequence[int]) -> Tuple[str, str]:
    """The columns of codon ending at reference position x (getBlock, misc.c:186-245)."""
    start = map_0[x - 3] + 1 if x > 3 else 1
    end = map_0[x]
    return seq_0[start - 1:end], seq_k[start - 1:end]


def _codons(block_0: str, block_k: str) -> Tuple[str, str]:
    a = "".join(ch for ch in block_0 if ch != "-")
    b = "".join(k for ch, k in zip(block_0, block_k) if ch != "-")
    return a, b


def _rev_rows(rows: Sequence[str]) -> List[str]:
    comp = {"T": "A", "U": "A", "C": "G", "G": "C", "A": "T"}   # revAln, rnaz_utils.c:316-348
    return ["".join(comp.get(ch, ch) for ch in reversed(r)) for r in rows]


def consensus(rows: Sequence[str]) -> str:
    """rnaz_utils.c:249-264 with encode_char of librna/pair_mat.h:26-38."""
    out = []
    for col in zip(*rows):
        freq = [0] * 8
        for ch in col:
            code = _CONS_ORDER.find(ch)
            code = 0 if code < 0 else code
            if code > 4:
                code -= 1
            freq[code] += 1
        c = fm = 0
        for s in range(8):
            if freq[s] > fm:
                c, fm = s, freq[s]
        out.append(_CONS_ORDER[c])
    return "".join(out)


class _Tables:
    _cache: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}

    @classmethod
    def get(cls, blosum: int):
        if blosum not in cls._cache:
            cls._cache[blosum] = api.code_tables(blosum)
        return cls._cache[blosum]


def _pep(pep: np.ndarray, codon: str) -> int:
    return int(pep[16 * _nt(codon[0]) + 4 * _nt(codon[1]) + _nt(codon[2])])


def _translate(codon: str, pep: np.ndarray) -> str:   # translateSeq, code.c:103-128
    if any(ch not in "ACGTUacgtu" for ch in codon):
        return "?"
    p = _pep(pep, codon)
    return "*" if p == -1 else _AA[p]


def extend_region(rows: Sequence[str], pos: int, direction: int, pep: np.ndarray) -> int:
    """Walk codon by codon from the segment to the next stop codon of the reference row (misc.c:555-626)."""
    seq = rows[0]
    L = _seq_length(seq)
    map_0 = _pos2col_map(seq)
    x = pos + 2 if direction == 0 else pos
    while True:
        block_0, _ = _get_block(x, seq, rows[1], map_0)
        codon = "".join(ch for ch in block_0 if ch != "-")
        if _pep(pep, codon) == -1:
            break
        if direction == 0:
            if x - 3 < 3:
                break
            x -= 3
        else:
            if x + 3 > L:
                break
            x += 3
    return x - 2 if direction == 0 else x


def _f(x: float) -> str:
    return "%.1f" % x


def _color_hss(out: List[str], rows: Sequence[str], bt, label: str, b: int, i: int, column_width: int, seqs_x: float,
               pep: np.ndarray, matrix: np.ndarray) -> None:
    """colorHSS, postscript.c:334-680.  bt = (states, z, transitions) arrays [row][position]."""
    N = len(rows)
    seq_0 = rows[0]
    map_0 = _pos2col_map(seq_0)
    states, _, transitions = bt if bt is not None else (None, None, None)
    syn: Dict[int, int] = {}
    non_syn: Dict[int, int] = {}
    for x in range(b + 2, i + 3, 3):
        syn_count = non_syn_count = 0
        seen = set()
        for k in range(1, N):
            block_0, block_k = _get_block(x, seq_0, rows[k], map_0)
            codon_a, codon_b = _codons(block_0, block_k)
            pep_a = _pep(pep, codon_a)
            if "-" in codon_b:
                continue
            key = (_nt(codon_b[0]), _nt(codon_b[1]), _nt(codon_b[2]))
            if key in seen:
                continue
            seen.add(key)
            pep_b = _pep(pep, codon_b)
            if pep_a != -1 and pep_b != -1:
                score = int(matrix[pep_a][pep_b])
                if codon_a != codon_b:
                    if pep_a == pep_b:
                        syn_count += 1
                    elif score < 0:
                        non_syn_count += 1
            else:
                non_syn_count += 1
        syn[x] = syn_count
        non_syn[x] = non_syn_count

    row_pitch = _LINE * (N + 2) + _BLOCK + _CONS + _RULER
    for x in range(b + 2, i + 3, 3):
        syn_color = _COLORS[1][min(syn[x], 5)]
        non_syn_color = _COLORS[0][min(non_syn[x], 5)]
        for k in range(N):
            block_0, block_k = _get_block(x, seq_0, rows[k], map_0)
            codon_a, codon_b = _codons(block_0, block_k)
            pep_a = _pep(pep, codon_a)
            if "-" in codon_b:
                score, pep_b = -1, -99   # contains gap
            else:
                pep_b = _pep(pep, codon_b)
                if pep_a != -1 and pep_b != -1:
                    score = int(matrix[pep_a][pep_b])
                else:
                    score, pep_b = -1, 99   # stop
            blen = len(block_0)
            for ii in range(blen):
                curr_col = map_0[x] - blen + ii
                block = math.ceil(np.float32(curr_col + 1) / np.float32(column_width))
                xx = seqs_x + (curr_col - (block - 1) * column_width) * _FONT_W
                yy = _START_Y + (block - 1) * row_pitch + _SS * block + (k + 1) * _LINE
                if k == 0 and label != "" and x == b + 2 and ii == 0:
                    out.append("0.15 0.5 0.6 sethsbcolor\n")
                    out.append("/Helvetica findfont\n")
                    out.append("[8 0 0 -8 0 0] makefont setfont\n")
                    out.append("(%s) %s %s string\n" % (label, _f(xx), _f(yy - 2 * _LINE)))
                    out.append("0.0 setgray\n")
                if k == 0 and label != "":   # translation line
                    off_l = 0.5 if ii == 0 else 0.0
                    off_r = 0.5 if ii == blen - 1 else 0.0
                    out.append("%s %s %s %s %s box\n" % (_f(xx + off_l), _f(yy - 1), _f(xx + _FONT_W - off_r), _f(yy - _LINE - 1), "0.15 0.5"))
                if k == 0 and ii == math.ceil(blen / 2.0) - 1:
                    out.append("/Courier findfont\n")
                    out.append("[10 0 0 -10 0 0] makefont setfont\n")
                    out.append("(%s) %s %s string\n" % (_translate(codon_a, pep), _f(xx), _f(yy - _LINE)))
                box = (_f(xx), _f(yy - 1), _f(xx + _FONT_W), _f(yy + _FONT_H + 1))
                if k == 0:
                    out.append("%s %s %s %s %s box\n" % (*box, "0.0 0.0"))
                    out.append("/Courier-Bold findfont\n")
                    out.append("[10 0 0 -10 0 0] makefont setfont\n")
                    out.append("(%s) %s %s string\n" % (block_k[ii], _f(xx), _f(yy)))
                if k > 0:
                    st, tr = int(states[k][x]), int(transitions[k][x])
                    if st == 0 and tr == 0:   # in frame
                        if score >= 0:
                            out.append("%s %s %s %s %s box\n" % (*box, syn_color if codon_a != codon_b else "0.0 0.0"))
                        else:
                            if pep_b == 99:
                                out.append("%s %s %s %s %s box\n" % (*box, "0.6 1.0"))
                            if pep_b == -99:
                                out.append("%s %s %s %s %s box\n" % (*box, "0.0 0.0"))
                            if pep_b != 99 and pep_b != -99:
                                out.append("%s %s %s %s %s box\n" % (*box, non_syn_color))
                        if pep_a == pep_b and codon_a != codon_b:
                            out.append("/Courier-Bold findfont\n")
                        else:
                            out.append("/Courier findfont\n")
                        out.append("[10 0 0 -10 0 0] makefont setfont\n")
                    if tr == 2:
                        out.append("/Courier-Bold findfont\n")
                        out.append("[10 0 0 -10 0 0] makefont setfont\n")
                        out.append("0.2 setgray\n")
                        out.append("%s %s %s %s box2\n" % box)
                        out.append("0.8 setgray\n")
                    if tr == 1 or (tr == 0 and st != 0):
                        out.append("/Courier findfont\n")
                        out.append("[10 0 0 -10 0 0] makefont setfont\n")
                        out.append("0.8 setgray\n")
                        out.append("%s %s %s %s box2\n" % box)
                        out.append("0 setgray\n")
                out.append("(%s) %s %s string\n" % (block_k[ii], _f(xx), _f(yy)))
                out.append("0 setgray\n")


def color_aln(block: AlnBlock, hss: dict, backtrack: Backtrack, blosum: int = 62) -> str:
    """EPS text for one high-scoring segment of `block` (colorAln, postscript.c:38-332).

    hss: a record of scoreAln (strand, frame, start, end, pvalue).  backtrack(strand, b, i) returns the
    (states, z, transitions) arrays of rc_batch_backtrack for that strand."""
    pep, matrix = _Tables.get(blosum)
    rows = [r.seq for r in block.rows]
    names = [r.name for r in block.rows]
    N = len(rows)
    length = len(rows[0])
    max_name = max(len(n) for n in names)
    seqs_x = _NAMES_X + max_name * _FONT_W + _NAME_STEP
    max_num = 10
    column_width = _COLUMN_WIDTH
    tmp_columns = column_width
    if length < column_width:
        column_width = length
        tmp_columns = length
    image_w = math.ceil(_NAMES_X + (max_name + tmp_columns + max_num) * _FONT_W + 2 * _NAME_STEP + _FONT_W + _NUM_STEP)
    image_h = _START_Y + math.ceil(np.float32(length) / np.float32(column_width)) * ((N + 2) * _LINE + _BLOCK + _CONS + _SS + _RULER)
    out: List[str] = [_HEADER.format(0, 0, int(image_w), int(image_h), int(image_h))]

    ruler = ["."] * (2 * length + 16)
    for i in range(length):
        if (i + 1) % 10 == 0 and (i + 1) % column_width != 0:
            digits = str(i + 1)[:max(length - 1, 0)]   # snprintf(tmpBuffer, length, ...)
            ruler[i:i + len(digits)] = digits
    ruler_s = "".join(ruler[:length])

    cons = consensus(rows)
    strand = hss["strand"]
    curr_y = _START_Y
    curr_pos = 0
    f32 = np.float32
    while curr_pos < length:
        out.append("0 setgray\n")
        curr_y += _SS + _LINE
        for r, name, seq in zip(block.rows, names, rows):
            chunk = seq[curr_pos:curr_pos + column_width]
            match = _seq_length(seq[:curr_pos + len(chunk)])
            if strand == "+":
                match += r.start
            else:
                match = _seq_length(seq) - match + 1
            out.append("(%s) %s %s string\n" % (name, _f(_NAMES_X), _f(curr_y)))
            out.append("(%i) %s %s string\n" % (match, _f(seqs_x + _FONT_W * len(chunk) + _NUM_STEP), _f(curr_y)))
            curr_y += _LINE
        curr_y += _RULER
        chunk = ruler_s[curr_pos:curr_pos + column_width]
        out.append("(%s) %s %s string\n" % (chunk, _f(seqs_x), _f(curr_y)))
        out.append("(%s) %s %s string\n" % (chunk, _f(seqs_x), _f(curr_y)))
        curr_y += _LINE
        curr_y += _CONS
        out.append("0.6 setgray\n")
        for i in range(curr_pos, min(curr_pos + column_width, length)):
            match = 0
            for seq in rows:
                if cons[i] == seq[i]:
                    match += 1
                if cons[i] == "U" and seq[i] == "T":
                    match += 1
                if cons[i] == "T" and seq[i] == "U":
                    match += 1
            score = f32(match - 1) / f32(N - 1)
            if cons[i] in "-_.":
                score = f32(0)
            bar = f32(_MAX_CONS_BAR) * score
            if bar == 0:
                bar = f32(1)
            xx = seqs_x + (i - curr_pos) * _FONT_W
            top = (f32(curr_y) + f32(_MAX_CONS_BAR)) - bar
            out.append("%s %s %s %s box2\n" % (_f(xx), _f(float(top)), _f(xx + _FONT_W), _f(curr_y + _MAX_CONS_BAR)))
        curr_y += _BLOCK
        curr_pos += column_width

    out.append("0.0 setgray\n")
    curr = rows if strand == "+" else _rev_rows(rows)
    start, end = int(hss["start"]), int(hss["end"])
    for part in range(3):
        if part == 0:
            b, e = extend_region(curr, start, 0, pep), start - 1
            label = ""
        elif part == 1:
            b, e = start, end
            p = float(hss["pvalue"])
            if p < 0.001:
                ps = "<1e-16\n" if p < 10e-16 else "%9.1e\n" % p
            else:
                ps = "%9.3f\n" % p
            label = "Frame %s%i p =%s" % (strand, int(hss["frame"]) + 1, ps)
        else:
            b, e = end + 1, extend_region(curr, end, 1, pep)
            if b >= e:
                break
            label = ""
        bt = backtrack(strand, b, e) if b + 2 < e + 3 else None
        _color_hss(out, curr, bt, label, b, e, column_width, seqs_x, pep, matrix)
    out.append("showpage\n")
    return "".join(out)
