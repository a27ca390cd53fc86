"""RNAcode's result listings (reference: src/misc.c:392-552 `printResults`, footer RNAcode.c:223-228).

Formats: 0 = the default table, 1 = GTF (`--gtf`), 2 = tab separated (`--tabular`).  The HSS
counter runs across alignment blocks and is not advanced after a `--best-only` break, exactly as
the reference's global `hitCounter` behaves."""
from __future__ import annotations

from dataclasses import dataclass
from typing import IO, List

import numpy as np


@dataclass
class ReportState:
    hit_counter: int = 0


def _c_e(value: float, prec: int, width: int, space_flag: bool) -> str:
    """C's "% W.Pe" / "%W.Pe" (Python prints the same mantissa/exponent form)."""
    s = f"{value:.{prec}e}"
    if space_flag and not s.startswith("-"):
        s = " " + s
    return s.rjust(width)


def _c_f(value: float, prec: int, width: int, space_flag: bool) -> str:
    s = f"{value:.{prec}f}"
    if space_flag and not s.startswith("-"):
        s = " " + s
    return s.rjust(width)


def print_results(out: IO[str], fmt: int, hss: List[dict], ref_name: str, state: ReportState, cutoff: float = 1.0,
                  best_only: bool = False, best_region: bool = False, eps=None, eps_cutoff: float = 0.05) -> None:
    """hss: the block's HSS (dicts as returned by Batch.scoreAln, any order), p-values filled.
    eps: optional callback eps(hit_counter, hss_record), called for every listed HSS with p < eps_cutoff
    before its line is written -- where the reference draws hss-<counter>.eps (misc.c:461-474)."""
    res = [dict(h, hide=False) for h in hss if h["score"] > 0.0]
    if best_region:   # misc.c:408-433: sort by start codon, hide the weaker of two overlapping HSS
        res.sort(key=lambda h: h["startSite"])
        curr, nxt = 0, 1
        while nxt < len(res):
            if not (res[curr]["endSite"] <= res[nxt]["startSite"]):
                if res[curr]["score"] > res[nxt]["score"]:
                    res[nxt]["hide"] = True
                else:
                    res[curr]["hide"] = True
                    curr = nxt
            else:
                curr = nxt
            nxt += 1
    res.sort(key=lambda h: -h["score"])
    cutoff32 = float(np.float32(cutoff))
    if not res or float(np.float32(res[0]["pvalue"])) > cutoff32:
        if fmt == 0:
            out.write("\nNo significant coding regions found.\n")
        return
    if fmt == 0:
        out.write("\n%6s%5s%7s%6s%6s%12s%12s%12s%9s%9s\n" % (" HSS # ", "Frame", "Length", "From", "To", "Name", "Start",
                                                           "End", "Score", "P"))
        out.write("======================================================================================\n")
    for h in res:
        p = float(np.float32(h["pvalue"]))
        if not (p < cutoff32):
            break
        if h["hide"]:
            continue
        if eps is not None and p < float(np.float32(eps_cutoff)):
            eps(state.hit_counter, h)
        length = h["endSite"] - h["startSite"] + 1
        if fmt == 0:
            out.write("%6i %4s%i%7i%6i%6i%12s%12i%12i%9.2f" % (state.hit_counter, h["strand"], h["frame"] + 1, length,
                                                              h["startSite"] + 1, h["endSite"] + 1, ref_name,
                                                              h["startGenomic"], h["endGenomic"], h["score"]))
            if p < 0.001:
                out.write("   <1e-16\n" if p < 10e-16 else _c_e(p, 1, 9, True) + "\n")
            else:
                out.write(_c_f(p, 3, 9, True) + "\n")
        elif fmt == 1:
            k = ref_name.find(".")
            name = ref_name if k < 0 else ref_name[k + 1:]
            out.write("%s\t%s\t%s\t%i\t%i\t%.2f|%s\t%s\t%s\t%s%i%s\n" % (
                name, "RNAcode", "CDS", h["startGenomic"] + 1, h["endGenomic"] + 1, h["score"], f"{p:.2e}", h["strand"], ".",
                'gene_id "Gene', state.hit_counter, '"; transcript_id "transcript 0";'))
        else:
            out.write("%i\t%s\t%i\t%i\t%i\t%i\t%s\t%i\t%i\t%7.3f\t" % (state.hit_counter, h["strand"], h["frame"] + 1, length,
                                                                     h["startSite"] + 1, h["endSite"] + 1, ref_name,
                                                                     h["startGenomic"], h["endGenomic"], h["score"]))
            out.write((_c_e(p, 3, 9, True) if p < 0.001 else _c_f(p, 3, 9, True)) + "\n")
        if best_only:
            break
        state.hit_counter += 1


def print_footer(out: IO[str], n_alignments: int, seconds: float, sampleN: int, Delta: float, Omega: float, omega: float,
                 stop_penalty: float) -> None:
    """RNAcode.c:223-228 (format 0 only)."""
    out.write("\n%i alignment(s) scored in %.2f seconds. Parameters used:\nN=%i, Delta=%.2f, Omega=%.2f, omega=%.2f, "
              "stop penalty=%.2f\n\n" % (n_alignments, seconds, sampleN, Delta, Omega, omega, stop_penalty))
