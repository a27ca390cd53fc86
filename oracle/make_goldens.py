#!/usr/bin/env python3
"""Generate tests/golden/*.json.gz from the UNMODIFIED reference objects (development container only).

Runs oracle/_ref/ref_harness (built by `make -C oracle ref` from /root/reference, see
oracle/Makefile and oracle/ref_harness.c) on the reference's example alignments and on blocks
from the deterministic synthetic generator, and stores inputs + reference outputs as JSON.
Only data is committed: alignment rows, tree/kappa, parameters, seed base -> models, HSS table,
per-sample maxima, mu, lambda, p-values, Sk digests, backtrack paths.

Usage:  python oracle/make_goldens.py            (needs /root/reference)
"""
from __future__ import annotations

import gzip
import json
import re
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from rnacode_amd.alnio import AlnBlock, AlnRow, read_alignment_file, write_harness_blocks  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

REF = os.environ.get("RNACODE_REF", "/root/reference")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden")
SEED = 42


def run_harness(blocks, samples, extra=()):
    with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
        path = fh.name
    try:
        write_harness_blocks(blocks, path)
        out = subprocess.check_output([HARNESS, path, "--samples", str(samples), "--seed", str(SEED), *extra])
    finally:
        os.unlink(path)
    # (printf writes "-nan" for the self-pair model's expected scores when a zero distance makes its probabilities exactly 1, 0, 0, 0)
    return [json.loads(re.sub(r"-?nan", "NaN", l)) for l in out.decode().splitlines() if l.strip()]


def block_json(b: AlnBlock):
    return dict(id=b.block_id, tree=b.tree, kappa=b.kappa,
                rows=[dict(name=r.name, seq=r.seq, start=r.start, length=r.length, strand=r.strand,
                           full_length=r.full_length) for r in b.rows])


PRINT_VARIANTS = {   # reference printResults (misc.c:392-552) under these option sets
    "list": ["--format", "0"],
    "gtf": ["--format", "1"],
    "tabular": ["--format", "2"],
    "tabular_p0.05": ["--format", "2", "--cutoff", "0.05"],
    "list_best_only": ["--format", "0", "--best-only"],
    "tabular_best_region": ["--format", "2", "--best-region"],
}


def run_print(blocks, samples, extra, variant_args):
    with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
        path = fh.name
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        outp = fh.name
    try:
        write_harness_blocks(blocks, path)
        subprocess.check_output([HARNESS, path, "--samples", str(samples), "--seed", str(SEED), "--no-backtrack",
                                 *extra, "--print-to", outp, *variant_args])
        with open(outp) as fh:
            return fh.read()
    finally:
        os.unlink(path)
        os.unlink(outp)


def emit(name, blocks, samples, params=None, extra=(), note="", with_print=False):
    res = run_harness(blocks, samples, extra)
    assert len(res) == len(blocks), (len(res), len(blocks))
    doc = dict(fixture=name, note=note, seed_base=SEED, samples=samples,
               params=params or {}, harness_args=list(extra),
               blocks=[dict(input=block_json(b), ref=r) for b, r in zip(blocks, res)])
    if with_print:
        ex = [a for a in extra if a != "--no-backtrack"]
        doc["print"] = {k: run_print(blocks, samples, ex, v) for k, v in PRINT_VARIANTS.items()}
    path = os.path.join(OUT, name + ".json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(doc, separators=(",", ":")).encode())
    nscored = sum(1 for r in res if "skipped" not in r)
    print(f"{name}: {len(blocks)} blocks ({nscored} scored), n={samples}, {os.path.getsize(path)} bytes")


def edge_blocks():
    """Hand-made blocks for edge cases the reference handles (tree/kappa fitted by treeML)."""
    out = []
    # lower case + U + N + IUPAC in several rows, gaps in the reference row incl. leading gaps
    rows = [
        ("e1.a", "--atgGCUaaaNNNgcaugc---gatcgaucgRYKatgccgUAA"),
        ("e1.b", "ccATGGCTAAAGGGGCATGC---GATCGATCGTTTATGCCGTAA"),
        ("e1.c", "--ATGGCAAAAGGTGCNTGCTTTGATCG-TCGTTTATGCCGTGA"),
        ("e1.d", "--ATGGCTAAGGGGGCATGC---GACCGATCGTTCATGCC-TAA"),
    ]
    out.append(AlnBlock([AlnRow(n, s, 100, sum(c != "-" for c in s), "+", 5000) for n, s in rows], "edge_chars"))
    # minimal: three rows, nine columns
    rows = [("m.a", "ATGGCTAAA"), ("m.b", "ATGGCAAAA"), ("m.c", "ATGGCTAAG")]
    out.append(AlnBlock([AlnRow(n, s, 0, 0, "?", 0) for n, s in rows], "edge_min3x9"))
    # too short / too few rows: skipped by the driver (RNAcode.c:142-150)
    out.append(AlnBlock([AlnRow("s.a", "A-T", 0, 0, "?", 0), AlnRow("s.b", "ACT", 0, 0, "?", 0),
                         AlnRow("s.c", "AGT", 0, 0, "?", 0)], "edge_short"))
    out.append(AlnBlock([AlnRow("t.a", "ATGGCTAAA", 0, 0, "?", 0), AlnRow("t.b", "ATGGCAAAA", 0, 0, "?", 0)],
                        "edge_two_rows"))
    # reference row on the minus strand with MAF coordinates
    rows = [("r.a", "ATGGCTAAAGGGGCATGCGATCGATCGTTTATGCCGTAA"), ("r.b", "ATGGCTAAAGGAGCATGCGATCGTTCGTTTATGCCGTAA"),
            ("r.c", "ATGGCAAAAGGGGCATGTGATCGATCGTTTATGCCTTAA"), ("r.d", "ATGGCTAAGGGGGCTTGCGATCGATCCTTTATGCCGTAA")]
    out.append(AlnBlock([AlnRow(n, s, 777, len(s), "-", 9999) for n, s in rows], "edge_minus_ref"))
    return out


def main():
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(OUT, exist_ok=True)
    ex = os.path.join(REF, "examples")

    for stem in ("coding.aln", "noncoding.aln", "coding.maf", "noncoding.maf"):
        emit(stem.replace(".", "_") + "_n100", read_alignment_file(os.path.join(ex, stem)), 100,
             note=f"reference example {stem}, tree+kappa from the reference's treeML", with_print=True)
    gp = read_alignment_file(os.path.join(ex, "genomic-preprocessed.maf"))
    emit("genomic_preprocessed_n100", gp, 100, note="BASELINE config 2 input, all 34 blocks", with_print=True)
    emit("genomic_preprocessed_n1000", gp, 1000, extra=("--no-backtrack",), note="BASELINE config 2")

    emit("edge_cases_n50", edge_blocks(), 50, note="character classes, minimal sizes, skip rules, minus-strand ref")

    s6 = synth_blocks(6, 6, 120, seed=1)
    emit("synth_6x120_n200", s6, 200, note="BASELINE config 3 shape (6 seqs x 120 cols), sidecar tree+kappa")
    s12 = synth_blocks(2, 12, 300, seed=2)
    emit("synth_12x300_n40", s12, 40, note="BASELINE config 5 shape (12 seqs x 300 cols), sidecar tree+kappa")
    s4 = synth_blocks(4, 5, 60, seed=3)
    emit("synth_5x60_blosum90_n100", s4, 100, params=dict(blosum=90), extra=("--blosum", "90"))
    emit("synth_5x60_pars_n100", s4, 100, params=dict(Delta=-7.5, Omega=-3.0, omega=-1.5, stopPenalty_k=-6.0),
         extra=("--pars", "-7.5", "-3.0", "-1.5", "-6.0"))
    emit("synth_5x60_stopearly_n100", s4, 100, params=dict(stopEarly=1, cutoff=0.05),
         extra=("--stop-early", "--cutoff", "0.05"))
    # many sequences (wide N), short
    s20 = synth_blocks(2, 20, 45, seed=4)
    emit("synth_20x45_n50", s20, 50, note="wide alignment")
    # rooted (bifurcating root) sidecar tree: Seq-Gen then evolves only two root branches (evolve.c:429-432)
    rooted = synth_blocks(2, 4, 48, seed=5)
    for b in rooted:
        # (A,B,(C,D)) -> ((A,B):0.01,(C,D)) keeps names; re-root by grouping the first two root children
        t = b.tree
        depth = 0
        cut = []
        for i, ch in enumerate(t):
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == "," and depth == 1:
                cut.append(i)
        b.tree = "(" + t[:cut[1]] + "):0.010000" + t[cut[1]:]
    emit("synth_rooted_4x48_n100", rooted, 100, note="rooted sidecar tree")




def emit_eps(name, base_fixture, blocks, samples, keep):
    """EPS plots of the reference (src/postscript.c through printResults --eps) for `blocks`: all file names the
    run produced, and the text of the files listed in `keep`.  Inputs and HSS tables are those of `base_fixture`."""
    tmpdir = tempfile.mkdtemp()
    with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
        path = fh.name
    try:
        write_harness_blocks(blocks, path)
        epsdir = os.path.join(tmpdir, "eps")
        subprocess.check_output([HARNESS, path, "--samples", str(samples), "--seed", str(SEED), "--no-backtrack",
                                 "--print-to", os.path.join(tmpdir, "list.txt"), "--format", "0", "--eps-dir", epsdir])
        names = sorted(os.listdir(epsdir), key=lambda s: int(s[4:-4]))
        files = {}
        for k in keep:
            with open(os.path.join(epsdir, k)) as fh:
                files[k] = fh.read()
    finally:
        os.unlink(path)
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)
    doc = dict(fixture=name, base=base_fixture, samples=samples, seed_base=SEED, eps_cutoff=0.05, names=names, files=files,
               note="conservation bars depend on the harness's re-statement of consensus() (see oracle/ref_harness.c)")
    out = os.path.join(OUT, name + ".json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps(doc, separators=(",", ":")).encode())
    print(f"{name}: {len(names)} plots, {len(files)} kept, {os.path.getsize(out)} bytes")


def main_chars():
    """Ambiguity codes / N / U / lower case sprinkled over synthetic blocks, and blocks wider than the register-resident kernels
    take (33..64 rows: the EXACT instantiations, > 64 rows: the generic kernels) -- all scored by the reference itself."""
    import numpy as np
    rng = np.random.RandomState(11)
    blocks = synth_blocks(6, 5, 60, seed=6)
    for b in blocks:
        for x in b.rows:
            row = bytearray(x.seq.encode())
            for q in np.flatnonzero(rng.rand(len(row)) < 0.06):
                if row[q] != ord("-"):
                    row[q] = ord(str(rng.choice(list("NNNRYKMSWBDHVUacgtn"))))
            x.seq = row.decode()
    emit("synth_iupac_5x60_n50", blocks, 50, note="6 % of the residues replaced by ambiguity codes, N, U or lower case")
    # stretches where every non-reference row is gapped (sigma = 0: equal consecutive scores, getHSS's 'equal score, longer segment'
    # rule decides, score.c:953-954), frame-shifting gap lengths, zero-length branches (degenerate cumulative rows, evolve.c:292)
    import re
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(77)
    ties = []
    for i in range(12):
        n = int(rng.randint(3, 7))
        cols = int(rng.randint(45, 121))
        b = synth_block(rng, n, cols, index=i, gaps=bool(i % 2), mean_branch=(0.02, 0.1, 0.3)[i % 3]).upper()
        rows = [bytearray(r.seq.encode()) for r in b.rows]
        for _ in range(int(rng.randint(1, 4))):
            length = int((3, 6, 9, 12, 4, 7)[rng.randint(6)])
            pos = int(rng.randint(0, cols - length))
            for r in range(1, n):
                rows[r][pos:pos + length] = b"-" * length
        for r, row in zip(b.rows, rows):
            r.seq = row.decode()
            r.length = sum(1 for ch in r.seq if ch != "-")
        if i % 2 == 0:
            lens = list(re.finditer(r":[0-9.]+", b.tree))
            for m in [lens[j] for j in rng.choice(len(lens), size=min(2, len(lens)), replace=False)]:
                b.tree = b.tree[:m.start()] + ":0.000000" + b.tree[m.end():]
        ties.append(b)
    emit("synth_ties_zero_branches_n60", ties, 60, note="common gap stretches (score ties), frame-shifting gaps, zero-length branches")
    emit("synth_5x60_delta_pos_n100", synth_blocks(4, 5, 60, seed=3), 100, params=dict(Delta=0.25, Omega=-4.0, omega=-2.0, stopPenalty_k=-8.0),
         extra=("--pars", "0.25", "-4.0", "-2.0", "-8.0"), note="Delta >= 0: the MAX3 with Delta decides cells (the EXACT instantiations)")
    emit("synth_long_5x900_n20", synth_blocks(1, 5, 900, seed=12) + synth_blocks(1, 4, 1500, seed=13), 20,
         note="long blocks: the sigma codes no longer fit in LDS (read from the staging scratch)")
    emit("synth_5x60_n5", synth_blocks(3, 5, 60, seed=14), 5, note="five samples only: the Gumbel fit on next to nothing")
    emit("synth_wide_40x90_n30", synth_blocks(1, 40, 90, seed=7), 30, note="40 rows: the EXACT instantiations for 33..64 rows")
    emit("synth_wide_70x120_n20", synth_blocks(1, 70, 120, seed=8) + synth_blocks(1, 130, 60, seed=9), 20,
         note="70 and 130 rows: the generic kernels for more than 64 rows (one tile and several tiles per DP row)")


def main_eps():
    ex = os.path.join(REF, "examples")
    emit_eps("eps_coding_aln_n100", "coding_aln_n100", read_alignment_file(os.path.join(ex, "coding.aln")), 100,
             ["hss-0.eps", "hss-1.eps"])
    gp = read_alignment_file(os.path.join(ex, "genomic-preprocessed.maf"))
    emit_eps("eps_genomic_preprocessed_n100", "genomic_preprocessed_n100", gp, 100,
             ["hss-8.eps", "hss-12.eps", "hss-33.eps", "hss-104.eps", "hss-133.eps"])


def main_breakmaf():
    """Input/output pairs of the reference's scripts/breakMAF.pl (run with perl): its own example
    (examples/genomic.maf -> the shipped examples/genomic-preprocessed.maf) and synthetic long blocks with other
    parameters."""
    from rnacode_amd.synth import to_maf
    script = os.path.join(REF, "scripts", "breakMAF.pl")
    cases = []
    with open(os.path.join(REF, "examples", "genomic.maf")) as fh:
        cases.append(dict(name="genomic.maf", args=[], input=fh.read()))
    txt = to_maf(synth_blocks(3, 5, 950, seed=3) + synth_blocks(2, 4, 120, seed=4))
    cases.append(dict(name="synthetic default", args=[], input=txt))
    cases.append(dict(name="synthetic 300/90", args=["--maxLength", "300", "--desiredLength", "90"], input=txt))
    for c in cases:
        c["output"] = subprocess.run(["perl", script, *c["args"]], input=c["input"].encode(), stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL, check=True).stdout.decode()
    with open(os.path.join(REF, "examples", "genomic-preprocessed.maf")) as fh:
        assert cases[0]["output"] == fh.read()      # the reference ships this output
    out = os.path.join(OUT, "breakmaf_cases.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as fh:
        fh.write(json.dumps(dict(fixture="breakmaf_cases", cases=cases), separators=(",", ":")).encode())
    print(f"breakmaf_cases: {len(cases)} cases, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "eps":
        main_eps()
    elif len(sys.argv) > 1 and sys.argv[1] == "breakmaf":
        main_breakmaf()
    elif len(sys.argv) > 1 and sys.argv[1] == "chars":
        main_chars()
    else:
        main()
        main_chars()
        main_eps()
        main_breakmaf()
