#!/usr/bin/env python3
"""Randomised pin of the oracle: random blocks (shapes, gaps, ambiguity codes, score ties, zero-length branches, parameter sets)
through oracle/_ref/ref_harness -- the unmodified reference objects, only where /root/reference exists -- and through the oracle;
per-sample maxima, HSS tables and fit results must be identical.   usage: tools/soak_oracle_vs_ref.py [seconds] [rng_seed]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from rnacode_amd.alnio import write_harness_blocks  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

HARNESS = os.path.join("oracle", "_ref", "ref_harness")
EXOTIC = os.environ.get("RC_SOAK_EXOTIC", "") not in ("", "0")   # degenerate block classes as well (tools/soak_exotic.py)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from soak_exotic import exotic  # noqa: E402


def key(h):
    return (-h["score"], h["strand"], h["frame"], h["start"], h["end"])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # replay: do the work of this round only (the generator is advanced through the others)
    t0 = time.time()
    rounds = done = bad = ub = nbt = nan_bt = 0
    while time.time() - t0 < budget:
        rounds += 1
        n_samples = int(rng.choice([1, 2, 5, 20, 64, 65, 100]))
        seed = int(rng.randint(1, 2 ** 31 - 1))
        pars, extra = {}, []
        r = rng.rand()
        if r < 0.2:
            pars = dict(Delta=float(np.float32(-rng.uniform(1, 15))), Omega=float(np.float32(-rng.uniform(1, 8))), omega=float(np.float32(-rng.uniform(0.5, 4))),
                        stopPenalty_k=float(np.float32(-rng.uniform(2, 12))))
        elif r < 0.3:
            pars = dict(Delta=float(np.float32(rng.uniform(0, 2))), Omega=-4.0, omega=-2.0, stopPenalty_k=-8.0)
        elif EXOTIC and r < 0.4:   # any sign and size
            pars = {k: float(np.float32(rng.uniform(-12, 3))) for k in ("Delta", "Omega", "omega", "stopPenalty_k")}
        if pars:
            extra += ["--pars"] + [repr(pars[k]) for k in ("Delta", "Omega", "omega", "stopPenalty_k")]
        blosum = 90 if rng.rand() < 0.15 else 62
        if blosum == 90:
            extra += ["--blosum", "90"]
        blocks = []
        for i in range(int(rng.randint(1, 6))):
            n = int(rng.choice([3, 3, 4, 5, 6, 8, 12, 20, 40, 70]))
            cols = int(rng.choice([9, 12, 30, 45, 60, 90, 120, 200, 400]))
            if n * cols > 3000:
                cols = max(9, 3000 // n)
            b = synth_block(rng, n, cols, index=i, gaps=bool(rng.rand() < 0.8), mean_branch=float(rng.choice([0.01, 0.1, 0.4]))).upper()
            rows = [bytearray(x.seq.encode()) for x in b.rows]
            if rng.rand() < 0.3:
                ln = int(rng.choice([3, 4, 6, 7])); pos = int(rng.randint(0, max(1, cols - ln)))
                for q in range(1, n):
                    rows[q][pos:pos + ln] = b"-" * min(ln, cols - pos)
                m = list(re.finditer(r":[0-9.]+", b.tree))[int(rng.randint(0, 3))]
                b.tree = b.tree[:m.start()] + ":0.000000" + b.tree[m.end():]
            if rng.rand() < 0.3:
                for row in rows:
                    for q in np.flatnonzero(rng.rand(len(row)) < 0.04):
                        if row[q] != ord("-"):
                            row[q] = ord(str(rng.choice(list("NNNRYKMSWBDHVUacgt"))))
            if rng.rand() < 0.03:   # an alignment without two of the four nucleotides: HKY85 with zero frequencies, NaN background models and the
                for row in rows:    # reference's NaN-order-dependent MAX macro (score.h:28; the class of tests/data/soak_hss_positive_delta.json)
                    row[:] = row.replace(b"A", b"C").replace(b"G", b"T")
            if rng.rand() < 0.04:   # any two or three of the four nucleotides: one purine and one pyrimidine leave Seq-Gen's alphaR 0 / 0 -- NaN branch
                keep = list(rng.permutation(list("ACGT"))[:int(rng.choice([2, 2, 3]))])   # matrices, every simulated residue state 0 (evolve.c:167-175)
                for k_, row in enumerate(rows):
                    rows[k_] = bytearray("".join(ch if ch in keep or ch not in "ACGT" else keep[(ord(ch) + q) % len(keep)] for q, ch in enumerate(row.decode())).encode())
            for x, row in zip(b.rows, rows):
                x.seq = row.decode(); x.length = sum(ch != "-" for ch in x.seq)
            if EXOTIC:
                exotic(rng, b)
            blocks.append(b)
        want_bt = bool(rng.rand() < 0.3)   # also the backtrack of the best HSS (score.c:558-797), a third of the rounds
        if only and rounds != only:
            if rounds > only:
                break
            continue
        with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
            path = fh.name
        if only:
            write_harness_blocks(blocks, "/tmp/replay.blk")
            print("replay: /tmp/replay.blk", "--samples", n_samples, "--seed", seed, extra)
        try:
            write_harness_blocks(blocks, path)
            out = subprocess.run([HARNESS, path, "--samples", str(n_samples), "--seed", str(seed), "--no-digest", *([] if want_bt else ["--no-backtrack"]), *extra], capture_output=True)
        finally:
            os.unlink(path)
        if out.returncode != 0:
            # the documented out-of-bounds draw (seqgen/evolve.c:173,304: state 4, the next branch's probabilities read past their array) can end in a
            # segmentation fault instead of a wrong sample: if the oracle counts a clamped draw in one of the round's blocks, that is what happened
            clamped = 0
            for b in blocks:
                p = ob.default_params(n_samples, blosum=blosum)
                for k, v in pars.items():
                    setattr(p, k, v)
                clamped += ob.run_block([x.seq.upper() for x in b.rows], [x.name for x in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, p, seed).clamped
            if clamped > 0:
                ub += 1
                print(f"reference UB round {rounds}: the harness died (rc {out.returncode}) on a round with {clamped} clamped draw(s)")
            else:
                print(f"harness failed (rc {out.returncode}) in round {rounds}: {out.stderr.decode()[:200]}")
                bad += 1
            continue
        try:   # printf writes nan / inf where JSON wants NaN / Infinity
            refs = [json.loads(re.sub(r"(?<![A-Za-z])(-?)inf(?![A-Za-z])", r"\1Infinity", re.sub(r"-?nan", "NaN", l)))
                    for l in out.stdout.decode().splitlines() if l.strip()]
        except json.JSONDecodeError as e:
            print(f"unparsable harness output in round {rounds}: {e}; {out.stdout.decode()[max(0, e.pos - 80):e.pos + 80]!r}")
            bad += 1
            continue
        for b, ref in zip(blocks, refs):
            if "skipped" in ref:
                continue
            p = ob.default_params(n_samples, blosum=blosum)
            for k, v in pars.items():
                setattr(p, k, v)
            res = ob.run_block([x.seq.upper() for x in b.rows], [x.name for x in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, p, seed)
            ok = np.array_equal(np.float32(res.maxScores), np.float32(ref["maxScores"])) and res.evd_rc == ref["evd_rc"]
            got, want = sorted(res.hss, key=key), sorted(ref["hss"], key=key)
            ok = ok and [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
                [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
            if ok and ref["evd_rc"] == 1:
                ok = np.float32(res.mu) == np.float32(ref["mu"]) and np.float32(res.lam) == np.float32(ref["lambda"])
            if ok and "backtrack" in ref:
                bt = ref["backtrack"]
                rows = [x.seq.upper() for x in b.rows]
                names = [x.name for x in b.rows]
                m = ob.get_models(b.tree, rows, names, b.kappa, blosum)
                rrows = ob.rev_aln(rows)
                mr = ob.get_models(b.tree, rrows, names, b.kappa, blosum)
                _h, skf, skr = ob.score_aln(rows, b.rows[0].start, b.rows[0].length, m, mr, p, want_sk=True)
                # the harness backtracks results[0] after an unstable qsort by score: with several HSS tied for the best score (identical rows: all six
                # frames) that may be one on either strand -- the path must be the oracle's for one of the strands that hold such an HSS
                strands = sorted({w["strand"] for w in want if np.float32(w["score"]) == np.float32(want[0]["score"])})
                idx = list(range(bt["i"], bt["b"] + 1, -3))
                same = False
                for strand in strands:
                    sk, rr = (skf, rows) if strand == "+" else (skr, rrows)
                    st, z, tr = ob.backtrack(bt["b"], bt["i"], sk, rr, p)
                    if any(st[k, idx[0]] == -1 for k in range(1, b.n)):
                        # NaN scores at the end of the HSS (a block without two of the nucleotides, Delta >= 0): no state of the pair wins the comparison
                        # at score.c:606-611, opt_state stays -1 and the reference's walk indexes its matrices with it -- undefined, nothing to compare
                        same = True
                        nan_bt += 1
                        break
                    same = same or all(list(st[k, idx]) == bt["k"][k - 1]["states"] and list(z[k, idx]) == bt["k"][k - 1]["z"] and
                                       list(tr[k, idx]) == bt["k"][k - 1]["transitions"] for k in range(1, b.n))
                ok = ok and same
                nbt += 1
            done += 1
            if not ok and res.clamped > 0:
                # a draw beyond the last cumulative frequency: Seq-Gen returns state 4 there, prints it as '\0' (the sampled row ends early) and reads
                # the next branch's probabilities out of bounds (seqgen/evolve.c:173,304) -- undefined behaviour the oracle does not imitate
                # (it clamps to state 3 and counts)
                ub += 1
                print(f"reference UB round {rounds}: {b.n} x {b.cols}, n={n_samples}, seed={seed}: {res.clamped} clamped draw(s), results differ as documented")
                continue
            if not ok and only:
                mo, mr = np.float32(res.maxScores), np.float32(ref["maxScores"])
                print("maxima differ at", np.flatnonzero(mo != mr)[:10], mo[mo != mr][:5], mr[mo != mr][:5], "evd", res.evd_rc, ref["evd_rc"], res.mu, ref.get("mu"))
                print("hss oracle", [(g["strand"], g["frame"], g["start"], g["end"], g["score"]) for g in got][:6])
                print("hss ref   ", [(w["strand"], w["frame"], w["start"], w["end"], w["score"]) for w in want][:6])
            if not ok:
                bad += 1
                print(f"MISMATCH round {rounds}: {b.n} x {b.cols}, n={n_samples}, seed={seed}, pars={pars}, blosum={blosum}")
    print(f"oracle vs reference: {rounds} harness runs, {done} blocks compared in {time.time() - t0:.0f} s, {bad} mismatches, "
          f"{ub} blocks where the reference ran into its out-of-bounds draw; {nbt} backtrack paths compared"
          + (f" ({nan_bt} of them from NaN scores: the reference's walk is undefined there, not compared)" if nan_bt else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
