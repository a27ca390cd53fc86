"""Randomised parity soak: random block shapes, sample counts, seeds and parameters through the HIP path and the
oracle until the time budget is used up; any difference in per-sample maxima, HSS tables or fit results is printed.
usage: python tools/soak_parity.py [seconds] [rng_seed] [wide|fat]     (RC_SOAK_EXOTIC=1: degenerate block classes as well)     wide: 65..300 rows only (the kernels for more than 64 rows);
       fat: one row count per batch, 3..6 rows, up to 156 columns (the batches that take k_null's two-rows-per-pass instantiations)
tests/test_gpu_soak.py runs a bounded slice of every mode under pytest -m gpu (soak(budget, seed, mode, max_blocks))."""
import re
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, "tests")
from oracle import binding as ob  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from soak_exotic import exotic  # noqa: E402

EXOTIC = __import__("os").environ.get("RC_SOAK_EXOTIC", "") not in ("", "0")   # degenerate block classes as well (tools/soak_exotic.py)


def key(h):
    return (-h["score"], h["strand"], h["frame"], h["start"], h["end"])


def soak(budget=60.0, rng_seed=1, mode="", max_blocks=None, log=print):
    """(batches, blocks compared, mismatches, backtrack paths compared); stops at the time budget or after max_blocks blocks."""
    rng = np.random.RandomState(rng_seed)
    wide = mode == "wide"
    fat = mode == "fat"
    ctx = api.Context(0)
    log(f"soak{' ' + mode if mode else ''}: seed {rng_seed}, the EVD fit's exp: {ctx.fit_exp_mode} (rc_ctx_fit_exp_mode; the oracle's is this process's C library)")
    t0 = time.time()
    rounds = blocks_done = bad = paths = 0
    while time.time() - t0 < budget and (max_blocks is None or blocks_done < max_blocks):
        rounds += 1
        n_samples = int(rng.choice([1, 2, 63, 64, 65, 100, 129, 200, 333]))
        if EXOTIC and rng.rand() < 0.15:
            n_samples = int(rng.choice([3, 4, 127, 128, 255, 256, 257, 512, 1000]))
        seed = int(rng.randint(1, 2 ** 31 - 1))
        pars = {}
        r = rng.rand()
        if r < 0.15:
            pars = dict(Delta=float(-rng.uniform(1, 15)), Omega=float(-rng.uniform(1, 8)), omega=float(-rng.uniform(0.5, 4)))
        elif r < 0.2:
            pars = dict(Delta=float(rng.uniform(0, 2)))
        elif r < 0.3:
            pars = dict(blosum=90)
        elif EXOTIC and r < 0.38:   # any sign and size for every gap and stop parameter (omega > 0 keeps a block out of the tiled kernels, Delta >= 0 takes the EXACT ones)
            pars = {k: float(np.float32(rng.uniform(-12, 3))) for k in ("Delta", "Omega", "omega", "stopPenalty_k", "stopPenalty_0") if rng.rand() < 0.7}
        blocks = []
        fat_n = int(rng.choice([3, 4, 5, 6]))
        most = int(__import__('os').environ.get('RC_SOAK_MAX_BLOCKS_PER_BATCH', 0)) or (5 if wide else 24)   # (tiny batches: the row-split path)
        for i in range(int(rng.randint(1, most + 1))):
            n = int(rng.choice([3, 3, 4, 5, 6, 6, 7, 8, 10, 12, 17, 32, 33, 40, 64, 65, 90, 130]))
            cols = int(rng.choice([9, 12, 30, 45, 60, 90, 120, 150, 200, 301]))
            if fat:
                n = fat_n
                cols = int(rng.choice([9, 12, 30, 45, 60, 90, 99, 120, 150, 156]))
            if wide:
                n = int(rng.choice([65, 66, 70, 90, 128, 129, 200, 300]))
                cols = int(rng.choice([9, 30, 60, 99, 120, 150, 200]))
                cols = max(9, min(cols, 9000 // n))
            elif EXOTIC and n <= 6 and n_samples <= 129 and rng.rand() < 0.04:
                cols = int(rng.choice([600, 999, 1500]))   # long blocks: rows longer than every staging area
            elif n * cols > 4000:
                cols = max(9, 4000 // n)
            if n > 32 and not wide:
                cols = min(cols, 45)
            b = synth_block(rng, n, cols, index=i, gaps=bool(rng.rand() < 0.8), mean_branch=float(rng.choice([0.01, 0.1, 0.4]))).upper()
            if rng.rand() < 0.3:   # common gap stretch (ties) / zero-length branch (degenerate rows)
                rows = [bytearray(x.seq.encode()) for x in b.rows]
                ln = int(rng.choice([3, 4, 6, 7])); pos = int(rng.randint(0, max(1, cols - ln)))
                for q in range(1, n):
                    rows[q][pos:pos + ln] = b"-" * min(ln, cols - pos)
                for x, row in zip(b.rows, rows):
                    x.seq = row.decode(); x.length = sum(ch != "-" for ch in x.seq)
                m = list(re.finditer(r":[0-9.]+", b.tree))[0]
                b.tree = b.tree[:m.start()] + ":0.000000" + b.tree[m.end():]
            if rng.rand() < 0.2:   # ambiguity codes, N and U among the residues (score.c:255-280 counts them as index 0, :389-392 scores N codons as 0)
                for x in b.rows:
                    row = bytearray(x.seq.encode())
                    for q in np.flatnonzero(rng.rand(len(row)) < 0.04):
                        if row[q] != ord("-"):
                            row[q] = ord(str(rng.choice(list("NNNRYKMSWBDHVU"))))
                    x.seq = row.decode()
            if rng.rand() < 0.05:   # two or three of the four nucleotides only: zero frequencies -- NaN or infinite background models and branch
                keep = list(rng.permutation(list("ACGT"))[:int(rng.choice([2, 2, 3]))])   # matrices (G and T only: every simulated residue state 0, evolve.c:167-175)
                for x in b.rows:
                    x.seq = "".join(ch if ch in keep or ch not in "ACGT" else keep[(ord(ch) + q) % len(keep)] for q, ch in enumerate(x.seq))
            if EXOTIC:
                exotic(rng, b)
            blocks.append(b)
        stop = rng.rand() < 0.2
        if stop:
            pars.update(stopEarly=1, cutoff=float(rng.choice([0.05, 0.2, 0.5])))
        hp = api.default_params(sampleN=n_samples, seed_base=seed, **pars)
        # a third of the rounds go through a stream of sub-batches (rc_stream_*) instead of one batch
        if rng.rand() < 0.33:
            m = api.Marshalled(blocks)
            m.set_trees()
            parts = list(api.score_stream(ctx, m, hp, int(rng.randint(1, 8)), depth=int(rng.randint(1, 4))))
        else:
            parts = [api.Batch(ctx, blocks, hp).run()]
        where = [(pb, i) for pb in parts for i in range(pb.n)]
        for (batch, i), b in zip(where, blocks):
            p = ob.default_params(n_samples, blosum=pars.get("blosum", 62))
            for k, v in pars.items():
                if k != "blosum":
                    setattr(p, k, v)
            if batch.status(i) != 0:
                continue
            res = ob.run_block([x.seq.upper() for x in b.rows], [x.name for x in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, p, seed)   # (the readers upper-case what they read, the library what it is given)
            rc0 = batch.getExtremeValuePars(i)[0]
            # --stop-early: a decided block (rc -1) may have fewer samples simulated here than the reference had when it left its loop
            why = []
            ok = np.array_equal(batch.maxScores(i), np.float32(res.maxScores)) or (stop and rc0 == -1 and res.evd_rc == -1)
            if not ok:
                why.append("maxima")
            got, want = sorted(batch.scoreAln(i), key=key), sorted(res.hss, key=key)
            same = [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
                [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
            if not same:
                why.append("hss")
            ok = ok and same
            rc, mu, lam = batch.getExtremeValuePars(i)
            same = rc == res.evd_rc and (rc != 1 or (abs(mu - res.mu) <= 1e-6 * max(1, abs(res.mu)) and abs(lam - res.lam) <= 1e-6 * max(1, abs(res.lam))))
            if not same:
                why.append(f"fit rc {rc} / {res.evd_rc}, mu {mu!r} / {res.mu!r}, lambda {lam!r} / {res.lam!r}")
            ok = ok and same
            if ok and got and rng.rand() < 0.15:   # state path of the best HSS: rc_batch_backtrack against the oracle's backtrack
                h = got[0]
                strand = 0 if h["strand"] == "+" else 1
                st, z, tr = batch.backtrack(i, strand, h["start"], h["end"])
                rows, names = [x.seq.upper() for x in b.rows], [x.name for x in b.rows]
                bl = pars.get("blosum", 62)
                m = ob.get_models(b.tree, rows, names, b.kappa, bl)
                rrows = ob.rev_aln(rows)
                mr = ob.get_models(b.tree, rrows, names, b.kappa, bl)
                _h, skf, skr = ob.score_aln(rows, b.rows[0].start, b.rows[0].length, m, mr, p, want_sk=True)
                ws, wz, wt = ob.backtrack(h["start"], h["end"], skf if strand == 0 else skr, rows if strand == 0 else rrows, p)
                idx = list(range(h["end"], h["start"] + 1, -3))
                for k in range(1, b.n):
                    ok = ok and list(st[k, idx]) == list(ws[k, idx]) and list(z[k, idx]) == list(wz[k, idx]) and list(tr[k, idx]) == list(wt[k, idx])
                paths += 1
            blocks_done += 1
            if not ok:
                bad += 1
                log(f"MISMATCH round {rounds} block {i}: {b.n} x {b.cols}, n={n_samples}, seed={seed}, pars={pars}: {'; '.join(why) or 'backtrack path'}"
                    f" (batch of {len(blocks)}, {'stream' if len(parts) > 1 or parts[0].n != len(blocks) else 'one batch'})")
                import json, os   # the block, to replay it: tools/soak_replay.py
                os.makedirs("gpurun_out/soak_fail", exist_ok=True)
                with open(f"gpurun_out/soak_fail/{mode or 'any'}_{rng_seed}_{rounds}_{i}.json", "w") as fh:
                    json.dump(dict(rows=[x.seq for x in b.rows], names=[x.name for x in b.rows], start=b.rows[0].start, length=b.rows[0].length, tree=b.tree,
                                   kappa=b.kappa, n_samples=n_samples, seed=seed, pars=pars, why=why,
                                   batch=[dict(rows=[x.seq for x in bb.rows], names=[x.name for x in bb.rows], start=bb.rows[0].start, length=bb.rows[0].length,
                                               tree=bb.tree, kappa=bb.kappa) for bb in blocks]), fh)
        for pb in parts:
            pb.close()
    ctx.close()
    log(f"soak{' ' + mode if mode else ''}: {rounds} batches, {blocks_done} blocks compared in {time.time() - t0:.0f} s, {bad} mismatches; {paths} backtrack paths compared")
    return rounds, blocks_done, bad, paths


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    return 1 if soak(budget, seed, mode)[2] else 0


if __name__ == "__main__":
    sys.exit(main())
