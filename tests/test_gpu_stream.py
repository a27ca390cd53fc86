"""The asynchronous forms of the hot path (rc_batch_run_async / rc_batch_wait, rc_stream_*): several batches in
flight on alternating HIP streams, host preparation of one batch overlapping the kernels of another, recycled
buffers.  Everything they return must equal what one synchronous batch returns, bit for bit -- and, at BASELINE
configs[2] size (10 000 blocks x 120 cols x 6 seqs, n = 1000), the oracle on a random sub-sample."""
import numpy as np
import pytest

from conftest import hss_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rnacode_amd import api
    c = api.Context(0)
    yield c
    c.close()


def mixed_blocks():
    from rnacode_amd.synth import synth_blocks
    blocks = (synth_blocks(40, 6, 120, seed=21) + synth_blocks(9, 4, 45, seed=22) + synth_blocks(7, 9, 150, seed=23)
              + synth_blocks(5, 3, 30, seed=24) + synth_blocks(3, 12, 210, seed=25))
    rng = np.random.RandomState(5)
    order = rng.permutation(len(blocks))
    return [blocks[i].upper() for i in order]


def collect(batch):
    return batch.maxScores_all(), batch.fits(), batch.scoreAln_all(), [batch.status(i) for i in range(batch.n)]


def test_stream_of_sub_batches_equals_one_batch(ctx):
    from rnacode_amd import api
    blocks = mixed_blocks()
    p = api.default_params(sampleN=200, seed_base=77)
    one = api.Batch(ctx, blocks, p).run()
    want_max, want_fit, want_hss, want_st = collect(one)
    one.close()
    m = api.Marshalled(blocks)
    m.set_trees()
    for sub, depth in ((7, 3), (16, 2), (64, 1), (1, 4)):
        stream = api.Stream(ctx, p, depth)
        for _pass in range(2):   # the second pass runs on recycled batches
            at = 0
            for batch in api.score_stream(ctx, m, p, sub, stream=stream):
                gm, gf, gh, gs = collect(batch)
                n = batch.n
                np.testing.assert_array_equal(gm, want_max[at:at + n])
                np.testing.assert_array_equal(gf[:, :1], want_fit[at:at + n, :1])
                np.testing.assert_allclose(gf[:, 1:3], want_fit[at:at + n, 1:3], rtol=1e-6)
                assert gh == want_hss[at:at + n] or all(
                    [{k: v for k, v in h.items() if k != "pvalue"} for h in a] == [{k: v for k, v in h.items() if k != "pvalue"} for h in b]
                    for a, b in zip(gh, want_hss[at:at + n]))
                assert gs == want_st[at:at + n]
                assert batch.clamped() <= 3   # draws past the cumulative vector: ~1e-7 per root draw
                at += n
                batch.close()
            assert at == len(blocks)
        stream.close()


def test_batches_in_flight_together(ctx):
    """Three resident batches between run_async and wait at once (alternating compute streams, own scratch and work
    queues each), twice, against their synchronous results."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    p = api.default_params(sampleN=320, seed_base=9)
    sets = [[b.upper() for b in synth_blocks(30, 6, 120, seed=s)] for s in (31, 32, 33)]
    want = []
    for bl in sets:
        b = api.Batch(ctx, bl, p).run()
        want.append(collect(b))
        b.close()
    batches = [api.Batch(ctx, bl, p) for bl in sets]
    for _ in range(2):
        for b in batches:
            b.run_async()
        for b, w in zip(batches, want):
            b.wait()
            got = collect(b)
            np.testing.assert_array_equal(got[0], w[0])
            np.testing.assert_array_equal(got[1][:, 0], w[1][:, 0])
            assert [[hss_key(h) for h in x] for x in got[2]] == [[hss_key(h) for h in x] for x in w[2]]
    for b in batches:
        b.close()


def test_stream_with_rejected_and_skipped_blocks(ctx):
    from rnacode_amd import api
    from rnacode_amd.alnio import AlnBlock, AlnRow
    from rnacode_amd.synth import synth_blocks
    good = [b.upper() for b in synth_blocks(6, 5, 60, seed=3)]
    short = AlnBlock([AlnRow("a", "AT"), AlnRow("b", "AT"), AlnRow("c", "AT")], "s", None, None)
    two = AlnBlock([AlnRow("a", "ATGGCTAAA"), AlnRow("b", "ATGGCAAAA")], "t", None, None)
    notree = AlnBlock([AlnRow("a", "ATGGCTAAA"), AlnRow("b", "ATGGCAAAA"), AlnRow("c", "ATGGCTAAG")], "n", None, None)
    blocks = [good[0], short, good[1], two, notree, good[2], good[3], notree, good[4], good[5]]
    p = api.default_params(sampleN=100, seed_base=4)
    ref = api.Batch(ctx, good, p).run()
    want = ref.maxScores_all()
    ref.close()
    m = api.Marshalled(blocks)
    m.set_trees(strict=False)
    rows, status = [], []
    for batch in api.score_stream(ctx, m, p, 3, depth=2):
        mx = batch.maxScores_all()
        for i in range(batch.n):
            status.append(batch.status(i))
            if batch.status(i) == api.RC_OK:
                rows.append(mx[i])
            else:
                assert (mx[i] == -1.0).all()
        batch.close()
    assert status == [api.RC_OK, api.RC_ERR_SKIP, api.RC_OK, api.RC_ERR_SKIP, api.RC_ERR_ARG, api.RC_OK, api.RC_OK, api.RC_ERR_ARG,
                      api.RC_OK, api.RC_OK]
    np.testing.assert_array_equal(np.stack(rows), want)


def test_config3_full_size_stream_and_resident_against_oracle(ctx):
    """BASELINE configs[2] as specified: 10 000 blocks x 120 cols x 6 seqs, n = 1000, once as one resident batch
    (160 000 work items over the eight XCD queues) and once as a stream of eight sub-batches; both must agree
    everywhere, and 32 randomly chosen blocks x the first 64 samples must equal the CPU oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = [b.upper() for b in synth_blocks(10000, 6, 120, seed=1)]
    p = api.default_params(sampleN=1000, seed_base=42)
    m = api.Marshalled(blocks)
    m.set_trees()
    big = api.Batch(ctx, m, p).run()
    allmax = big.maxScores_all()
    fits = big.fits()
    assert big.clamped() <= 100     # about 1e-7 per root draw
    assert allmax.shape == (10000, 1000)
    assert np.isfinite(allmax).all() and ((allmax > 0) | (allmax == -1.0)).all()
    at = 0
    for batch in api.score_stream(ctx, m, p, 1250, depth=3):
        np.testing.assert_array_equal(batch.maxScores_all(), allmax[at:at + batch.n])
        np.testing.assert_array_equal(batch.fits()[:, 0], fits[at:at + batch.n, 0])
        np.testing.assert_allclose(batch.fits()[:, 1:3], fits[at:at + batch.n, 1:3], rtol=1e-6)
        at += batch.n
        batch.close()
    assert at == 10000
    rng = np.random.RandomState(123)
    picks = sorted(rng.choice(10000, 32, replace=False).tolist())
    for i in picks:
        b = blocks[i]
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa,
                           ob.default_params(64), 42)
        np.testing.assert_array_equal(allmax[i, :64], np.float32(res.maxScores))
        got = sorted(big.scoreAln(i), key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
               [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]
    # batch-composition invariance: the picked blocks scored alone as one small batch
    small = api.Batch(ctx, [blocks[i] for i in picks], p).run()
    np.testing.assert_array_equal(small.maxScores_all(), allmax[picks])
    small.close()
    big.close()


def test_config5_one_gpus_share_streamed_against_oracle(ctx):
    """BASELINE configs[4] at ONE GPU's share of eight: 12 500 blocks x 300 cols x 12 seqs at n = 10 000 (1.25e8 sample-alignments,
    about 8 s of GPU) through rc_stream_* in sub-batches of 1000.  Checked: every block fitted; a 1000-block slice scored as one
    resident batch gives the stream's maxima and fits; sample-range sharding by seed (score.c:1004-1048 with seed_base + s) reproduces
    the full rows; eight random blocks x their first 64 samples and their HSS tables equal the CPU oracle.  (The two reference-scored
    blocks of this shape are the golden synth_12x300_n40, test_gpu_parity.)"""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    nb, n, seed = 12500, 10000, 42
    blocks = [b.upper() for b in synth_blocks(nb, 12, 300, seed=1)]
    p = api.default_params(sampleN=n, seed_base=seed)
    m = api.Marshalled(blocks)
    m.set_trees()
    rng = np.random.RandomState(321)
    picks = sorted(rng.choice(nb, 8, replace=False).tolist())
    lo_s, hi_s = 3000, 4000                      # the slice that is also scored as one resident batch
    in_slice = [3100, 3555, 3999]
    keep_max, keep_hss = {}, {}
    fits = np.zeros((nb, 4), dtype=np.float32)
    slice_max = np.zeros((hi_s - lo_s, n), dtype=np.float32)
    clamped = 0
    at = 0
    for batch in api.score_stream(ctx, m, p, 1000, depth=3):
        fits[at:at + batch.n] = batch.fits()
        clamped += batch.clamped()
        for i in picks + in_slice:
            if at <= i < at + batch.n:
                keep_max[i] = batch.maxScores(i - at)
                keep_hss[i] = batch.scoreAln(i - at)
        a, b = max(at, lo_s), min(at + batch.n, hi_s)
        if a < b:
            slice_max[a - lo_s:b - lo_s] = batch.maxScores_all()[a - at:b - at]
        at += batch.n
        batch.close()
    assert at == nb
    assert clamped <= 2000                       # about 1e-7 per root draw: 1.25e8 x 300 root draws
    assert (fits[:, 0] == 1).all()               # every block scored and fitted (no stop-early here)
    assert np.isfinite(fits[:, 1:3]).all() and (fits[:, 2] > 0).all()
    # ---- the 1000-block slice as one resident batch
    part = api.Batch(ctx, blocks[lo_s:hi_s], p).run()
    np.testing.assert_array_equal(part.maxScores_all(), slice_max)
    np.testing.assert_array_equal(part.fits()[:, 0], fits[lo_s:hi_s, 0])
    np.testing.assert_allclose(part.fits()[:, 1:3], fits[lo_s:hi_s, 1:3], rtol=1e-6)
    for i in in_slice:
        assert part.scoreAln(i - lo_s) == keep_hss[i]
    part.close()
    # ---- sample-range sharding: samples [0, 4992) and [4992, 10000) as two runs, seeds seed and seed + 4992
    some = [blocks[i] for i in picks[:4]]
    cut = 4992
    first = api.Batch(ctx, some, api.default_params(sampleN=cut, seed_base=seed)).run()
    second = api.Batch(ctx, some, api.default_params(sampleN=n - cut, seed_base=seed + cut)).run()
    for t, i in enumerate(picks[:4]):
        np.testing.assert_array_equal(np.concatenate([first.maxScores(t), second.maxScores(t)]), keep_max[i])
    first.close(); second.close()
    # ---- the CPU oracle on eight random blocks: the first 64 samples and the native HSS table
    for i in picks:
        b = blocks[i]
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa,
                           ob.default_params(64), seed)
        np.testing.assert_array_equal(keep_max[i][:64], np.float32(res.maxScores))
        got = sorted(keep_hss[i], key=hss_key)
        want = sorted(res.hss, key=hss_key)
        assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in got] == \
               [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want]


def test_stream_plan_sizes(ctx):
    """rc_stream_plan: the sizes sum to the number of blocks; after the first, every sub-batch fills the chip's wavefront slots a
    whole number of times (r = slots / sample groups blocks do so once) and none exceeds eight rounds -- except the last, which takes a
    tail of less than half a sub-batch along --; short inputs are one batch;
    and streaming by the plan gives the results of one resident batch."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    p = api.default_params(sampleN=1000)
    for n in (1, 33, 300, 1250, 10000, 123457):
        sizes = api.stream_plan(ctx, p, n, 1)
        assert sum(sizes) == n and all(x > 0 for x in sizes)
        if len(sizes) > 1:
            r = sizes[1]
            assert all(x % r == 0 and x <= 8 * r for x in sizes[1:-1]) and sizes[-1] < 12 * r
    assert api.stream_plan(ctx, p, 300, 1) == [300]
    assert len(api.stream_plan(ctx, p, 10000, 10)) < len(api.stream_plan(ctx, p, 10000, 1))   # more row-count classes: larger sub-batches
    blocks = [b.upper() for b in synth_blocks(700, 5, 60, seed=21)]
    p = api.default_params(sampleN=1000, seed_base=7)
    m = api.Marshalled(blocks)
    m.set_trees()
    whole = api.Batch(ctx, m, p).run()
    want = whole.maxScores_all()
    got = []
    plan = api.stream_plan(ctx, p, len(blocks), 1)
    assert len(plan) >= 2
    for sb in api.score_stream(ctx, m, p):
        got.append(sb.maxScores_all())
        sb.close()
    np.testing.assert_array_equal(np.concatenate(got), want)
    whole.close()


def test_native_scan_by_lanes_equals_the_fused_scan_and_the_oracle():
    """Batches of >= 64 blocks per class leave getHSS of the native block to k_native_scan (one lane per strand x frame matrix); smaller
    ones keep it inside k_native_dp (one wavefront per matrix).  The same blocks as one large batch and in slices of 40 give the same
    HSS tables -- coding-like blocks with many positive entries, random ones, different lengths in one class -- equal to the oracle's on
    a sample of the blocks."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block, synth_blocks
    rng = np.random.RandomState(77)
    blocks = [b.upper() for b in synth_blocks(150, 5, 90, seed=31)]
    blocks += [synth_block(rng, 5, int(rng.choice([30, 45, 60, 120, 150])), index=i, gaps=bool(i % 3), mean_branch=float(rng.choice([0.01, 0.05, 0.4]))).upper()
               for i in range(150)]                                   # short branches: conserved, long positive stretches
    blocks += [b.upper() for b in synth_blocks(80, 9, 60, seed=32)]    # a second class in the same batch
    p = api.default_params(sampleN=64, seed_base=3)
    ctx = api.Context(0)

    def tables(sub):
        batch = api.Batch(ctx, sub, p).run()
        out = [[(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in sorted(batch.scoreAln(i), key=hss_key)] for i in range(len(sub))]
        batch.close()
        return out

    lanes = tables(blocks)
    fused = [t for lo in range(0, len(blocks), 40) for t in tables(blocks[lo:lo + 40])]
    ctx.close()
    assert lanes == fused
    assert sum(len(t) for t in lanes) > 5 * len(blocks)
    op = ob.default_params(64)
    for i in list(range(0, len(blocks), 23)):
        b = blocks[i]
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, op, 3)
        want = sorted(res.hss, key=hss_key)
        assert lanes[i] == [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"])) for h in want], i


def test_stop_early_runs_fewer_samples_and_agrees_with_the_oracle(monkeypatch):
    """--stop-early (score.c:992,1036-1042): blocks whose first samples already beat the native score more often
    than the cutoff allows are decided (rc = -1) and get no further samples; all others are sampled in full.
    Decisions, fits and HSS equal the oracle's, which runs the reference's loop literally."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    # the two-round schedule is only used for batches that fill the chip several times; let this small one take it
    monkeypatch.setenv("RC_STOP_MIN_ITEMS", "1")
    ctx = api.Context(0)
    blocks = [b.upper() for b in synth_blocks(40, 5, 90, seed=8)]
    n, seed, cutoff = 640, 42, 0.05
    p = api.default_params(sampleN=n, seed_base=seed, stopEarly=1, cutoff=cutoff)
    batch = api.Batch(ctx, blocks, p).run()
    first_fits = batch.fits()
    batch.run()      # a second run of the same batch starts from clean stop marks
    np.testing.assert_array_equal(batch.fits(), first_fits)
    mx = batch.maxScores_all()
    full = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed, cutoff=cutoff)).run()
    fmx = full.maxScores_all()
    op = ob.default_params(n)
    op.stopEarly, op.cutoff = 1, cutoff
    decided = 0
    first = ((int(cutoff * n) + 1 + 64 + 63) // 64) * 64
    for i, b in enumerate(blocks):
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, op, seed)
        rc, mu, lam = batch.getExtremeValuePars(i)
        assert rc == res.evd_rc
        np.testing.assert_array_equal(mx[i, :first], fmx[i, :first])
        if rc == 1:
            np.testing.assert_array_equal(mx[i], fmx[i])
            assert abs(mu - res.mu) <= 1e-6 * max(1, abs(res.mu)) and abs(lam - res.lam) <= 1e-6
        elif (mx[i, first:] == -1.0).all() and (fmx[i, first:] > 0).any():
            decided += 1
        assert [hss_key(h)[:5] for h in sorted(batch.scoreAln(i), key=hss_key)] == [hss_key(h)[:5] for h in sorted(full.scoreAln(i), key=hss_key)]
    assert decided > 0    # random (non-coding) blocks: most are decided after the first round
    batch.close(); full.close()
    ctx.close()


def test_sample_range_sharding_on_the_hip_path(ctx):
    """BASELINE configs[1] (genomic-preprocessed.maf, n = 1000): few blocks, so the SAMPLE range is what gets split
    (SURVEY 8e).  Three 'ranks' run one after the other on the one GPU of the test box, each through
    hip_sample_sharded with a gather that concatenates their slices; the result must equal the reference's goldens."""
    from conftest import load_golden
    from helpers import block_from_golden, close_p
    from rnacode_amd import api
    from rnacode_amd import distributed as rd
    doc = load_golden("genomic_preprocessed_n1000")
    entries = doc["blocks"][:12]
    blocks = [block_from_golden(e) for e in entries]
    n, world = doc["samples"], 3
    slices = {}

    def score_only(rank):
        def gather(local):
            slices[rank] = local
            raise StopIteration
        try:
            rd.hip_sample_sharded(ctx, blocks, rank, world, gather, sampleN=n, seed_base=doc["seed_base"])
        except StopIteration:
            pass

    for r in range(world):
        score_only(r)
    full = np.concatenate([slices[r] for r in range(world)], axis=1)
    res = rd.hip_sample_sharded(ctx, blocks, 1, world, lambda local: full, sampleN=n, seed_base=doc["seed_base"])
    for e, got in zip(entries, res):
        ref = e["ref"]
        if "skipped" in ref:
            assert got is None
            continue
        rc, mu, lam, hss = got
        assert rc == ref["evd_rc"]
        assert abs(mu - ref["mu"]) <= 1e-6 * max(1, abs(ref["mu"])) and abs(lam - ref["lambda"]) <= 1e-6
        gh, wh = sorted(hss, key=hss_key), sorted(ref["hss"], key=hss_key)
        assert [(np.float32(h["score"]),) + hss_key(h)[1:5] for h in gh] == [(np.float32(h["score"]),) + hss_key(h)[1:5] for h in wh]
        for a, b in zip(gh, wh):
            assert close_p(a["pvalue"], b["pvalue"])
    rows = [i for i, e in enumerate(entries) if "skipped" not in e["ref"] and e["ref"]["maxScores"]]
    for i in rows:
        np.testing.assert_array_equal(full[i], np.float32(entries[i]["ref"]["maxScores"]))


def _bench(tmp_path, gpus, extra=(), env_extra=None):
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT
    dump = str(tmp_path / f"g{gpus}.npy")
    full = str(tmp_path / f"g{gpus}_full.json")
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    args = ["--blocks", "50", "--samples", "192"] if "--blocks" not in extra else []
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), *args, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--sub-batches", "3", "--dump-gathered", dump, "--full-json", full, *extra],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return _bench_record(r.stdout, full), np.load(dump)


def _bench_record(stdout, full):
    """bench.py prints ONE compact line (the driver keeps 2000 characters of output) and writes the full record to --full-json: the
    line must parse, carry the contract's keys and fit; the tests read the record."""
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    line = json.loads(lines[0])
    assert len(lines[0]) <= 1950 and lines[0].startswith('{"metric"')
    with open(full) as fh:
        rec = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line and (line[k] == rec[k] or abs(line[k] - rec[k]) <= 1e-3 * abs(rec[k])), k
    assert "roofline" in line and "config" in line and "legs" in line
    return rec


def test_bench_starts_its_own_ranks_and_two_hip_ranks_gather_the_single_rank_result(tmp_path):
    """`bench.py --gpus 2` without torchrun launches two ranks itself (before touching the GPU); both run the HIP path
    on the one GPU of the test box (RC_BENCH_TEST_BACKEND=gloo: the collective goes over gloo, everything else is the
    production code) and the gathered maxima equal a single rank's."""
    one, m1 = _bench(tmp_path, 1)
    two, m2 = _bench(tmp_path, 2, env_extra={"RC_BENCH_TEST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["blocks_per_gpu"] == 25 and "stream" in two and two["stream"]["value"] > 0
    assert two["weak_scaling"]["blocks_per_gpu"] == 50 and two["weak_scaling"]["value"] > 0
    assert m1.shape == (50, 192)
    np.testing.assert_array_equal(m1, m2)
    assert two["n_ranks_seen"] == 2 and len(two["devices"]["ranks"]) == 2 and two["config"]["exchange_ab"]["chosen"] == "synchronous"
    fits, mf = _bench(tmp_path, 2, extra=("--exchange", "fits"), env_extra={"RC_BENCH_TEST_BACKEND": "gloo"})
    assert mf.shape == (50, 4) and (mf[:, 0] == 1).all()
    # N > 1 also times the native driver's own --gpus N mode on the whole input (rank 0, the others wait on the host)
    line, _ = _bench(tmp_path, 2, env_extra={"RC_BENCH_TEST_BACKEND": "gloo", "RC_BENCH_TEST_NATIVE_MULTI": "1"})
    nm = line["native_driver_multi_gpu"]
    assert nm["gpus_1"]["value"] > 0 and nm["gpus_2"]["value"] > 0 and nm["gpus_2"]["hss_lines"] == nm["gpus_1"]["hss_lines"] > 0, nm
    # the overlapped order and the warm-up comparison of the two orders with two ranks (the collective staged through host copies over gloo)
    hook = {"RC_BENCH_TEST_BACKEND": "gloo", "RC_BENCH_TEST_OVERLAP_ON_CPU": "1"}
    for extra, want in (("--overlap",), "overlapped"), (("--no-overlap",), "synchronous"), ((), None):
        line, mo = _bench(tmp_path, 2, extra=extra, env_extra=hook)
        ab = line["config"]["exchange_ab"]
        if want is None:
            assert ab["chosen"] in ("overlapped", "synchronous") and ab["ms_per_step_overlapped"] > 0 and ab["ms_per_step_synchronous"] > 0
        else:
            assert ab["chosen"] == want
        np.testing.assert_array_equal(m1, mo)


def test_eight_ranks_on_one_gpu_gather_the_single_rank_result(tmp_path):
    """The first run on an 8-GPU node made boring: `bench.py --gpus 8` with all eight ranks on device 0
    (the collective over gloo, everything else the production code), once with a block count that divides by eight and once
    ragged; the gathered maxima equal the single rank's, and every rank is seen."""
    for nblocks in (64, 75):
        args = ("--blocks", str(nblocks), "--samples", "128", "--no-weak")
        one, m1 = _bench(tmp_path, 1, extra=args)
        eight, m8 = _bench(tmp_path, 8, extra=args, env_extra={"RC_BENCH_TEST_BACKEND": "gloo"})
        assert eight["n_gpus"] == 8 and eight["n_ranks_seen"] == 8 and len(eight["devices"]["ranks"]) == 8
        assert m1.shape == (nblocks, 128)
        np.testing.assert_array_equal(m1, m8)
        assert eight["value"] > 0 and eight["config"]["exchange_ab"]["chosen"] == "synchronous"


def test_eight_ranks_on_the_long_block_config_and_the_automatic_exchange(tmp_path):
    """`bench.py --config c5 --gpus 8` (the shape of BASELINE configs[4], 64 blocks of it, all eight ranks on device 0): the gathered maxima equal
    a single rank's; every rank's host-inclusive stream is also measured as ONE continuous stream with an exchange per pass (round 6); and
    `--exchange auto` gathers the maxima while a rank's share is small and switches to the 16-byte fit rows above the limit (64 MB per rank and
    step: configs[4] whole is 500 MB per rank; lowered here through the test hook)."""
    args = ("--config", "c5", "--blocks", "64", "--samples", "256", "--no-weak")
    one, m1 = _bench(tmp_path, 1, extra=args)
    eight, m8 = _bench(tmp_path, 8, extra=args, env_extra={"RC_BENCH_TEST_BACKEND": "gloo"})
    assert eight["n_gpus"] == 8 and eight["n_ranks_seen"] == 8 and m1.shape == (64, 256)
    np.testing.assert_array_equal(m1, m8)
    assert "all-gather of maxima (auto)" in eight["config"]["exchange"]
    assert eight["stream"]["continuous"]["ms_per_step"] > 0 and eight["stream"]["ms_per_step"] > 0
    fits, mf = _bench(tmp_path, 8, extra=args, env_extra={"RC_BENCH_TEST_BACKEND": "gloo", "RC_BENCH_EXCHANGE_AUTO_MB": "0.001"})
    assert "all-gather of fits (auto)" in fits["config"]["exchange"] and mf.shape == (64, 4) and (mf[:, 0] == 1).all()


def test_bench_survives_a_failing_collective_library(tmp_path):
    """If the RCCL path raises -- at init, in the warm-up comparison of the two exchange orders, or in a gather -- bench.py falls back
    to the synchronous exchange through host memory, records the exception text in config.exchange_ab and still prints its line with
    exit code 0 (RC_BENCH_TEST_FAIL_EXCHANGE injects the failure)."""
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT
    one, m1 = _bench(tmp_path, 1, extra=("--blocks", "40", "--samples", "128"))
    for where in ("overlap", "gather"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, RC_BENCH_FORCE_DIST="1", RC_BENCH_TEST_FAIL_EXCHANGE=where)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        dump = str(tmp_path / ("fail_%s.npy" % where))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "40", "--samples", "128",
                            "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sub-batches", "2", "--dump-gathered", dump,
                            "--full-json", dump + ".json"],
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        rec = _bench_record(r.stdout, dump + ".json")
        ab = rec["config"]["exchange_ab"]
        assert "error" in ab and "injected" in ab["error"] and ab["chosen"] == "synchronous (host fallback)", ab
        np.testing.assert_array_equal(np.load(dump), m1)


def test_bench_refuses_a_world_that_does_not_match_gpus(tmp_path):
    import os
    import subprocess
    import sys

    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--blocks", "8", "--samples", "64", "--steps", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_under_torchrun_goes_through_rccl(tmp_path):
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with the one GPU a test
    box has: RC_BENCH_FORCE_DIST=1 takes the multi-rank path with a world of one, so init_process_group("nccl"), the all-gather
    of the kernel-written tensor (resident and stream mode), the reductions and the barriers all run through RCCL."""
    import json
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RC_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one, m1 = _bench(tmp_path, 1, extra=("--blocks", "40", "--samples", "128"))
    # --overlap: step i's all-gather overlapped with step i + 1 (rnacode_amd.distributed.OverlappedGather); --no-overlap: one after the
    # other; neither: both orders timed during the warm-up, the faster one runs the timed region (config.exchange_ab says which)
    for extra, overlapped in (("--overlap",), True), (("--no-overlap",), False), ((), None):
        dump = str(tmp_path / ("rccl_%s.npy" % overlapped))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "40", "--samples", "128",
                            "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sub-batches", "2", "--dump-gathered", dump,
                            "--full-json", dump + ".json", *extra],
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = _bench_record(r.stdout, dump + ".json")
        ab = line["config"]["exchange_ab"]
        if overlapped is None:
            assert ab["ms_per_step_synchronous"] > 0 and ab["ms_per_step_overlapped"] > 0 and ab["steps_each"] >= 3
            overlapped = ab["chosen"] == "overlapped"
            assert overlapped == (ab["ms_per_step_overlapped"] < ab["ms_per_step_synchronous"])
        else:
            assert ab["chosen"] == ("overlapped" if overlapped else "synchronous") and "forced" in ab
        assert line["config"]["exchange"].startswith("all-gather of maxima") and ("overlapped" in line["config"]["exchange"]) == overlapped
        assert line["n_gpus"] == 1 and line["stream"]["value"] > 0
        # the line proves which devices the ranks sat on, and the RCCL channel cap is recorded
        assert line["n_ranks_seen"] == 1 and line["devices"]["distinct_devices"] == 1 and len(line["devices"]["ranks"]) == 1
        assert line["config"]["rccl_env"]["NCCL_MAX_NCHANNELS"] == "2"
        np.testing.assert_array_equal(np.load(dump), m1)


def test_two_contexts_on_two_host_threads_do_not_interfere():
    """The library keeps no state outside rc_ctx / rc_batch (the last-error text is per thread): two host threads, each with its own context
    on the same device, score different batches at the same time; every result equals the one a single context gives."""
    import threading
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    sets = [[b.upper() for b in synth_blocks(60, 5 + t, 60 + 30 * t, seed=50 + t)] for t in range(2)]
    pars = [api.default_params(sampleN=130 + 70 * t, seed_base=7 + t) for t in range(2)]
    ctx = api.Context(0)
    want = []
    for blocks, p in zip(sets, pars):
        batch = api.Batch(ctx, blocks, p).run()
        want.append([(batch.maxScores(i).copy(), batch.getExtremeValuePars(i)) for i in range(batch.n)])
        batch.close()
    ctx.close()
    got = [None, None]
    errors = []

    def worker(t):
        try:
            c = api.Context(0)
            for _ in range(3):
                batch = api.Batch(c, sets[t], pars[t]).run()
                got[t] = [(batch.maxScores(i).copy(), batch.getExtremeValuePars(i)) for i in range(batch.n)]
                batch.close()
            c.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        assert len(got[t]) == len(want[t])
        for (gm, gf), (wm, wf) in zip(got[t], want[t]):
            np.testing.assert_array_equal(gm, wm)
            assert gf == wf


def test_many_row_count_classes_side_by_side_equal_one_class_at_a_time():
    """A batch with six row-count classes and enough items that its class launches run side by side on separate streams without the
    small-launch split: every block's maxima, fit and HSS equal those of single-class batches (one launch at a time,
    split or not as their size asks), and three of them equal the oracle's."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(31)
    blocks = [synth_block(rng, int(rng.randint(3, 9)), int(rng.randint(45, 121)), index=i, gaps=True).upper() for i in range(420)]
    n, seed = 1000, 77
    p = api.default_params(sampleN=n, seed_base=seed)
    ctx = api.Context(0)
    mixed = api.Batch(ctx, blocks, p).run()
    by_n = {}
    for i, b in enumerate(blocks):
        by_n.setdefault(b.n, []).append(i)
    assert len(by_n) == 6
    for rows, idx in by_n.items():
        solo = api.Batch(ctx, [blocks[i] for i in idx], p).run()
        for j, i in enumerate(idx):
            np.testing.assert_array_equal(mixed.maxScores(i), solo.maxScores(j))
            assert mixed.getExtremeValuePars(i) == solo.getExtremeValuePars(j)
            assert mixed.scoreAln(i) == solo.scoreAln(j)
        solo.close()
    for i in (5, 211, 419):
        b = blocks[i]
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), seed)
        np.testing.assert_array_equal(mixed.maxScores(i), np.float32(res.maxScores))
    mixed.close()
    ctx.close()


def test_stream_of_mixed_sub_batches_equals_one_batch():
    """Sub-batches with six row-count classes each, three of them in flight: their side-by-side class launches share the context's class
    streams and hardware queues.  Every block's maxima and fit equal those of the whole input scored as one batch."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(32)
    blocks = [synth_block(rng, int(rng.randint(3, 9)), int(rng.randint(45, 100)), index=i, gaps=True).upper() for i in range(2400)]
    p = api.default_params(sampleN=300, seed_base=9)
    ctx = api.Context(0)
    whole = api.Batch(ctx, blocks, p).run()
    want_max = whole.maxScores_all().copy()
    want_fit = whole.fits().copy()
    whole.close()
    m = api.Marshalled(blocks)
    m.set_trees()
    st = api.Stream(ctx, p, 3)
    for _ in range(2):   # the second pass runs on recycled batches
        at = 0
        for part in api.score_stream(ctx, m, p, 600, depth=3, stream=st, ramp=False):
            np.testing.assert_array_equal(part.maxScores_all(), want_max[at:at + part.n])
            # (the fit of a sub-batch that finds the chip idle runs four wavefronts per block, the others one: another summation order)
            np.testing.assert_allclose(part.fits(), want_fit[at:at + part.n], rtol=1e-6, atol=1e-6)
            at += part.n
            part.close()
        assert at == len(blocks)
    st.close()
    ctx.close()


def test_tail_sharing_on_and_off_give_the_same_results(monkeypatch):
    """k_null's tail sharing: items taken near the end of the work queues are published and their strand x frame
    parts claimed one by one, by the owner and by workgroups that have run out of work; partial maxima meet in an atomic max.  Batches
    sized so that most or all items are shared (a little over one round of the chip's wavefront slots, several row-count classes side
    by side, --stop-early rounds) must give, bit for bit, what RC_TAIL_SHARING=0 gives -- maxima, fits, HSS tables -- as one batch and
    as a stream, and a sample of blocks must equal the CPU oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block, synth_blocks
    rng = np.random.RandomState(77)
    cases = [
        ("one class, 1.2 rounds", [b.upper() for b in synth_blocks(313, 6, 120, seed=3)], dict(sampleN=1000, seed_base=5)),
        ("six classes side by side", [synth_block(rng, int(rng.randint(3, 9)), int(rng.randint(40, 130)), index=i, gaps=True).upper()
                                      for i in range(900)], dict(sampleN=700, seed_base=9)),
        ("stop-early rounds", [b.upper() for b in synth_blocks(400, 5, 90, seed=8)], dict(sampleN=1000, seed_base=11, stopEarly=1, cutoff=0.05)),
    ]
    monkeypatch.setenv("RC_STOP_MIN_ITEMS", "1")
    got = {}
    for sharing in ("1", "0"):
        monkeypatch.setenv("RC_TAIL_SHARING", sharing)
        c = api.Context(0)
        for name, blocks, kw in cases:
            p = api.default_params(**kw)
            m = api.Marshalled(blocks)
            m.set_trees()
            b = api.Batch(c, m, p).run()
            res = [b.maxScores_all().copy(), b.fits().copy(), [b.scoreAln(i) for i in range(0, b.n, 37)]]
            b.close()
            streamed = []
            for sb in api.score_stream(c, m, p, [len(blocks) // 3, len(blocks) - len(blocks) // 3]):
                streamed.append(sb.maxScores_all().copy())
                sb.close()
            res.append(np.concatenate(streamed))
            got[(name, sharing)] = res
        c.close()
    for name, blocks, kw in cases:
        on, off = got[(name, "1")], got[(name, "0")]
        if not kw.get("stopEarly"):
            np.testing.assert_array_equal(on[0], off[0], err_msg=name)
            np.testing.assert_array_equal(on[3], off[3], err_msg=name + " (stream)")
            np.testing.assert_array_equal(on[0], on[3], err_msg=name + " (stream vs batch)")
        else:   # stopped blocks keep whatever samples their rounds reached; the fits' verdicts and the undecided blocks' rows must agree
            np.testing.assert_array_equal(on[1][:, 0], off[1][:, 0], err_msg=name)
            live = on[1][:, 0] == 1
            np.testing.assert_array_equal(on[0][live], off[0][live], err_msg=name)
        np.testing.assert_array_equal(on[1], off[1], err_msg=name + " (fits)")
        assert on[2] == off[2], name
    # and against the oracle: five blocks of the first case, their first 64 samples
    name, blocks, kw = cases[0]
    for i in (0, 77, 150, 233, 312):
        blk = blocks[i]
        want = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa,
                            ob.default_params(64), kw["seed_base"])
        np.testing.assert_array_equal(got[(name, "1")][0][i][:64], np.float32(want.maxScores))


def test_two_rows_per_pass_equal_one_row_per_pass_and_the_oracle():
    """k_null<N-1, true, false, DUAL>: rows a and a + 1 of S share their sigma look-ups, row a + 1's values wait in
    a 32-register buffer.  Batches of ONE row count each (3..6 rows: the classes that take the two-row kernel; a batch of such classes
    only also runs its native-block kernels first), gapped blocks of 30..156 columns -- up to 52 codon sites, so rows longer than the
    buffer (their first cells are made twice), rows whose first site is a frame-shift event (they go alone) and pairs that change
    their cell kind together all occur.  The same blocks in a batch that also holds a class of eight rows are scored by the one-row
    kernels (the two-row ones run only where nothing shares their SIMDs): same maxima, fits and HSS tables bit for bit, and a sample
    of blocks must equal the CPU oracle."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(123)
    cases = {n: [synth_block(rng, n, int(rng.randint(30, 157)), index=i, gaps=True).upper() for i in range(260)] for n in (3, 4, 5, 6)}
    other = [synth_block(rng, 8, 60, index=i, gaps=True).upper() for i in range(4)]
    kw = dict(sampleN=600, seed_base=17)   # 2600 (block, 64-sample) items: whole items, not their strand x frame parts
    got = {}
    c = api.Context(0)
    for n, blocks in cases.items():
        for dual in ("1", "0"):
            b = api.Batch(c, blocks + (other if dual == "0" else []), api.default_params(**kw)).run()
            m = len(blocks)
            got[(n, dual)] = (b.maxScores_all()[:m].copy(), b.fits()[:m].copy(), [b.scoreAln(i) for i in range(0, m, 11)], b.null_kernel())
            b.close()
    c.close()
    for n, blocks in cases.items():
        on, off = got[(n, "1")], got[(n, "0")]
        # the two instantiations really ran (three rows: two other sequences leave a quarter of the registers free, always two-row)
        assert on[3].endswith(", true, 0>") and off[3].endswith(", true, 0>" if n == 3 else ", false, 0>"), (on[3], off[3])
        np.testing.assert_array_equal(on[0], off[0], err_msg=f"{n} rows")
        np.testing.assert_array_equal(on[1], off[1], err_msg=f"{n} rows (fits)")
        assert on[2] == off[2]
        for i in (0, 141, 259):
            blk = blocks[i]
            want = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa,
                                ob.default_params(64), kw["seed_base"])
            np.testing.assert_array_equal(on[0][i][:64], np.float32(want.maxScores), err_msg=f"{n} rows, block {i}")


def test_tail_sharing_stress_every_item_late(monkeypatch):
    """k_null's tail sharing hands an owner's sigma codes to helper workgroups of the same XCD behind an agent-scope release (the
    documented cross-CU hand-off).  This is its tripwire.  One workgroup per CU (RC_GRID_CAP_PER_CU=1) and a little more than one item per workgroup, no strand x frame split:
    EVERY item is late, is published, and has its six parts claimed by its owner and by the workgroups that found the queues empty --
    on every launch, hundreds of launches, the staging scratch rewritten with other blocks' codes each time (a stream recycles its
    three batch objects; resident batches are re-run in turn), both code-word paths (LDS-staged and from L2).  Every launch must
    reproduce the maxima computed with sharing off, bit for bit; a stale code word shows as a wrong maximum."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    monkeypatch.setenv("RC_GRID_CAP_PER_CU", "1")
    monkeypatch.setenv("RC_SPLIT_FACTOR", "0")
    shapes = [(6, 120), (4, 90), (8, 150), (12, 100), (3, 60), (10, 201)]
    n = 320                                    # five sample groups x 60 blocks = 300 items on 256 workgroups
    sets = []
    for t, (rows, cols) in enumerate(shapes):
        m = api.Marshalled([b.upper() for b in synth_blocks(60, rows, cols, seed=900 + t)])
        m.set_trees()
        sets.append(m)
    p = api.default_params(sampleN=n, seed_base=3)
    monkeypatch.setenv("RC_TAIL_SHARING", "0")
    c0 = api.Context(0)
    want = []
    for m in sets:
        b = api.Batch(c0, m, p).run()
        want.append(b.maxScores_all().copy())
        b.close()
    c0.close()
    monkeypatch.setenv("RC_TAIL_SHARING", "1")
    launches = 0
    for release in ("agent", "agent, second context"):
        c = api.Context(0)
        st = api.Stream(c, p, 3)
        for rep in range(14):                                             # streamed: helpers leave at once (stealWait = 0)
            order = np.random.RandomState(rep).permutation(len(sets))
            pend = []

            def take():
                k = pend.pop(0)
                sb = st.next()
                np.testing.assert_array_equal(sb.maxScores_all(), want[k], err_msg=f"{release}, stream pass {rep}, shape {shapes[k]}")
                sb.close()

            for k in order:
                if st.pending >= st.depth:
                    take()
                    launches += 1
                st.submit(sets[k], 0, 60)
                pend.append(k)
            while pend:
                take()
                launches += 1
        st.close()
        batches = [api.Batch(c, m, p) for m in sets]                      # resident: helpers wait for owners that are still simulating
        for rep in range(12):
            for k, b in enumerate(batches):
                b.run()
                np.testing.assert_array_equal(b.maxScores_all(), want[k], err_msg=f"{release}, resident run {rep}, shape {shapes[k]}")
                launches += 1
        for b in batches:
            b.close()
        c.close()
    assert launches >= 300


def test_a_batch_may_be_destroyed_after_its_context():
    """A batch keeps its context's buffer pool and list of outgrown buffers alive (shared references, destroyed after the buffers that
    use them: rc_batch's member order), and does not touch the context itself when it goes -- the order the header discourages but
    the library has always claimed to survive (use-after-free in BufPool)."""
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    blocks = [b.upper() for b in synth_blocks(6, 5, 60, seed=11)]
    p = api.default_params(sampleN=64, seed_base=3)
    for running in (False, True):
        c = api.Context(0)
        done = api.Batch(c, blocks, p).run()
        want = done.maxScores_all().copy()
        other = api.Batch(c, blocks, p)
        if running:
            api._check(api.lib().rc_batch_run_async(other._h))   # still RUNNING when its context goes
        c.close()                # synchronises the device, drains the pool
        other.close()
        done.close()
        c2 = api.Context(0)      # and the next context starts clean
        again = api.Batch(c2, blocks, p).run()
        np.testing.assert_array_equal(again.maxScores_all(), want)
        again.close()
        c2.close()


def test_a_context_gives_its_kept_buffers_back_when_asked():
    """rc_ctx_trim (a context keeps up to a sixteenth of the device's memory of destroyed batches' buffers until it is
    destroyed): after a batch is destroyed its device buffers stay with the context, after trim the device has them back, and the next
    batch on the trimmed context gives the same numbers."""
    import ctypes
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    blocks = [b.upper() for b in synth_blocks(400, 6, 120, seed=12)]
    p = api.default_params(sampleN=256, seed_base=5)
    c = api.Context(0)
    b = api.Batch(c, blocks, p).run()
    want = b.maxScores_all().copy()
    used = free_bytes()
    b.close()
    kept = free_bytes()
    c.trim()
    freed = free_bytes()
    assert kept - used < (8 << 20), (used, kept)          # the batch's buffers went to the pool, not back to the device
    assert freed - kept > (32 << 20), (kept, freed)       # ... and trim gives them back
    again = api.Batch(c, blocks, p).run()
    np.testing.assert_array_equal(again.maxScores_all(), want)
    again.close()
    c.close()


def test_no_step_of_a_small_multi_class_batch_stalls():
    """BASELINE configs[1] as one batch, 200 times over in a process that has used every stream of the library before (a resident
    batch, a stream of sub-batches, tree fits): no warm step may take more than ten times the median.  Round 4's bench line had a
    1.6 s step among 5 ms ones there: with eight hardware queues per stream priority the process oversubscribed the device's queue
    slots, and the scheduler then time-slices the queues (tools/stall_hunt.py, profiles/r05/stall_hunt.txt)."""
    import time

    import bench
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    ctx = api.Context(0)
    api.lib().rc_set_stream_cache(0)
    big = [b.upper() for b in synth_blocks(2000, 6, 120, seed=1)]
    p = api.default_params(sampleN=1000, seed_base=42)
    m = api.Marshalled(big)
    m.set_trees()
    resident = api.Batch(ctx, m, p).run()
    st = api.Stream(ctx, p, 3)
    for sb in api.score_stream(ctx, m, p, [250, 500, 1250], stream=st):
        sb.fits()
        sb.close()
    api.fit_trees(big[:64], ctx=ctx)
    blocks, n, seed, pk = bench.golden_blocks("genomic_preprocessed_n1000")
    pf = api.default_params(sampleN=n, seed_base=seed, **pk)
    times = []
    for i in range(203):
        t0 = time.perf_counter()
        b = api.Batch(ctx, blocks, pf).run()
        for j in range(b.n):
            if b.status(j) == api.RC_OK:
                b.scoreAln(j)
        b.close()
        if i >= 3:
            times.append((time.perf_counter() - t0) * 1e3)
    med = sorted(times)[len(times) // 2]
    worst = max(times)
    st.close(); resident.close(); ctx.close()
    assert worst <= 10.0 * med, f"a step took {worst:.1f} ms, the median is {med:.2f} ms"


def test_rows_split_over_workgroups_equal_the_fused_scan_and_the_oracle(monkeypatch):
    """Tiny batches (a caller that scores block by block: RNAcode.c:164-216 through the shim) split every strand x frame part's rows
    over up to eight workgroups (k_null<.., ROWS>: the S values go to a buffer in getHSS's entry order) and fold them afterwards
    (k_null_rowscan).  One block, two blocks of different row counts, ragged sample counts, gapped blocks with frame-shift events, a
    block of three codons, --stop-early in rounds: per-sample maxima, fits and HSS tables equal RC_ROW_SPLIT=0's bit for bit, and the
    oracle's."""
    from oracle import binding as ob
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(404)
    cases = []
    for rows, cols, n in ((10, 201, 1000), (4, 76, 1000), (8, 342, 1000), (6, 120, 100), (3, 30, 64), (12, 150, 129), (7, 97, 333), (5, 9, 200), (31, 60, 64)):
        cases.append(([synth_block(rng, rows, cols, index=0, gaps=True).upper()], n, {}))
    cases.append(([synth_block(rng, 6, 150, index=0, gaps=True).upper(), synth_block(rng, 9, 90, index=1, gaps=True).upper()], 192, {}))
    cases.append(([synth_block(rng, 8, 120, index=0, gaps=False, mean_branch=0.02).upper()], 1000, dict(stopEarly=1, cutoff=0.05)))
    got = {}
    for split in ("1", "0"):
        monkeypatch.setenv("RC_ROW_SPLIT", split)
        monkeypatch.setenv("RC_STOP_MIN_ITEMS", "0")     # --stop-early in rounds even for one block
        c = api.Context(0)
        for k, (blocks, n, kw) in enumerate(cases):
            b = api.Batch(c, blocks, api.default_params(sampleN=n, seed_base=23, **kw)).run()
            got[(k, split)] = (b.maxScores_all().copy(), b.fits().copy(), [b.scoreAln(i) for i in range(b.n)], int(b.clamped()))
            b.close()
        c.close()
    for k, (blocks, n, kw) in enumerate(cases):
        on, off = got[(k, "1")], got[(k, "0")]
        np.testing.assert_array_equal(on[0], off[0], err_msg=f"case {k}")
        np.testing.assert_array_equal(on[1], off[1], err_msg=f"case {k} (fits)")
        assert on[2] == off[2] and on[3] == off[3]
        if not kw:
            blk = blocks[0]
            m = min(n, 64)
            want = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa,
                                ob.default_params(m), 23)
            np.testing.assert_array_equal(on[0][0][:m], np.float32(want.maxScores), err_msg=f"case {k} against the oracle")
