"""The randomised evidence that used to be manual (tools/soak_parity.py, tools/knob_suite.sh), bounded so that it runs with
`pytest -m gpu`: (1) a slice of the parity soak in each of its modes -- random shapes, sample counts, seeds, matrices, gap
parameters, ambiguity codes, ties, --stop-early, streams of random sub-batches, backtrack paths -- HIP path against the CPU oracle;
(2) a parity subset -- reference-scored goldens and a stream-equals-batch check -- under every non-default RC_* switch of the
library (each is read when a context is created, so every case gets a context of its own)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _soak_module():
    spec = importlib.util.spec_from_file_location("soak_parity", os.path.join(ROOT, "tools", "soak_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# mode, environment, blocks: 2000 blocks in all (the wide mode's blocks are 65..300 rows each: fewer of them)
SOAK = [
    ("", {}, 700),
    ("fat", {}, 700),
    ("wide", {}, 60),
    ("", {"RC_LDS_MAX_BYTES": "0"}, 300),
    ("", {"RC_GENERIC_MIN_ROWS": "3"}, 240),
    ("", {"RC_GENERIC_MIN_ROWS": "65", "RC_TILED_MIN_ROWS": "65"}, 240),   # the k_null<36..63> instantiations: by default blocks of 34 rows and more take the tiled or the generic kernels (rc_device.h)
    ("", {"RC_TILED_MIN_ROWS": "17", "RC_TILED_MAX_L": "100000"}, 240),   # k_tiled_dp for every block of 17..64 rows
    ("wide", {"RC_TILED_MAX_ROWS": "64"}, 40),                            # k_generic_dp for every block of more than 64 rows (by default most of them take the tiled kernels)
    ("", {"RC_SOAK_EXOTIC": "1"}, 400),   # degenerate inputs among the blocks (tools/soak_exotic.py: one or two letters only, kappa 0.01..900, zero or saturated branches, gap rows ...)
]


@pytest.mark.parametrize("mode,env,blocks", SOAK, ids=["default", "fat", "wide", "codes-from-L2", "all-generic", "templated-to-64-rows", "tiled-from-17-rows", "generic-past-64-rows",
                              "degenerate-blocks"])
def test_soak_slice(monkeypatch, mode, env, blocks):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lines = []
    rounds, done, bad, paths = _soak_module().soak(budget=40.0, rng_seed=20260 + len(mode) + 7 * len(env) + blocks, mode=mode, max_blocks=blocks,
                                                   log=lines.append)
    assert bad == 0, "\n".join(lines)
    assert done >= min(blocks, 20), lines


KNOBS = [
    {"RC_TAIL_SHARING": "0"}, {"RC_FAT_STREAM_MIN_ITEMS": "0"},
    {"RC_HOST_THREADS": "1", "RC_SPLIT_FACTOR": "0"}, {"RC_SPLIT_FACTOR": "100"}, {"RC_NATIVE_WAVES_PER_CU": "1"},
    {"RC_NATIVE_WAVES_PER_CU": "32"}, {"RC_TOGETHER_MAX_MB": "0"}, {"RC_STOP_ROUNDS": "6", "RC_STOP_MIN_ITEMS": "0"},
    {"RC_GENERIC_MIN_ROWS": "3"}, {"RC_GENERIC_MIN_ROWS": "65", "RC_TILED_MIN_ROWS": "65"}, {"RC_TILED_MIN_ROWS": "17", "RC_TILED_MAX_L": "100000"}, {"RC_FIT_EXP": "generic"}, {"RC_FIT_EXP": "fused"}, {"RC_LDS_MAX_BYTES": "0"}, {"RC_LDS_MAX_BYTES": "65536"}, {"RC_GRID_CAP_PER_CU": "2"},
    {"RC_GENERIC_SCRATCH_MB": "64"}, {"RC_SERIAL_NATIVE": "1"}, {"RC_HIGH_OCCUPANCY": "0"}, {"RC_HIGH_OCCUPANCY": "2"},
    {"RC_TREE_LDS_MAX": "0"}, {"GPU_MAX_HW_QUEUES": "2"},
]


def _hss_key(h):
    return (-h["score"], h["strand"], h["frame"], h["start"], h["end"])


@pytest.mark.parametrize("env", KNOBS, ids=[",".join(f"{k}={v}" for k, v in e.items()) for e in KNOBS])
def test_parity_subset_under_every_switch(monkeypatch, env):
    from helpers import block_from_golden, param_kwargs
    from rnacode_amd import api
    from rnacode_amd.synth import synth_block
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = api.Context(0)
    # reference-scored goldens: maxima, HSS tables and fit verdicts
    for name in ("coding_aln_n100", "synth_6x120_n200", "edge_cases_n50", "synth_12x300_n40", "synth_5x60_stopearly_n100", "synth_wide_70x120_n20", "synth_wide_40x90_n30"):
        doc = load_golden(name)
        blocks = [block_from_golden(e) for e in doc["blocks"]]
        batch = api.Batch(ctx, blocks, api.default_params(seed_base=doc["seed_base"], **param_kwargs(doc))).run()
        assert batch.clamped() == 0
        for i, e in enumerate(doc["blocks"]):
            ref = e["ref"]
            if "skipped" in ref:
                continue
            got, want = sorted(batch.scoreAln(i), key=_hss_key), sorted(ref["hss"], key=_hss_key)
            assert [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == \
                   [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want], (name, i)
            rc, mu, lam = batch.getExtremeValuePars(i)
            assert rc == ref["evd_rc"], (name, i)
            if ref["maxScores"] and not doc.get("params", {}).get("stopEarly"):
                np.testing.assert_array_equal(batch.maxScores(i), np.float32(ref["maxScores"]), err_msg=f"{name} block {i}")
        batch.close()
    # a mixed batch (six row counts, gapped, a little over one round of the chip) as one batch and as a stream of three sub-batches
    rng = np.random.RandomState(4242)
    mixed = [synth_block(rng, int(rng.choice([3, 4, 6, 7, 9, 12])), int(rng.randint(40, 181)), index=i, gaps=True).upper() for i in range(360)]
    p = api.default_params(sampleN=1000, seed_base=31)
    m = api.Marshalled(mixed)
    m.set_trees()
    b = api.Batch(ctx, m, p).run()
    whole = b.maxScores_all().copy()
    fits = b.fits().copy()
    b.close()
    parts = []
    for sb in api.score_stream(ctx, m, p, [100, 140, 120]):
        parts.append((sb.maxScores_all().copy(), sb.fits().copy()))
        sb.close()
    np.testing.assert_array_equal(whole, np.concatenate([x[0] for x in parts]))
    np.testing.assert_array_equal(fits, np.concatenate([x[1] for x in parts]))
    ctx.close()
    # ... and those maxima are the default build's (computed once per session without any switch)
    key = "default"
    if key not in _BASE:
        for k in env:
            monkeypatch.delenv(k)
        c0 = api.Context(0)
        b0 = api.Batch(c0, m, p).run()
        _BASE[key] = (b0.maxScores_all().copy(), b0.fits().copy())
        b0.close()
        c0.close()
    np.testing.assert_array_equal(whole, _BASE[key][0])
    np.testing.assert_array_equal(fits, _BASE[key][1])


_BASE = {}
