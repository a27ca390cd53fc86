"""CPU-side tests: alignment readers, synthetic generator, and that the C-ABI library loads and
exports every symbol include/rnacode_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_golden
from rnacode_amd import alnio, synth


def test_library_exports_every_declared_symbol():
    from rnacode_amd import api
    hdr = open(os.path.join(ROOT, "include", "rnacode_hip.h")).read()
    declared = set(re.findall(r"\b(rc_[a-z_0-9A-Z]+)\s*\(", hdr))
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)
    if not os.path.exists(api.LIB_PATH):
        api.build_library()
    lib = ctypes.CDLL(api.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product must fail, never fall back to a CPU path."""
    from rnacode_amd import api
    if api.lib().rc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.RnacodeError) as ei:
        api.Context(0)
    assert ei.value.code == api.RC_ERR_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_default_params_match_reference_defaults():
    from rnacode_amd import api
    p = api.default_params()
    assert (p.Delta, p.Omega, p.omega, p.stopPenalty_0, p.stopPenalty_k) == (-10.0, -4.0, -2.0, -9999.0, -8.0)
    assert (p.blosum, p.sampleN, p.cutoff, p.stopEarly) == (62, 100, 1.0, 0)


def test_maf_reader_skips_and_fields():
    text = """##maf version=1
# comment
a score=1
s hg.chr1 10 9 + 100 ATGGCTAAA
i hg.chr1 N 0 C 0
s mm.chr2 20 9 - 200 ATGGCAAAA
q mm.chr2 999999999
e rn.chr3 0 10 + 20 I

a score=2
s hg.chr1 30 3 + 100 A-TG
s mm.chr2 40 4 + 200 ACTG
"""
    blocks = list(alnio.read_maf(text.splitlines(True)))
    assert [b.n for b in blocks] == [2, 2]
    assert blocks[0].rows[1].strand == "-" and blocks[0].rows[1].start == 20 and blocks[0].rows[1].full_length == 200
    assert blocks[1].ref_len == 3 and blocks[1].cols == 4
    with pytest.raises(ValueError):
        list(alnio.read_maf(["a\n", "s x 1 2 + 3\n"]))
    with pytest.raises(ValueError):
        list(alnio.read_maf(["a\n", "s x 1 2 + 3 ACG\n", "s y 1 2 + 3 AC\n"]))


def test_clustal_reader_interleaved():
    text = ("CLUSTAL W(1.81) multiple sequence alignment\n\n\nseqA    ATG-C\nseqB    ATGGC\n        *** *\n\n"
            "seqA    TTT\nseqB    TTA\n")
    (b,) = list(alnio.read_clustal(text.splitlines(True)))
    assert [r.seq for r in b.rows] == ["ATG-CTTT", "ATGGCTTA"]
    assert b.rows[0].start == 0 and b.rows[0].length == 0 and b.rows[0].strand == "?"


def test_golden_inputs_round_trip_through_readers(tmp_path):
    doc = load_golden("genomic_preprocessed_n100")
    blocks = [alnio.AlnBlock([alnio.AlnRow(r["name"], r["seq"], r["start"], r["length"], r["strand"], r["full_length"])
                              for r in e["input"]["rows"]], e["input"]["id"]) for e in doc["blocks"]]
    p = tmp_path / "x.maf"
    p.write_text(synth.to_maf(blocks))
    again = alnio.read_alignment_file(str(p))
    assert len(again) == len(blocks) == 34
    for a, b in zip(again, blocks):
        assert [(r.name, r.seq, r.start, r.length, r.strand) for r in a.rows] == \
               [(r.name, r.seq, r.start, r.length, r.strand) for r in b.rows]


def test_synthetic_generator_is_deterministic_and_well_formed():
    a = synth.synth_blocks(3, 6, 120, seed=1)
    b = synth.synth_blocks(3, 6, 120, seed=1)
    assert [[r.seq for r in x.rows] for x in a] == [[r.seq for r in x.rows] for x in b]
    assert [x.tree for x in a] == [x.tree for x in b]
    for blk in a:
        assert blk.n == 6 and blk.cols == 120 and blk.tree.endswith(");") and 1.5 <= blk.kappa <= 4.0
        assert all(set(r.seq) <= set("ACGT-") for r in blk.rows)
        assert any("-" in r.seq for r in blk.rows)
        for r in blk.rows:
            assert r.name in blk.tree and r.length == sum(c != "-" for c in r.seq)
    # the committed golden inputs came from this generator
    doc = load_golden("synth_6x120_n200")
    regen = synth.synth_blocks(6, 6, 120, seed=1)
    for e, g in zip(doc["blocks"], regen):
        assert [r["seq"] for r in e["input"]["rows"]] == [r.seq for r in g.rows]
        assert e["input"]["tree"] == g.tree


def _build_c_example(out_path, source="example_batch.c"):
    import subprocess
    lib_dir = os.path.join(ROOT, "rnacode_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", source), "-L" + lib_dir, "-lrnacode_hip",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", out_path]
    subprocess.check_call(cmd)


def test_header_is_plain_c_and_the_example_links(tmp_path):
    """include/rnacode_hip.h compiles as strict C99 and integration/example_batch.c / example_stream.c link against the library."""
    from rnacode_amd import api
    api.lib()
    _build_c_example(str(tmp_path / "example_batch"))
    _build_c_example(str(tmp_path / "example_stream"), "example_stream.c")


def test_native_driver_is_built_and_fails_loudly_without_a_gpu(tmp_path):
    """rnacode_amd/rnacode_hip (rc_cli.cpp) is built next to the library; without a HIP device it says so instead of scoring
    anything on the CPU (this test runs in the GPU-less container; on a GPU box the device exists and it scores)."""
    import subprocess
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: rnacode_hip" in r.stderr
    maf = tmp_path / "in.maf"
    maf.write_text("a score=0\ns a.chr1 0 9 + 100 ATGGCTAAA\ns b.chr1 0 9 + 100 ATGGCAAAA\ns c.chr1 0 9 + 100 ATGGCTAAG\n\n")
    from rnacode_amd import api
    r = subprocess.run([exe, str(maf), "-n", "64", "-t"], capture_output=True, text=True)
    if api.lib().rc_device_count() == 0:
        assert r.returncode != 0 and "no HIP device" in r.stderr and r.stdout == ""
    else:
        assert r.returncode == 0
    bad = tmp_path / "bad.maf"
    bad.write_text("a score=0\ns a.chr1 0 9 + 100\n")
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid MAF format" in r.stderr
