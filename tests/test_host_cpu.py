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


def _native_dump(path):
    import subprocess
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    r = subprocess.run([exe, "--dump-blocks", str(path)], capture_output=True, text=True)
    if r.returncode != 0:
        return r.stderr.strip()
    blocks, cur = [], None
    for line in r.stdout.splitlines():
        f = line.split(" ")
        if f[0] == "B":
            cur = []
        elif f[0] == "S":
            cur.append((f[1], int(f[2]), int(f[3]), f[4], f[5]))
        elif f[0] == "E":
            blocks.append(cur)
    return blocks


def _python_dump(path):
    from rnacode_amd import alnio
    try:
        return [[(r.name, r.start, r.length, r.strand, r.seq) for r in b.rows] for b in alnio.read_alignment_file(str(path))]
    except ValueError as e:
        return "ERROR: " + str(e)


def test_native_reader_parses_what_the_python_reader_parses(tmp_path):
    """The two drivers share no reader code (rc_cli.cpp read_maf/read_clustal, alnio.py): random MAF and ClustalW texts with the
    things rnaz_utils.c:44-234 has to cope with -- i/e/q and comment lines, blank lines, tabs, CRLF, blocks without 's' lines,
    wrapped ClustalW blocks with conservation lines -- must come out of both as the same blocks, and the malformed ones
    (wrong field count, unequal lengths, inconsistent names) must be refused by both with the same message."""
    import numpy as np
    rng = np.random.RandomState(5)
    alpha = np.array(list("ACGTacgtNn-"))
    cases = 0
    for it in range(60):
        if it % 3 != 2:   # MAF
            out = ["##maf version=1\n"] if rng.rand() < 0.5 else []
            for b in range(rng.randint(0, 6)):
                out.append("a score=%d\n" % rng.randint(0, 1000) if rng.rand() < 0.8 else "a\n")
                cols = int(rng.randint(1, 80))
                bad_len = it % 10 == 7 and b == 1
                for r in range(rng.randint(0, 7)):
                    seq = "".join(rng.choice(alpha, cols - (1 if bad_len and r == 1 else 0)))
                    sep = "\t" if rng.rand() < 0.3 else " " * int(rng.randint(1, 4))
                    f = ["s", "sp%d.chr%d" % (r, rng.randint(1, 23)), str(rng.randint(0, 10 ** 6)), str(cols), "+-"[rng.randint(2)], str(10 ** 7), seq]
                    if it % 10 == 9 and b == 0 and r == 0:
                        f = f[:6]          # field count
                    out.append(sep.join(f) + ("\r\n" if rng.rand() < 0.2 else "\n"))
                    if rng.rand() < 0.3:
                        out.append("i sp%d.chr1 C 0 C 0\n" % r)
                    if rng.rand() < 0.1:
                        out.append("q sp%d.chr1 %s\n" % (r, "9" * cols))
                if rng.rand() < 0.3:
                    out.append("e sp9.chr2 100 20 + 5000 I\n")
                if rng.rand() < 0.2:
                    out.append("# a comment\n")
                out.append("\n" if rng.rand() < 0.9 else "")
            text = "".join(out)
        else:             # ClustalW, wrapped at 60 columns
            n, cols = int(rng.randint(1, 8)), int(rng.randint(1, 200))
            names = ["seq_%d" % i for i in range(n)]
            seqs = ["".join(rng.choice(alpha, cols)) for _ in range(n)]
            out = ["CLUSTAL W (1.83) multiple sequence alignment\n", "\n", "\n"]
            for at in range(0, cols, 60):
                for i in range(n):
                    name = names[i] if not (it % 12 == 5 and at > 0 and i == 0) else "other"
                    out.append("%-16s%s%s\n" % (name, seqs[i][at:at + 60], " %d" % min(cols, at + 60) if rng.rand() < 0.3 else ""))
                out.append(" " * 16 + "".join(rng.choice(list(" *"), min(60, cols - at))) + "\n")
                out.append("\r\n" if rng.rand() < 0.2 else "\n")
            text = "".join(out)
        p = tmp_path / ("case%d.txt" % it)
        p.write_bytes(text.encode())
        got, want = _native_dump(p), _python_dump(p)
        assert got == want, (it, text[:400])
        cases += 1
    assert cases == 60


def test_native_driver_limit_prunes_rows_like_prune_aln(tmp_path):
    """--limit (pruneAln, rnaz_utils.c:724-752): a row stays if its name STARTS with one of the listed strings; columns are untouched."""
    maf = tmp_path / "in.maf"
    maf.write_text("a score=0\ns hg18.chr1 0 9 + 100 ATGGCTAAA\ns mm9.chr1 0 9 + 100 ATGGCAAAA\ns hg18b.chr1 0 9 + 100 ATGGCTAAG\ns canFam2.chr3 0 9 + 100 ATG---AAG\n\n"
                   "a score=1\ns mm9.chr2 5 3 + 100 ATG\ns rn4.chr2 5 3 + 100 ATG\n\n"
                   "a score=2\ns rn4.chr2 5 3 + 100 ATG\n\n")
    import subprocess
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    r = subprocess.run([exe, "--dump-blocks", "--limit", "hg18,canFam,mm", str(maf)], capture_output=True, text=True)
    assert r.returncode == 0
    assert r.stdout == ("B 0\nS hg18.chr1 0 9 + ATGGCTAAA\nS mm9.chr1 0 9 + ATGGCAAAA\nS hg18b.chr1 0 9 + ATGGCTAAG\nS canFam2.chr3 0 9 + ATG---AAG\nE\n"
                        "B 1\nS mm9.chr2 5 3 + ATG\nE\n")
    assert r.stderr.count("Skipping alignment.") == 1   # the block that lost every row


def test_which_kernel_a_block_takes_and_the_tile_sizes(tmp_path):
    """rc_device.h decides a block's class in one place (block_class): per-row-count kernels to 31 rows, the tiled ones (32..64 rows of short
    blocks, 65..111 rows of blocks of up to 200 residues, every block from 112 rows on; omega <= 0) and the generic ones (from 37 rows on); tiled_kt picks the tile size with the least work.  Compiled on the host."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not on PATH")
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "rc_device.h"
#include <cstdio>
using namespace rc;
int main() {
  const ClassRule D = kDefaultClassRule;
  ClassRule W = D; W.tiledMaxRows = 300; W.tiledAnyLMinRows = 1000;
  ClassRule allGeneric = D; allGeneric.genericMinRows = 3;
  ClassRule off = D; off.genericMinRows = 65; off.tiledMinRows = 65;
  ClassRule offW = off; offW.tiledMaxRows = 300;
  int bad = 0;
  for (int nk = 16; nk <= 499; nk++) {
    const int kt = tiled_kt(nk), tiles = (nk + kt - 1) / kt;
    bad += kt < kTiledMinKT || kt > kTiledMaxKT || tiles * kt - nk >= kt;          // never a whole tile of padding
    for (int other = kTiledMinKT; other <= kTiledMaxKT && nk <= 63; other++) {      // ... and to 64 rows no other size does it in fewer passes
      const int t2 = (nk + other - 1) / other;
      bad += t2 < tiles;
    }
  }
  bad += block_class(6, 118, -2.0f, D) != 5;                                  // the headline's class: N - 1
  bad += block_class(31, 300, -2.0f, D) != 30;
  bad += block_class(32, 150, -2.0f, D) != kTiledClass0 + tiled_kt(31) - kTiledMinKT;
  bad += block_class(32, 300, -2.0f, D) != 31;                                // long blocks below 37 rows: the per-row-count kernels
  bad += block_class(40, 150, -2.0f, D) != kTiledClass0 + tiled_kt(39) - kTiledMinKT;
  bad += block_class(40, 300, -2.0f, D) != kGenericClass;
  bad += block_class(40, 150, 0.5f, D) != kGenericClass;                      // omega > 0: never the tiled kernels
  bad += block_class(64, 90, -2.0f, D) != kTiledClass0 + tiled_kt(63) - kTiledMinKT;
  bad += block_class(65, 90, -2.0f, D) != kTiledClass0 + tiled_kt(64) - kTiledMinKT;      // ... and on to 128 rows for blocks of up to 200 residues
  bad += block_class(111, 200, -2.0f, D) != kTiledClass0 + tiled_kt(110) - kTiledMinKT;
  bad += block_class(100, 201, -2.0f, D) != kGenericClass;
  bad += block_class(112, 3000, -2.0f, D) != kTiledClass0 + tiled_kt(111) - kTiledMinKT;   // from 112 rows on at every length
  bad += block_class(500, 300, -2.0f, D) != kTiledClass0 + tiled_kt(499) - kTiledMinKT;
  bad += block_class(40, 150, -2.0f, allGeneric) != kGenericClass;                     // RC_GENERIC_MIN_ROWS=3: everything generic
  bad += block_class(40, 150, -2.0f, off) != 39;                              // both switched off: k_null<39>
  bad += block_class(200, 150, -2.0f, W) != kTiledClass0 + tiled_kt(199) - kTiledMinKT;  // RC_TILED_MAX_ROWS=300, RC_TILED_ANY_L_MIN_ROWS=1000
  bad += block_class(100, 300, -2.0f, W) != kGenericClass;                               // ... the short ones
  bad += block_class(301, 150, -2.0f, W) != kGenericClass;
  bad += block_class(100, 150, -2.0f, offW) != kGenericClass;                            // RC_TILED_MIN_ROWS=65 switches all of it off
  std::printf("%d\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "rnacode_amd", "csrc"), str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).strip() == "0"
