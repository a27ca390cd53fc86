"""rc_refexp.h restates exp() as the reference's host C library computes it -- glibc's algorithm, table recomputed, a generic variant and one
with fused multiply-adds -- so that the EVD fit's Newton iteration takes the reference's steps.  Here, without a
GPU: one of the two variants must equal this machine's exp() on every one of four million arguments, subnormal results, overflow,
underflow and the special values included, and tools/make_exp_table.py must reproduce the table in the header."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "rc_refexp.h"
#include <cmath>
#include <cstdio>
int main() {
  long bad[2] = {0, 0}, n = 0;
  uint64_t s = 88172645463325252ull;
  for (int range = 0; range < 4; range++) {
    const double lo = range == 0 ? -745.2 : range == 1 ? -40 : range == 2 ? -1e-17 : 0, hi = range == 3 ? 709.9 : range == 2 ? 1e-17 : 0;
    for (long i = 0; i < 1000000; i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double x = lo + (s >> 11) * (1.0 / 9007199254740992.0) * (hi - lo), want = std::exp(x);
      n++;
      for (int v = 0; v < 2; v++) bad[v] += rc::exp_bits(rc::ref_exp(x, v == 1)) != rc::exp_bits(want);
    }
  }
  const double sp[] = {-746.0, -750.0, -1100.0, 710.0, 1100.0, -0.0, 0.0, -INFINITY, INFINITY, 1.0, -1.0};
  for (double x : sp) for (int v = 0; v < 2; v++) bad[v] += rc::exp_bits(rc::ref_exp(x, v == 1)) != rc::exp_bits(std::exp(x));
  std::printf("%ld arguments: generic differs on %ld, fused on %ld\n", n, bad[0], bad[1]);
  return 0;
}
'''


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
def test_one_variant_is_this_machines_exp(tmp_path):
    (tmp_path / "t.cpp").write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "rnacode_amd", "csrc"), str(tmp_path / "t.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    generic, fused = (int(x) for x in re.search(r"generic differs on (\d+), fused on (\d+)", out).groups())
    assert min(generic, fused) == 0, out
    # with the CPU-specific variants of the C library switched off its exp must be the generic one
    env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-FMA,-FMA4,-AVX2_Usable,-AVX2")
    out = subprocess.check_output([str(exe)], text=True, env=env)
    generic2, _ = (int(x) for x in re.search(r"generic differs on (\d+), fused on (\d+)", out).groups())
    if fused == 0 and generic != 0:      # (a C library that honours the tunable: glibc on x86-64)
        assert generic2 == 0, out


def test_the_table_is_what_its_definition_gives():
    made = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "make_exp_table.py")], text=True)
    want = re.findall(r"0x[0-9a-f]{16}", made)
    text = open(os.path.join(ROOT, "rnacode_amd", "csrc", "rc_refexp.h")).read()
    have = re.findall(r"0x[0-9a-f]{16}(?=ull)", text[text.index("#define RC_EXP_TAB"):text.index("#undef RC_EXP_TAB")])
    assert len(want) == 256 and have == want
