"""Code-generation guard for the kernel that takes 96 % of the device time (needs hipcc, no GPU).

k_null's cell loops only run at their measured rate while the compiler can prove their control flow wave-uniform: one value that
reaches a branch through a vector register (round 3: a __shfl result deciding which item a workgroup helps with) turned every loop
of the kernel into an exec-masked loop -- five more instructions per cell and 49 -> 65 ms per launch, with all parity tests green.
This test compiles the N-1 = 2..6 unit the way the Makefile does and looks at the ISA of k_null<5, true, false> (the bench
workload's instantiation)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = "_ZN2rc6k_nullILi5ELb1ELb0EEEvNS_8NullArgsEPKhPKNS_8DevBlockEPKiPKjSA_S3_PjPf"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_k_null_cell_loops_are_wave_uniform_and_tight(tmp_path):
    out = tmp_path / "rc_null_a.s"
    flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", *flags, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "rnacode_amd", "csrc", "rc_null_a.hip"), "-o", str(out)])
    txt = out.read_text()
    body = txt[txt.index(NAME + ":"):]
    body = body[:body.index(".Lfunc_end")]
    blocks, cur = [], None
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            cur = []
            blocks.append(cur)
        elif cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
            cur.append(ln.strip())
    cells = [b for b in blocks if sum("ds_bpermute_b32" in x for x in b) == 5]   # one sigma look-up per sequence: a cell loop
    # the four cell kinds: before the row's first frame-shift event, at an event, after the last one, between events
    assert len(cells) == 4, [len(b) for b in cells]
    for b in cells:
        assert not any("exec" in x for x in b), "a cell loop manipulates exec: its trip count is no longer wave-uniform\n" + "\n".join(b)
        assert not any(x.startswith(("scratch_", "buffer_")) for x in b)
    # instructions per cell (round 2/3 builds: 50, 50, 53, 58)
    assert sorted(len(b) for b in cells) <= [50, 50, 53, 58], sorted(len(b) for b in cells)
    meta = txt[txt.index(".name:           " + NAME):]
    vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    assert vgprs <= 128, vgprs   # four wavefronts per SIMD (16 per CU is what the LDS allows)
    head = txt[txt.rindex(".amdhsa_kernel " + NAME):]
    assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", head).group(1)) == 0
