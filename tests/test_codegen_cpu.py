"""Code-generation guard for the kernel that takes 96 % of the device time (needs hipcc, no GPU).

k_null's cell loops only run at their measured rate while the compiler can prove their control flow wave-uniform: one value that
reaches a branch through a vector register (round 3: a __shfl result deciding which item a workgroup helps with) turned every loop
of the kernel into an exec-masked loop -- five more instructions per cell and 49 -> 65 ms per launch, with all parity tests green.
This test compiles the N-1 = 2..6 unit the way the Makefile does and looks at the ISA of k_null<5, true, false> (the bench
workload's instantiation): the one-row cell loops, and the loops that walk two rows on one set of look-ups."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE_ROW = "_ZN2rc6k_nullILi5ELb1ELb0ELb0ELi0EEEvNS_8NullArgsEPKhPKNS_8DevBlockEPKiPKjSA_S3_PjPf"    # k_null<5, true, false, false>
TWO_ROWS = "_ZN2rc6k_nullILi5ELb1ELb0ELb1ELi0EEEvNS_8NullArgsEPKhPKNS_8DevBlockEPKiPKjSA_S3_PjPf"   # k_null<5, true, false, true>


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_k_null_cell_loops_are_wave_uniform_and_tight(tmp_path):
    txt = _compile_unit(tmp_path, "rc_null_a")

    def loops_of(name):
        body = txt[txt.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        blocks, cur = [], None
        for ln in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", ln):
                cur = []
                blocks.append(cur)
            elif cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
                cur.append(ln.strip())
        lookups = [b for b in blocks if sum("ds_bpermute_b32" in x for x in b) == 5]   # one sigma look-up per sequence: a cell (or cell pair) body
        for b in lookups:
            assert not any(re.match(r"s_\w+ exec\b", x) or "saveexec" in x for x in b), \
                "a cell loop writes exec: its trip count is no longer wave-uniform\n" + "\n".join(b)
            assert not any(x.startswith(("scratch_", "buffer_")) for x in b)
        loops = [b for b in lookups if any(x.startswith("s_cbranch") for x in b[-3:])]
        meta = txt[txt.index(".name:           " + name):]
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        head = txt[txt.rindex(".amdhsa_kernel " + name):]
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", head).group(1)) == 0
        return loops, vgprs

    # one row per pass: the three straight-line cell kinds -- before the row's first frame-shift event, after its last one, between
    # events (round 2/3 builds: 50, 53, 58 instructions per cell; the event cell branches per sequence and spans several blocks)
    loops, vgprs = loops_of(ONE_ROW)
    singles = [b for b in loops if len(b) <= 60 and sum(x.startswith("v_") for x in b) >= 25]
    assert len(singles) == 3 and sorted(len(b) for b in singles) <= [50, 53, 58], sorted(len(b) for b in singles)
    assert vgprs <= 80, vgprs        # leaves 192 VGPRs per SIMD to the kernels that run beside it
    # two rows per pass: the pair loops keep row a + 1's S value with ONE indexed register move (s_set_gpr_idx)
    # and never copy the 32-register row buffer: 75, 83, 91 instructions per TWO cells
    loops, vgprs = loops_of(TWO_ROWS)
    pairs = [b for b in loops if any("s_set_gpr_idx_on" in x for x in b)]
    assert len(pairs) == 3 and sorted(len(b) for b in pairs) <= [75, 83, 91], sorted(len(b) for b in pairs)
    for b in pairs:
        assert sum("s_set_gpr_idx_on" in x for x in b) == 1 and not any(x.startswith("v_mov_b64") for x in b)
    assert vgprs <= 128, vgprs       # four wavefronts per SIMD (16 per CU is what the LDS allows)


PLAIN = "_ZN2rc6k_nullILi6ELb0ELb0ELb0ELi0EEEvNS_8NullArgsEPKhPKNS_8DevBlockEPKiPKjSA_S3_PjPf"      # k_null<6, false, false, false>
OCC6 = "_ZN2rc10k_null_occILi6EEEvNS_8NullArgsEPKhPKNS_8DevBlockEPKiPKjSA_S3_PjPf"             # k_null_occ<6>
NCW_WAIT_WINDOW = 12   # instructions behind a row's first fetch within which its full wait stands
_FETCH = re.compile(r"(?:global_load_(?:dword|ubyte|ushort) (v\d+), v\d+, s\[\d+:\d+\]|ds_read_(?:b32|u8|u16) (v\d+), v\d+)")


_UNITS = ("rc_null_a", "rc_null_b", "rc_null_c", "rc_null_d", "rc_null_t", "rc_kernels")   # what the tests of this module look at
_COMPILES = {}   # unit -> (process, path of its assembly): all started by the first test that needs one (minutes of hipcc side by side, not in a row)


def _compile_unit(tmp_path, unit):
    """The gfx950 assembly of one instantiation unit, compiled the way the Makefile does."""
    if not _COMPILES:
        import tempfile
        d = tempfile.mkdtemp(prefix="rc_codegen_")
        flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
        for u in _UNITS:
            out = os.path.join(d, u + ".s")
            _COMPILES[u] = (subprocess.Popen(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", *flags, "-I", os.path.join(ROOT, "include"),
                                              os.path.join(ROOT, "rnacode_amd", "csrc", u + ".hip"), "-o", out]), out)
    proc, out = _COMPILES[unit]
    assert proc.wait() == 0, unit + " does not compile"
    with open(out) as f:
        return f.read()


def _audit_pending_registers(txt, name):
    """In every innermost loop of kernel `name` that holds asm fetches (the cell loops: look-ups, then the next site's words loaded INTO
    the registers just read, waited for by the next cell), no instruction of the loop -- and none between the loop's exits and the
    row's closing wait -- may copy (v_mov) or spill (scratch_store) a register such a fetch writes: the compiler does not know the
    data is still on its way.  Returns the number of loops audited and the fetch instructions seen."""
    body = txt[txt.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    blocks, order = {}, []
    cur, hdr = None, None
    inner = set()   # labels the assembly printer marks "This Inner Loop Header": the innermost loops
    for ln in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(?:;.*Header=(BB\d+_\d+) Depth=(\d+))?", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = {"ins": [], "header": ("." + "L" + m.group(2)) if m.group(2) else None, "depth": int(m.group(3) or 0), "asm": []}
            order.append(cur)
            continue
        if cur is not None and "This Inner Loop Header" in ln and not blocks[cur]["ins"]:
            inner.add(cur)
            continue
        m = re.match(r"^; %bb\.\d+:\s*;.*Header=(BB\d+_\d+) Depth=(\d+)", ln)
        if m and cur is not None:   # an unlabelled block of the same loop: keep appending
            continue
        if cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
            blocks[cur]["ins"].append(ln.strip())
    loops = {}
    for lab in order:
        b = blocks[lab]
        h = b["header"] or lab
        loops.setdefault(h, []).append(lab)
    audited = fetches = 0
    # Round 6: a span's pair loop (two cells of fetch distance) is entered behind ONE fetch into the second register set, issued in the block in
    # front of the loop: from that fetch to the loop's header nothing may copy or spill its destination either.
    for k, lab in enumerate(order):
        ins = blocks[lab]["ins"]
        if lab in inner:
            continue
        for at, x in enumerate(ins):
            m = re.match(r"global_load_(?:dword|ubyte|ushort) (v\d+), v\d+, s\[\d+:\d+\]", x)
            if not m:
                continue
            reg = m.group(1)
            tail = ins[at + 1:]
            if any(y.startswith("s_waitcnt") and "vmcnt(0)" in y for y in tail[:NCW_WAIT_WINDOW]):
                continue   # a row's first site: fetched and waited for on the spot (load_words)
            nxt = order[k + 1] if k + 1 < len(order) else None
            if nxt in inner and not any(y.startswith(("s_cbranch", "s_branch")) for y in tail):   # falls through into the pair loop
                bad = [y for y in tail if (re.match(r"v_mov_b32(_e32)? v\d+, %s$" % reg, y) or (y.startswith("scratch_store") and re.search(r"\b%s\b" % reg, y)))]
                assert not bad, name + ": the register of a pair loop's entry fetch is copied or spilled in front of the loop:\n" + "\n".join(bad)
    for h, labs in loops.items():
        if h not in inner:
            continue   # (outer loops only hold such entry fetches and the rows' first sites; their blocks reuse the cells' registers for other values)
        ins = [x for lab in labs for x in blocks[lab]["ins"]]
        # the sigma look-ups of a cell: ds_bpermute_b32, or (codes from L2) ds_read_b32 from the LDS tables at an address masked with 0xfc
        table_addr = {m.group(1) for x in ins for m in [re.match(r"v_and_b32 (v\d+), 0xfc, v\d+", x)] if m}
        if not any(x.startswith("ds_bpermute_b32") for x in ins) and not table_addr:
            continue
        regs = set()
        for at, x in enumerate(ins):
            m = _FETCH.match(x)
            if m:
                la = re.match(r"ds_read_b32 v\d+, (v\d+)", x)
                if la and la.group(1) in table_addr:
                    # the register's last writer decides: masked with 0xfc -> a look-up (its result is waited for inside the cell), not a fetch
                    writer = next((y for y in reversed(ins[:at]) if re.match(r"\S+ %s\b" % la.group(1), y)), "")
                    if writer.startswith("v_and_b32") and "0xfc" in writer:
                        continue
                regs.add(m.group(1) or m.group(2))
                fetches += 1
        if not regs:
            continue
        audited += 1

        def bad(x):
            if re.match(r"v_mov_b32(_e32)? v\d+, (v\d+)$", x) and x.split(", ")[1] in regs:
                return True
            return x.startswith("scratch_store") and any(re.search(r"\b%s\b" % r, x) for r in regs)
        viol = [x for x in ins if bad(x)]
        # behind the loop's exits, up to the wait that closes the row
        for lab in labs:
            for x in blocks[lab]["ins"]:
                m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", x)
                tgt = m and (m.group(1) or m.group(2))
                if tgt and tgt not in labs and tgt in blocks:
                    k = order.index(tgt)
                    seen = 0
                    while k < len(order) and seen < 120:
                        stop = False
                        for y in blocks[order[k]]["ins"]:
                            seen += 1
                            if y.startswith("s_waitcnt") and "vmcnt(0)" in y:
                                stop = True
                                break
                            if bad(y):
                                viol.append(order[k] + ": " + y)
                        if stop or order[k] in labs:
                            break
                        k += 1
        assert not viol, name + ": a register an asm fetch leaves pending is copied or spilled:\n" + "\n".join(viol[:10])
    return audited, fetches


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_codes_from_l2_cell_keeps_its_fetched_words_in_place(tmp_path):
    """k_null<N-1, false, false> fetches the next cell's code words with asm loads INTO the registers the look-ups have just read, and
    the next cell waits for them (rc_null_kernel.h, lookup).  The compiler does not know those registers are pending: a copy out of
    one of them before that wait reads garbage.  A build that cut the row loops by kind of fetch had exactly that (a v_mov at a join,
    in front of the wait) and got every row with a frame-shift event wrong; the parity suite caught it, this test names the cause."""
    txt = _compile_unit(tmp_path, "rc_null_a")
    for name in (PLAIN, OCC6):     # (k_null_occ<6>: the same code compiled for 80 registers, a few of them spilled)
        audited, fetches = _audit_pending_registers(txt, name)
        assert audited >= 4 and fetches >= 16, (name, audited, fetches)     # four kinds of cell, two code words each, both arms
    body = txt[txt.index(PLAIN + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert "global_load_ubyte" in body and "ds_read_u8" in body      # the narrow tail word (six codes: a dword and a byte)
    meta = txt[txt.index(".name:           " + PLAIN):]
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 96      # five wavefronts per SIMD


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_pending_fetch_register_is_copied_or_spilled_at_twelve_rows(tmp_path):
    """The same audit for the instantiations BASELINE configs[4] runs (12 rows: k_null<11, false, false> and k_null_occ<11>, the
    latter compiled for 96 registers with some twenty spilled)."""
    txt = _compile_unit(tmp_path, "rc_null_b")
    names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
    for want in ("k_nullILi11ELb0ELb0ELb0ELi0E", "k_null_occILi11E"):
        audited, fetches = _audit_pending_registers(txt, next(n for n in names if want in n))
        assert audited >= 4 and fetches >= 24, (want, audited, fetches)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_wide_simulation_and_native_dp_stay_lean(tmp_path):
    """Round 5 rebuilt two kernels around what their ISA showed.  k_generic_sim (blocks of more than 64 rows) had ~50 scalar
    instructions and nine scalar loads per tree node -- 215 s_load in the kernel, a vector load of the parent's index with a full
    wait behind it -- and is now fed by coalesced lane loads, ds_bpermute and v_readlane: a handful of scalar loads in all.
    k_native_dp<N-1> branched per sequence on z and copied all 3 (N-1) states at every join (1598 lines of ISA at ten rows, 20 v_mov
    per sequence and codon): a codon where no sequence is out of frame must be ONE straight-line block of 3 adds and a max3 per
    sequence, and the out-of-frame codons a second one that selects with the wave-uniform z."""
    txt = _compile_unit(tmp_path, "rc_kernels")
    for variant in ("ILb0EE", "ILb1EE"):   # the wide-block layout, and the tile-major one for k_tiled_dp (round 6)
        sim = txt[txt.index("_ZN2rc13k_generic_sim" + variant):]
        sim = sim[sim.index(":\n"):sim.index(".Lfunc_end")]
        assert sim.count("s_load_") <= 40, sim.count("s_load_")
        assert sim.count("ds_bpermute_b32") >= 32 and sim.count("v_readlane_b32") >= 16      # thresholds and control words of eight nodes
        assert "scratch_" not in sim
    name = "_ZN2rc11k_native_dpILi9EEEvNS_10NativeArgsE"
    body = txt[txt.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    blocks, cur = [], None
    for ln in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", ln):
            cur = []
            blocks.append(cur)
        elif cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
            cur.append(ln.strip())
    # (the compiler hoists the in-frame sums in front of the test of z: the in-frame block is what remains, nine max3 and the running sum)
    fast = [b for b in blocks if sum(x.startswith("v_max3_f32") for x in b) == 9 and len(b) <= 60 and not any(x.startswith("v_cndmask") for x in b)]
    assert fast, "no short straight-line block for a codon where every sequence is in frame"
    assert not any(x.startswith("s_cbranch") for x in fast[0][:-2])
    select = [b for b in blocks if sum(x.startswith("v_cndmask_b32") for x in b) >= 9 * 6]
    assert select and sum(x.startswith("s_cbranch") for x in select[0]) <= 4, "the out-of-frame codons branch per sequence again"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_tiled_kernels_keep_four_wavefronts_per_simd_and_spill_nothing_in_a_cell(tmp_path):
    """k_tiled_dp<KT> (rc_null_tiled.h, blocks of 34..64 rows) exists because a SIMD needs four wavefronts to issue at its rate: every instantiation
    must fit 128 VGPRs and 10 KB of LDS (sixteen workgroups per CU), and whatever the register allocator spills to get there must stay out of the
    cell loops -- the blocks that hold a tile's sigma look-ups (a ds_read_b32 per sequence).  (A ring of two or three register sets for a deeper
    prefetch failed exactly this: 44..190 registers spilled into the cells.)"""
    txt = _compile_unit(tmp_path, "rc_null_t")
    names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
    seen = 0
    for kt in (12, 13, 14, 15):
        for sem in ("Lb0E", "Lb1E"):
            name = next(n for n in names if "k_tiled_dpILi%dE%s" % (kt, sem) in n)
            meta = txt[txt.index(".name:           " + name):]
            assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 128, name
            head = txt[txt.rindex(".amdhsa_kernel " + name):]
            assert int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", head).group(1)) <= 10240, name
            body = txt[txt.index(name + ":"):]
            body = body[:body.index(".Lfunc_end")]
            blocks, cur = [], None
            for ln in body.split("\n"):
                if re.match(r"^\.LBB\d+_\d+:", ln):
                    cur = []
                    blocks.append(cur)
                elif cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
                    cur.append(ln.strip())
            cells = [b for b in blocks if sum(x.startswith("ds_read_b32") for x in b) >= kt - 1]
            assert cells, name
            for b in cells:
                assert not any(x.startswith("scratch_") for x in b), name + ": a spill inside a cell\n" + "\n".join(b)
                seen += 1
    assert seen >= 16


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("unit,rows", [("rc_null_c", range(16, 22)), ("rc_null_d", range(22, 32))])
def test_kernels_held_to_one_more_wavefront_keep_their_cells_clean(tmp_path, unit, rows):
    """Round 6: k_null<N-1> for 17..22 rows is held to 128 VGPRs (four wavefronts per SIMD), for 25..30 rows to 168 (three) -- a SIMD needs four
    wavefronts to issue at its rate (rc_null_kernel.h, null_min_waves).  The register allocator spills 3..35 registers to get there: the
    straight-line cells must stay (nearly) free of them -- none up to 21 rows and from 25 to 29, at most two scratch accesses in a cell at 22 and
    30 --, and no register an asm fetch leaves pending may be copied or spilled in front of its wait (the audit of the 7- and 12-row kernels
    above), in any of the one-row kernels of 17..32 rows."""
    txt = _compile_unit(tmp_path, unit)
    names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
    for nk in rows:
        name = next(n for n in names if "k_nullILi%dELb0ELb0ELb0ELi0E" % nk in n)
        meta = txt[txt.index(".name:           " + name):]
        vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        assert vgprs <= (128 if nk <= 21 else 168 if (nk <= 23 or 24 <= nk <= 29) else 256), (nk, vgprs)
        body = txt[txt.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        blocks, cur = [], None
        for ln in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", ln):
                cur = []
                blocks.append(cur)
            elif cur is not None and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
                cur.append(ln.strip())
        cells = [b for b in blocks if sum(x.startswith("ds_read_b32") for x in b) >= nk - 1]   # a sigma look-up per sequence: a straight-line cell
        assert len(cells) >= 4, (nk, len(cells))
        worst = max(sum(x.startswith("scratch_") for x in b) for b in cells)
        assert worst <= (0 if nk <= 20 or 22 <= nk <= 28 or nk >= 30 else 2), (nk, worst)
        audited, fetches = _audit_pending_registers(txt, name)
        assert audited >= 4, (nk, audited, fetches)
