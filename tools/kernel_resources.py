#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the built library, from the code objects' metadata (no GPU needed):
tools/kernel_resources.py [out.json]  ->  {demangled kernel name: {vgpr, agpr, sgpr, spill_vgpr, scratch_bytes, lds_bytes, waves_per_simd}}

waves_per_simd is what the unified 512-entry register file of a gfx950 SIMD allows (VGPRs + AGPRs, allocation granule 8, at most 8): the
occupancy bound of a kernel that asks for no LDS.  tools/rows_sweep.py joins this with the kernel name a batch reports."""
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def waves_per_simd(vgpr: int, agpr: int) -> int:
    # (.vgpr_count of a gfx90a+ code object is already the unified total: architectural VGPRs rounded up to the AGPR offset, plus the AGPRs)
    total = -(-max(vgpr, 1) // 8) * 8
    return max(1, min(8, 512 // total))


def kernel_resources(build_dir: str = os.path.join(ROOT, "rnacode_amd", "csrc", "build")) -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(build_dir, "rc_*.o"))):
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for co in glob.glob(local + ".*gfx950"):
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
                for chunk in notes.split("\n  - .agpr_count:")[1:]:
                    chunk = ".agpr_count:" + chunk
                    def field(name, default=0):
                        m = re.search(r"\.%s:\s+(\S+)" % name, chunk)
                        return m.group(1) if m else default
                    name = field("name", "")
                    rec = dict(vgpr=int(field("vgpr_count")), agpr=int(field("agpr_count")), sgpr=int(field("sgpr_count")),
                               spill_vgpr=int(field("vgpr_spill_count")), spill_sgpr=int(field("sgpr_spill_count")),
                               scratch_bytes=int(field("private_segment_fixed_size")), lds_bytes=int(field("group_segment_fixed_size")),
                               unit=os.path.basename(obj)[:-2])
                    rec["waves_per_simd"] = waves_per_simd(rec["vgpr"], rec["agpr"])
                    out[name] = rec
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    res = {}
    for n, d in zip(names, dem):
        d = re.sub(r"^void ", "", d)
        d = d.replace("(bool)1", "true").replace("(bool)0", "false")
        d = re.sub(r"\(int\)(\d+)", r"\1", d)
        d = re.sub(r"\(rc::.*$|\(unsigned.*$|\(double.*$|\(float.*$|\(int.*$|\(.*\)$", "", d)
        res[d] = out[n]
    return res


if __name__ == "__main__":
    r = kernel_resources()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(r, f, indent=1, sort_keys=True)
    for k in sorted(r, key=lambda s: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", s)]):
        v = r[k]
        print(f"{k:60s} vgpr {v['vgpr']:3d} agpr {v['agpr']:3d} sgpr {v['sgpr']:3d} spill {v['spill_vgpr']:4d} scratch {v['scratch_bytes']:5d} B lds {v['lds_bytes']:5d} waves/SIMD {v['waves_per_simd']}")
