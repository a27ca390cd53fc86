"""End-to-end timing of the stand-alone driver on a synthetic 10k-block MAF (file -> trees -> scores -> listing).
usage: python tools/run_cli_10k.py [n_blocks] [mixed]     mixed: 3..12 rows, 60..200 columns instead of 6 x 120"""
import os
import sys
import tempfile
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import cli  # noqa: E402
from rnacode_amd.synth import synth_blocks, to_maf  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = tempfile.mkdtemp()
maf = os.path.join(d, "in.maf")
if len(sys.argv) > 2 and sys.argv[2] == "mixed":
    import numpy as np
    from rnacode_amd.synth import synth_block
    rng = np.random.RandomState(5)
    blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True) for i in range(nb)]
else:
    blocks = synth_blocks(nb, 6, 120, seed=1)
with open(maf, "w") as fh:
    fh.write(to_maf(blocks))
t = time.time()
rc = cli.main([maf, "-n", "1000", "-t", "-o", os.path.join(d, "out.tsv")])
dt = time.time() - t
print(f"cli rc={rc}: {nb} blocks end to end in {dt:.2f} s ({os.path.getsize(maf) / 1e6:.1f} MB MAF, "
      f"{sum(1 for _ in open(os.path.join(d, 'out.tsv')))} HSS lines)")
# the native driver (rc_cli.cpp), a fresh process each time: library load and context creation included
import subprocess  # noqa: E402
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rnacode_amd", "rnacode_hip")
for _ in range(2):
    t = time.time()
    r = subprocess.run([exe, maf, "-n", "1000", "-t", "-o", os.path.join(d, "out_native.tsv")], capture_output=True, text=True,
                       env=dict(os.environ, RC_CLI_TIMES="1"))
    print(r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")
    dt = time.time() - t
    same = open(os.path.join(d, "out_native.tsv")).read() == open(os.path.join(d, "out.tsv")).read()
    print(f"native rc={r.returncode}: {nb} blocks end to end in {dt:.2f} s, listing identical to the Python driver's: {same}")
