import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rnacode_amd import api
from rnacode_amd.synth import synth_blocks
ctx = api.Context(0)
for nb, rows, cols in ((4000, 9, 150), (4000, 12, 150), (2000, 13, 120), (1000, 21, 90), (600, 32, 60), (300, 40, 45)):
    blocks = [b.upper() for b in synth_blocks(nb, rows, cols, seed=5)]
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=64))
    os.environ["X"] = "1"
    for _ in range(2): batch.run()
    acc = 0.0
    for _ in range(5):
        batch.run(); t, _n = batch.timing(); acc += t["native"]
    print(f"{nb} blocks of {rows} x {cols}: native {acc/5:.3f} ms", flush=True)
    batch.close()
ctx.close()
