"""The tree + kappa estimator (host and GPU) and the native driver on degenerate blocks (tools/soak_exotic.py: one letter, two letters, identical rows,
a row of gaps, a gappy reference): no hang, no failure, same topologies; kappa differs between device and host only where the likelihood does not
depend on it (identical rows, one or two letters: lnL equal to the sixth decimal).  python tools/exotic_trees.py  (on a GPU box)"""
import sys, time, re, subprocess, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from rnacode_amd import api
from rnacode_amd.synth import synth_block, to_maf
from soak_exotic import exotic
rng = np.random.RandomState(5)
blocks, kinds = [], []
while len(blocks) < 300:
    n = int(rng.choice([3, 4, 6, 8, 12, 20])); cols = int(rng.choice([12, 30, 60, 120]))
    b = synth_block(rng, n, cols, index=len(blocks), gaps=True).upper()
    k = None
    for _ in range(50):
        k = exotic(rng, b)
        if k and not k.startswith("kappa") and not k.startswith("branches"):
            break
        k = None
    if k is None:
        continue
    blocks.append(b); kinds.append(k)
for i in range(30):
    b = synth_block(rng, 5, 30, index=1000 + i, gaps=True).upper()
    keep = "GT" if i % 2 else "AC"
    for x in b.rows:
        x.seq = "".join(c if c not in "ACGT" else keep[(ord(c) + q) % 2] for q, c in enumerate(x.seq))
    blocks.append(b); kinds.append("two letters")
for b in blocks:
    b.tree, b.kappa = None, 0.0
ctx = api.Context(0)
t = time.time(); dev = api.fit_trees(blocks, ctx=ctx); print("device estimator: %.2f s" % (time.time() - t))
host = api.fit_trees(blocks)
names = lambda nwk: re.sub(r":[0-9.eE+-]+", "", nwk)
diff_topo = sum(1 for d, h in zip(dev, host) if (d is None) != (h is None) or (d and names(d[0]) != names(h[0])))
diff_kappa = sum(1 for d, h in zip(dev, host) if d and h and abs(d[1] - h[1]) > 1e-3 * max(1.0, abs(h[1])))
print("blocks", len(blocks), "topology differs", diff_topo, "kappa differs", diff_kappa, "None on device", sum(d is None for d in dev))
for b, f in zip(blocks, dev):
    b.tree, b.kappa = f
open("/tmp/exotic.maf", "w").write(to_maf(blocks))
t = time.time()
r = subprocess.run(["rnacode_amd/rnacode_hip", "/tmp/exotic.maf", "-n", "100", "-t", "-o", "/tmp/exotic.tsv"], capture_output=True, text=True, timeout=300)
print("native driver on the MAF (trees fitted on the GPU): rc", r.returncode, "%.2f s" % (time.time() - t), r.stderr[-300:])
print(open("/tmp/exotic.tsv").read().count("\n"), "listing lines")
import collections
c = collections.Counter(); worst = {}
for k, d, h in zip(kinds, dev, host):
    if d and h and abs(d[1] - h[1]) > 1e-3 * max(1.0, abs(h[1])):
        c[k] += 1
        worst.setdefault(k, []).append((round(d[1], 4), round(h[1], 4)))
print(dict(c)); print({k: v[:4] for k, v in worst.items()})
lnl_d, lnl_h = [], []
bad = [i for i, (d, h) in enumerate(zip(dev, host)) if abs(d[1] - h[1]) > 1e-3 * max(1.0, abs(h[1]))][:8]
for i in bad:
    print(kinds[i], "lnL device tree %.6f host tree %.6f" % (api.tree_lnl(blocks[i], dev[i][0], dev[i][1]), api.tree_lnl(blocks[i], host[i][0], host[i][1])))
