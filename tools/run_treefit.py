"""Time the tree estimator on the GPU (rc_fit_trees_device) against the host threads (rc_fit_trees)
and report how far the two agree.  usage: python tools/run_treefit.py [n_blocks] [n_seqs] [cols]"""
import re
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api, synth  # noqa: E402


def lengths(nwk):
    return [float(x) for x in re.findall(r":([0-9.]+)", nwk)]


def topo(nwk):
    return re.sub(r":[0-9.]+", "", nwk)


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cols = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    blocks = synth.synth_blocks(nb, n, cols, seed=1)
    ctx = api.Context(0)
    api.fit_trees(blocks[:8], ctx=ctx)
    ll = []
    t = time.time(); dev = api.fit_trees(blocks, ctx=ctx, lnl=ll); t_dev = time.time() - t
    t = time.time(); host = api.fit_trees(blocks); t_host = time.time() - t
    same = sum(1 for a, b in zip(dev, host) if topo(a[0]) == topo(b[0]))
    dl = max(max(abs(x - y) for x, y in zip(lengths(a[0]), lengths(b[0]))) for a, b in zip(dev, host) if topo(a[0]) == topo(b[0]))
    dk = max(abs(a[1] - b[1]) / b[1] for a, b in zip(dev, host))
    print(f"blocks {nb} x {n} x {cols}: device {t_dev:.3f} s, host threads {t_host:.3f} s; same topology {same}/{nb}, "
          f"max |dlen| {dl:.2e}, max rel dkappa {dk:.2e}")


if __name__ == "__main__":
    main()
