#!/usr/bin/env python3
"""Which sub-batch schedule gives the fastest streamed pass?  tools/stream_plan_sweep.py [blocks] [host_threads]
Schedules are lists of sub-batch sizes; r = blocks that fill the chip's wavefront slots once (slots / sample groups)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
blocks = [b.upper() for b in synth_blocks(nb, 6, 120, seed=1)]
ctx = api.Context(0)
ctx.set_host_threads(threads)
api.lib().rc_set_stream_cache(0)
params = api.default_params(sampleN=1000)
m = api.Marshalled(blocks)
m.set_trees()
batch = api.Batch(ctx, m, params)
for _ in range(3):
    batch.run()
t0 = time.perf_counter()
for _ in range(10):
    batch.run()
res_ms = (time.perf_counter() - t0) / 10 * 1e3
print(f"{nb} blocks, {threads} host threads: resident {res_ms:.3f} ms")
r = 4096 // 16


def fill(head, unit_cap):
    """head, then sub-batches doubling from the last head entry up to unit_cap, the rest in the last one"""
    out, left = [], nb
    for s in head:
        s = min(s, left)
        if s > 0:
            out.append(s); left -= s
    nxt = (head[-1] if head else r) * 2
    while left > 0:
        s = min(left, min(nxt, unit_cap))
        if left - s < r // 2:
            s = left
        out.append(s); left -= s
        nxt *= 2
    return out


def even(k):
    s = -(-nb // k)
    return [min(s, nb - i) for i in range(0, nb, s)]


plans = {
    "even 4 (r02 default for a shard)": even(4),
    "ramp 78,156,313.. (current)": [78, 156] + even(4)[:-1] + [nb - 78 - 156 - sum(even(4)[:-1])] if nb == 1250 else even(16),
    "r/8, r, 2r, 4r..": fill([r // 8, r], 8 * r),
    "r/4, r, 2r, 4r..": fill([r // 4, r], 8 * r),
    "r/8, r, rest": fill([r // 8, r], 1 << 30)[:2] + [nb - r // 8 - r] if nb > r + r // 8 else [nb],
    "r/2, rest": [r // 2, nb - r // 2],
    "r, rest": [r, nb - r],
    "r/8, r/2, 2r, rest": [r // 8, r // 2, 2 * r, nb - r // 8 - r // 2 - 2 * r] if nb > 3 * r else [nb],
    "one batch": [nb],
}
st = api.Stream(ctx, params, 3)
for name, sizes in plans.items():
    sizes = [s for s in sizes if s > 0]
    assert sum(sizes) == nb, (name, sizes)
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        lo = 0
        sent = 0
        while sent < len(sizes) or st.pending:
            while sent < len(sizes) and st.pending < st.depth:
                st.submit(m, lo, lo + sizes[sent]); lo += sizes[sent]; sent += 1
            sb = st.next(); sb.fits(); sb.close()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:])
    print(f"{name:36s} {str(sizes):44s} median {ts[len(ts) // 2]:7.3f} ms  best {ts[0]:7.3f}  ({res_ms / ts[len(ts) // 2] * 100:5.1f} % of resident)")
st.close(); batch.close(); ctx.close()
