#!/usr/bin/env python3
"""One rank's shard of configs[3] (1250 blocks) as a stream, a few passes, with markers between the passes on stderr:
   RC_TRACE=1 python tools/trace_shard.py [blocks] [host_threads] [sub_blocks] 2> host_stamps.txt       (the library's stage stamps)
   rocprofv3 --kernel-trace -d out -- python3 tools/trace_shard.py; tools/kernel_timeline.py out/.../*kernel_trace.csv 10"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sub = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else 0   # sub-batch sizes (default: rc_stream_plan)
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
st = api.Stream(ctx, params, 3)
times = []
for rep in range(8):
    print(f"---- pass {rep} starts at {time.monotonic() * 1e6:.1f} us", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    rows = []
    for sb in api.score_stream(ctx, m, params, sub, stream=st):
        sb.fits()
        t, _ = sb.timing()
        rows.append((sb.n, (time.perf_counter() - t0) * 1e3, t))
        sb.close()
    times.append((time.perf_counter() - t0) * 1e3)
    print(f"---- pass {rep} ends at {time.monotonic() * 1e6:.1f} us", file=sys.stderr, flush=True)
    if rep == 7:
        for n, done, t in rows:
            print(f"  sub-batch of {n:5d}: results on the host at {done:7.3f} ms; device intervals: start->fit {t['total']:.3f}, k_null {t['null']:.3f}, native {t['native']:.3f}, fit {t['evd_fit']:.3f}")
print(f"{nb} blocks, {threads} host threads, sub-batches {sub or api.stream_plan(ctx, params, nb, 1)}: resident {res_ms:.3f} ms, stream passes " + " ".join(f"{t:.3f}" for t in times))
st.close()
batch.close()
ctx.close()
