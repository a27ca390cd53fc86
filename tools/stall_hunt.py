#!/usr/bin/env python3
"""Find the step that takes hundreds of times the median (1.6 s among 5 ms steps in bench.py's c2 leg).

Repeats what bench.py does before and inside its c1 / c2 legs -- a resident 10 000-block batch, a streamed pass, then the shipped
files as one batch each, batch after batch -- with RC_TRACE=1, a marker line per step on the same stderr, and lists every step that
took more than 10 x the median together with the library's trace lines inside it ("slow call" lines name the runtime call).

    RC_TRACE=1 python tools/stall_hunt.py [files_per_leg] [--light] 2> trace.txt     (--light: no resident batch / stream before the legs)
    python tools/stall_hunt.py --report trace.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mark(text):
    os.write(2, ("[py %14.1f us] %s\n" % (time.monotonic() * 1e6, text)).encode())


def report(path):
    steps = []   # (leg, index, ms, lines)
    cur = None
    for line in open(path, errors="replace"):
        if line.startswith("[py") and "] begin " in line:
            cur = [line.split("] begin ")[1].strip(), None, []]
        elif line.startswith("[py") and "] end " in line and cur:
            cur[1] = float(line.split(" ms=")[1])
            leg, idx = cur[0].rsplit(" ", 1)
            steps.append((leg, int(idx), cur[1], cur[2]))
            cur = None
        elif cur is not None:
            cur[2].append(line.rstrip())
    legs = {}
    for leg, idx, ms, lines in steps:
        legs.setdefault(leg, []).append((idx, ms, lines))
    for leg, rows in legs.items():
        xs = sorted(ms for _, ms, _ in rows)
        med = xs[len(xs) // 2]
        slow = [(i, ms, ln) for i, ms, ln in rows if ms > 10 * med]
        print("%s: %d steps, min %.3f median %.3f p99 %.3f max %.3f ms, %d over 10 x median" %
              (leg, len(xs), xs[0], med, xs[min(len(xs) - 1, int(0.99 * len(xs)))], xs[-1], len(slow)))
        for i, ms, ln in slow[:6]:
            print("  step %d: %.3f ms" % (i, ms))
            for l in ln:
                if "slow call" in l:
                    print("     " + l)
            stamps = [(float(l.split()[1]), l) for l in ln if l.startswith("[rc ") and "us]" in l and "slow call" not in l]
            for (t0, a), (t1, b) in zip(stamps, stamps[1:]):
                if t1 - t0 > 0.3 * ms * 1e3:
                    print("     %.3f ms between\n       %s\n       %s" % ((t1 - t0) * 1e-3, a, b))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    files = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
    light = "--light" in sys.argv
    import gc
    import bench
    from rnacode_amd import api
    from rnacode_amd.synth import synth_blocks
    ctx = api.Context(0)
    api.lib().rc_set_stream_cache(0)
    keep = []
    if not light:
        blocks = [b.upper() for b in synth_blocks(10000, 6, 120, seed=1)]
        params = api.default_params(sampleN=1000, seed_base=42)
        m = api.Marshalled(blocks)
        m.set_trees()
        batch = api.Batch(ctx, m, params)
        for _ in range(3):
            batch.run()
        sub = api.stream_plan(ctx, params, len(blocks), 1)
        stream = api.Stream(ctx, params, depth=3)
        for _ in range(2):
            lo = 0
            sent = []
            for sz in sub:
                sent.append((lo, lo + sz)); lo += sz
            i = 0
            while i < len(sent) or stream.pending:
                while i < len(sent) and stream.pending < stream.depth:
                    stream.submit(m, sent[i][0], sent[i][1]); i += 1
                sb = stream.next(); sb.fits(); sb.close()
        keep = [batch, stream, m]
    only = [a.split("=")[1] for a in sys.argv if a.startswith("--only=")]
    for key in ("c1", "c2"):
        name, _ = bench.FILE_LEGS[key]
        blocks_f, n, seed, pk = bench.golden_blocks(name)
        params_f = api.default_params(sampleN=n, seed_base=seed, **pk)
        scored = [b for b in blocks_f if b.tree is not None and b.n >= 3]

        def as_batch():
            b = api.Batch(ctx, blocks_f, params_f).run()
            for i in range(b.n):
                if b.status(i) == api.RC_OK:
                    b.scoreAln(i)
            b.close()

        def per_block():
            for blk in scored:
                b = api.Batch(ctx, [blk], params_f)
                b.getModels(0); b.run(); b.scoreAln(0); b.getExtremeValuePars(0)
                b.close()

        for label, fn, reps in (("batch", as_batch, files), ("per_block", per_block, max(10, files // 10))):
            if only and "%s.%s" % (key, label) not in only:
                continue
            fn()
            gc.collect(); gc.disable()
            for i in range(reps):
                mark("begin %s.%s %d" % (key, label, i))
                t0 = time.perf_counter()
                fn()
                mark("end %s.%s %d ms=%.3f" % (key, label, i, (time.perf_counter() - t0) * 1e3))
            gc.enable()
    del keep
    ctx.close()


if __name__ == "__main__":
    main()
