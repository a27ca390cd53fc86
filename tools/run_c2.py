"""BASELINE configs[1] (genomic-preprocessed.maf at n = 1000, from the golden fixture) as one batch: latency and device-time breakdown."""
import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from helpers import block_from_golden
from rnacode_amd import api
doc = load_golden("genomic_preprocessed_n1000")
blocks = [block_from_golden(e) for e in doc["blocks"] if "skipped" not in e["ref"]]
ctx = api.Context(0)
api.lib().rc_set_stream_cache(0)
t0 = time.perf_counter(); batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000, seed_base=42)); t1 = time.perf_counter()
batch.run(); 
ts = []
for _ in range(5):
    t = time.perf_counter(); batch.run(); ts.append(time.perf_counter() - t)
sa, cs = batch.work()
print("C2 genomic-preprocessed n=1000: prep %.3f s, run %.2f ms, %.3g col*samples/s, timing %s" % (t1 - t0, min(ts) * 1e3, cs / min(ts), batch.timing()))
