#!/bin/bash
# Long-block shape (1000 blocks x 300 cols x 12 seqs, n = 10 000): k_null time and HBM-side traffic when fewer workgroups per CU are
# resident (RC_GRID_CAP_PER_CU), i.e. when the per-wavefront code scratch of all resident wavefronts together is smaller.
#   16 workgroups per CU (default): 4096 x 77 KB per strand x frame = 315 MB > 256 MB MALL + 32 MB L2
#   12: 236 MB, 8: 158 MB
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5_residency
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-extra"
for cap in 0 12 8; do
  export RC_GRID_CAP_PER_CU=$cap
  python3 "$REPO/bench.py" $ARGS > "$OUT/bench_cap$cap.json" 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_cap$cap" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/fetch_cap$cap.log"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_cap$cap" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/write_cap$cap.log"
  python3 - "$OUT" $cap <<'PY'
import csv, glob, json, sys
out, cap = sys.argv[1], sys.argv[2]
d = json.load(open(f"{out}/bench_cap{cap}.json"))
res = {}
for name in ("fetch", "write"):
    tot = cnt = 0
    for f in glob.glob(f"{out}/{name}_cap{cap}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_null<11, false, false" in row["Kernel_Name"]:
                tot += float(row["Counter_Value"]); cnt += 1
    res[name] = tot / max(cnt, 1) * 1024
alg = d["roofline"]["algorithmic_bytes_per_launch"]
print(f"workgroups per CU cap {cap or 16}: k_null {d['roofline']['kernel_ms']:.1f} ms, step {d['ms_per_step']:.1f} ms, FETCH_SIZE {res['fetch'] / 1e9:.1f} GB, WRITE_SIZE {res['write'] / 1e9:.1f} GB per launch "
      f"= {(res['fetch'] + res['write']) / alg:.1f} x algorithmic uncorrected, {(2 * res['fetch'] + res['write']) / alg:.1f} x with the 2 x FETCH correction; valu fraction {d['roofline']['valu']['frac']:.3f}")
PY
  find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*agent_info*" -delete
done
