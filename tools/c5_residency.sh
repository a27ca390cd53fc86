#!/bin/bash
# Long-block shape (1000 blocks x 300 cols x 12 seqs, n = 10 000): k_null time and HBM-side traffic when fewer workgroups per CU are
# resident (RC_GRID_CAP_PER_CU), i.e. when the code scratch of all resident workgroups together is smaller.  A workgroup's hot set is
# the codes of one strand x frame, 100 sites x 9 bytes x 64 samples = 57.6 KB:
#   20 workgroups per CU (k_null_occ, the default for a one-class batch): 5120 x 57.6 KB = 295 MB  > 256 MB Infinity Cache
#   16: 236 MB, 12: 177 MB, 8: 118 MB, 6: 88 MB
# The question: is it HBM or the L2 fill rate that binds?  If HBM did, time would drop once the hot set is
# cache-resident; FETCH_SIZE counts at the L2 and cannot tell (it sees Infinity-Cache hits as fetches too).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5_residency
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-extra"
for cap in ${CAPS:-0 16 12 8 6}; do
  export RC_GRID_CAP_PER_CU=$cap
  python3 "$REPO/bench.py" $ARGS --full-json "$OUT/bench_cap$cap.json" > /dev/null 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_cap$cap" -- python3 "$REPO/bench.py" $ARGS --full-json /dev/null > /dev/null 2> "$OUT/fetch_cap$cap.log"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_cap$cap" -- python3 "$REPO/bench.py" $ARGS --full-json /dev/null > /dev/null 2> "$OUT/write_cap$cap.log"
  python3 - "$OUT" $cap <<'PY'
import csv, glob, json, sys
out, cap = sys.argv[1], sys.argv[2]
d = json.load(open(f"{out}/bench_cap{cap}.json"))
kern = d["roofline"]["kernel"]
res = {}
for name in ("fetch", "write"):
    tot = cnt = 0
    for f in glob.glob(f"{out}/{name}_cap{cap}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kern.replace("rc::", "") in row["Kernel_Name"]:
                tot += float(row["Counter_Value"]); cnt += 1
    res[name] = tot / max(cnt, 1) * 1024
alg = d["roofline"]["hbm"]["algorithmic_bytes_per_launch"]
ms = d["roofline"]["kernel_ms"]
print(f"workgroups per CU cap {cap if cap != '0' else 'none'}: {kern} {ms:.1f} ms, step {d['ms_per_step']:.1f} ms, FETCH_SIZE {res['fetch'] / 1e9:.1f} GB, WRITE_SIZE {res['write'] / 1e9:.1f} GB per launch "
      f"= {(2 * res['fetch'] + res['write']) / alg:.1f} x algorithmic with the 2 x FETCH correction, {(2 * res['fetch'] + res['write']) / ms / 1e9:.2f} TB/s; valu fraction {d['roofline']['valu']['frac']:.3f}")
PY
  find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*agent_info*" -delete
done
