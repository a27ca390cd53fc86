#!/bin/bash
# SQ counters of the wide-block kernels (k_generic_sim, k_generic_dp) for one library build: tools/pmc_wide.sh [lib.so] [time_wide args]  (every profiler run under its own timeout)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
LIB=${1:-}; shift
ARGS="${*:-100 300 256 1000}"
OUT=$REPO/gpurun_out/pmc_wide
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
[ -n "$LIB" ] && export RC_LIB_PATH=$REPO/$LIB
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/time_wide.py" $ARGS > /dev/null 2> "$OUT/p$i.log" || echo "set $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc_wide'
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for kn in ('k_generic_sim', 'k_generic_dp'):
            if kn in r['Kernel_Name']:
                acc[(kn, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (kn, k), v in sorted(acc.items()): print(f"{kn:14s} {k:28s} sum={sum(v):.6g} launches={len(v)}")
PY
find "$OUT" -name "*agent_info*" -delete
