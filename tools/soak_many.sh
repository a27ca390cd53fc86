#!/bin/bash
# Many soaks side by side on one GPU box (the oracle is single-threaded CPU work and the boxes have cores to spare: fifty processes
# compared 1.24 million blocks in 17 minutes -- how round 5 found its two oldest mismatches).
#   tools/soak_many.sh [mode: "" | fat | wide] [first seed] [processes] [seconds each] [ENV=value ...]
# What a soak cannot match it writes to gpurun_out/soak_fail/*.json: tools/soak_replay.py file.json [alone]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
MODE=${1:-}; FIRST=${2:-1}; PROCS=${3:-10}; SECS=${4:-200}; shift $(( $# < 4 ? $# : 4 ))   # (a plain `shift 4` with fewer arguments shifts NOTHING: "fat 100" then reached env as a program name)
OUT=gpurun_out/soak_many; mkdir -p $OUT
for i in $(seq 0 $((PROCS - 1))); do
  sd=$((FIRST + i))
  (env "$@" timeout $((SECS + 300)) python tools/soak_parity.py $SECS $sd $MODE > $OUT/${MODE:-any}_$sd.txt 2>&1) &
done
wait
grep -h "MISMATCH" $OUT/${MODE:-any}_*.txt | sort -u
grep -h "^soak.*blocks compared" $OUT/${MODE:-any}_*.txt | sed 's/^soak[^:]*:/soak:/' | awk '{n++; b += $4; m += $10; p += $12} END {print "blocks compared", b, "mismatches", m, "backtrack paths", p, "in", n, "soaks"; if (n == 0 || b == 0) {print "NO SOAK RAN (see the logs)"; exit 1}}'
