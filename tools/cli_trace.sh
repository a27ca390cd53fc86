#!/bin/bash
# stage stamps of the native driver on the 10 000-block C3 MAF (RC_TRACE: the library's submit / prepared / uploaded / queued / wait / done marks), orderly vs immediate exit
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, ".")
from rnacode_amd.synth import synth_blocks, to_maf
open("/tmp/c3.maf", "w").write(to_maf(synth_blocks(10000, 6, 120, seed=1)))
PY
for i in 1 2 3; do RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/c3.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | tail -1; sleep 0.3; done
echo "--- with RC_TRACE"
RC_TRACE=1 RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/c3.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | tail -60
echo "--- orderly teardown, wall by /usr/bin/env time"
for i in 1 2 3; do python3 -c "import subprocess,time,os; t=time.time(); subprocess.run([\"rnacode_amd/rnacode_hip\",\"/tmp/c3.maf\",\"-n\",\"1000\",\"-t\",\"-o\",\"/tmp/o.tsv\"], env=dict(os.environ, RC_CLI_TEARDOWN=\"1\")); print(\"teardown run:\", round(time.time()-t,3))"; sleep 0.3; done
for i in 1 2 3; do python3 -c "import subprocess,time; t=time.time(); subprocess.run([\"rnacode_amd/rnacode_hip\",\"/tmp/c3.maf\",\"-n\",\"1000\",\"-t\",\"-o\",\"/tmp/o.tsv\"]); print(\"immediate exit run:\", round(time.time()-t,3))"; sleep 0.3; done
