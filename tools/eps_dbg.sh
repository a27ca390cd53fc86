cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import load_golden
from helpers import block_from_golden
doc=load_golden("coding_aln_n100")
b=block_from_golden(doc["blocks"][0])
open("/tmp/in.aln","w").write("CLUSTAL W (1.83) multiple sequence alignment\n\n" + "".join(f"{r.name:<40s}{r.seq}\n" for r in b.rows) + "\n")
open("/tmp/t.tsv","w").write(f"{doc['blocks'][0]['ref']['tree']}\t{doc['blocks'][0]['ref']['kappa']!r}\n")
PY
rnacode_amd/rnacode_hip /tmp/in.aln --trees /tmp/t.tsv -n 100 --seed-base 42 -e -i 0.05 -d /tmp/plots; echo rc=$?; ls -la /tmp/plots | head
