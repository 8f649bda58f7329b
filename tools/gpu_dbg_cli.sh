cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from breakid_amd import synth, bamio
from tools import make_golden
ds, refgene = [(d, r) for n, d, r in make_golden.datasets() if n == "small"][0]
tmp = "/tmp/dbgcli"; os.makedirs(tmp, exist_ok=True)
bam = os.path.join(tmp, "s.bam"); ds.write_bam(bam, aligned=True); bamio.write_bai(bam)
side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
open("/tmp/dbgcli/cmd.txt", "w").write(" ".join(["-i", bam, "-o", tmp + "/out", "-n", side["nib"], "-all", "-fast"]))
open("/tmp/dbgcli/env.txt", "w").write(side["install"])
PY
export BREAKID_INSTALLDIR=$(cat /tmp/dbgcli/env.txt) BK_DEBUG=multi
for g in 1 2 3; do breakid_amd/bin/BreakID $(cat /tmp/dbgcli/cmd.txt) -gpus $g -comm local 2>&1 | grep -E "multi|valid"; cat /tmp/dbgcli/out_performance.txt | tail -1 | cut -f1-5; done
breakid_amd/bin/BreakID $(cat /tmp/dbgcli/cmd.txt) 2>&1 | grep -E "valid"; cat /tmp/dbgcli/out_performance.txt | tail -1 | cut -f1-5
