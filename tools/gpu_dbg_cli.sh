cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, tempfile, subprocess
sys.path.insert(0, ".")
from breakid_amd import synth, bamio
from tools import make_golden
ds, refgene = [(d, r) for n, d, r in make_golden.datasets() if n == "g1"][0]
tmp = "/tmp/dbgcli"; os.makedirs(tmp, exist_ok=True)
bam = os.path.join(tmp, "g1.bam"); ds.write_bam(bam, aligned=True); bamio.write_bai(bam)
side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
open("/tmp/dbgcli/cmd.txt", "w").write(" ".join(["breakid_amd/bin/BreakID", "-i", bam, "-o", tmp + "/out", "-n", side["nib"], "-all", "-gpus", "4", "-comm", "local", "-fast"]))
open("/tmp/dbgcli/env.txt", "w").write(side["install"])
PY
export BREAKID_INSTALLDIR=$(cat /tmp/dbgcli/env.txt) BK_ABORT_ON_BAD_ALLOC=1
$(cat /tmp/dbgcli/cmd.txt) 2>&1 | tail -30
