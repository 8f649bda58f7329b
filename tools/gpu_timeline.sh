# kernel timeline of the bench step (start / end of every dispatch, per queue) for offline analysis: tools/timeline_report.py
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/timeline
mkdir -p $D
BK_DEBUG=lanes BREAKID_GROUP_LANES=${LANES:-12} timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $D/raw -o tl -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $D/bench.log 2> $D/bench.err
cp $(find $D/raw -name "*kernel_trace.csv" | head -n 1) $D/kernel_trace.csv
rm -rf $D/raw
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/timeline/kernel_trace.csv")))
keep = [r for r in rows if not r["Kernel_Name"].startswith("void at::") and "rocprim" not in r["Kernel_Name"]]
import gzip
with gzip.open("gpurun_out/timeline/kernel_trace_product.csv.gz", "wt") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "start", "end", "grid", "wg", "lds"])
    for r in keep:
        n = r["Kernel_Name"]
        n = n.replace("(anonymous namespace)::", "").split("(")[0]
        w.writerow([n, r["Queue_Id"], r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", "")])
print(len(rows), len(keep), list(rows[0].keys()))
PY
rm -f $D/kernel_trace.csv
ls -la $D
tail -3 $D/bench.log | cut -c1-600
