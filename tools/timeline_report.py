"""Offline reading of tools/gpu_timeline.sh's kernel timeline: per hardware queue of the last bench step, the kernels in order with
start (ms from the step's k_stream), duration and the gap in front of them.  Usage: python tools/timeline_report.py [csv.gz] [--all]"""
import csv, gzip, sys
fn = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/timeline/kernel_trace_product.csv.gz"
rows = [(r["name"], int(r["queue"]), int(r["start"]), int(r["end"]), int(r["grid"] or 0), int(r["wg"] or 0)) for r in csv.DictReader(gzip.open(fn, "rt"))]
rows.sort(key=lambda r: r[2])
starts = [r[2] for r in rows if r[0] == "k_stream"]
t0 = starts[-1]
t1 = max(r[3] for r in rows)
step = [r for r in rows if r[2] >= t0]
print("last step: %d dispatches, %.2f ms" % (len(step), (t1 - t0) / 1e6))
queues = sorted(set(r[1] for r in step))
for q in queues:
    ks = [r for r in step if r[1] == q]
    busy = sum(r[3] - r[2] for r in ks)
    print("queue %d: %d dispatches, busy %.2f ms, first %.2f last %.2f" % (q, len(ks), busy / 1e6, (ks[0][2] - t0) / 1e6, (ks[-1][3] - t0) / 1e6))
if "--all" in sys.argv:
    for q in queues:
        print("==== queue", q)
        prev = None
        for r in [r for r in step if r[1] == q]:
            gap = (r[2] - prev) / 1e3 if prev else 0
            print("%9.3f  +%8.1f us  %9.1f us  grid %8d  %s" % ((r[2] - t0) / 1e6, gap, (r[3] - r[2]) / 1e3, r[4], r[0]))
            prev = r[3]
