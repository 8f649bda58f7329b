"""What bounded the last feed of a tools/gpu_feedtrace.sh run: H2D copies (busy time of the copy engine, rate), kernels (busy
time, sum of durations).  Usage: python tools/feedtrace_report.py gpurun_out/feedtrace_<tag>"""
import csv, glob, re, sys, collections
d = sys.argv[1]
cp = list(csv.DictReader(open(glob.glob(d + "/*memory_copy_trace.csv")[0])))
kr = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
h2d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cp if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 31
last = h2d[-n:]
t0, t1 = last[0][0], last[-1][1]
busy = sum(b - a for a, b in last)
print("last %d big H2D copies: span %.2f ms, copy engine busy %.2f ms (%.0f %%), mean %.3f ms each" % (n, (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), busy / n / 1e6))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kr if int(r["Start_Timestamp"]) >= t0]
tend = max(e for _, e, _ in ks)
print("kernels from the first of those copies on: span %.2f ms" % ((tend - t0) / 1e6))
ev = sorted([(s, 1) for s, e, _ in ks] + [(e, -1) for s, e, _ in ks])
cur, lastt, busyk, area = 0, t0, 0, 0
for t, dlt in ev:
    if cur > 0:
        busyk += t - lastt
    area += cur * (t - lastt)
    cur += dlt
    lastt = t
print("some kernel running %.2f ms (%.0f %% of the span), mean concurrency while busy %.2f" % (busyk / 1e6, 100.0 * busyk / (tend - t0), area / max(busyk, 1)))
tot = collections.defaultdict(float)
for s, e, nme in ks:
    m = re.search(r"(k_\w+(<[^>]*>)?)", nme)
    tot[m.group(1) if m else nme[:30]] += (e - s) / 1e6
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:7]:
    print("  %-32s %8.2f ms" % (k, v))
