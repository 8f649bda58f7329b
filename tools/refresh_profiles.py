"""Copies the summaries of a bench + rocprofv3 session (<dir>/bench_final.log, <dir>/{ks,fetch,write}_*.csv) into profiles/
under a round prefix.  Usage: python tools/refresh_profiles.py gpurun_out/prof_r02 r02

On the GPU box (one gpurun call):
  python bench.py > gpurun_out/prof_r02/bench_final.log
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02/ks_raw -o ks -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r02/f_raw -o fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r02/w_raw -o write -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0
then copy <raw>/*/ks_kernel_stats.csv, fetch_counter_collection.csv, write_counter_collection.csv next to bench_final.log."""
import csv, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, sys.argv[1]) + "/"
R = sys.argv[2] if len(sys.argv) > 2 else "r03"
P = os.path.join(root, "profiles") + "/"
line = [l for l in open(d + "bench_final.log") if l.startswith("{")][-1]
open(P + R + "_bench_620M_1gpu.json", "w").write(line)
shutil.copy(d + "ks_kernel_stats.csv", P + R + "_kernel_stats_620M_1gpu.csv")


def pmc(fn, name):
    tot = {}
    with open(d + fn) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != name:
                continue
            t = tot.setdefault(row["Kernel_Name"], [0.0, set()])
            t[0] += float(row["Counter_Value"])
            t[1].add(row["Dispatch_Id"])
    return tot


f, w = pmc("fetch_counter_collection.csv", "FETCH_SIZE"), pmc("write_counter_collection.csv", "WRITE_SIZE")
keep = ("k_stream", "k_split_records", "k_sd_count", "k_sd_emit", "k_join_pairs", "k_accumulate", "k_bp_cov", "k_bp_depth", "k_bp_vote", "k_bp_regions")
with open(P + R + "_pmc_hbm_stream_kernels.csv", "w") as o:
    o.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,traffic_bytes_per_launch(2*FETCH*1024+WRITE*1024)\n")
    for k in sorted(f):
        if not any(x in k for x in keep):
            continue
        n = max(1, len(f[k][1]))
        fk = f[k][0] / n
        wk = w.get(k, [0.0, {1}])[0] / max(1, len(w.get(k, [0, {1}])[1]))
        name = k[len("(anonymous namespace)::"):].split("(")[0] if k.startswith("(") else k.split("(")[0]
        o.write('"%s",%d,%.3f,%.3f,%d\n' % (name, n, fk, wk, int(2 * fk * 1024 + wk * 1024)))
with open(P + R + "_pmc_raw_stream_kernels.csv", "w") as o:
    for i, fn in enumerate(("fetch_counter_collection.csv", "write_counter_collection.csv")):
        for j, l in enumerate(open(d + fn)):
            if (j == 0 and i == 0) or any(x in l for x in keep):
                o.write(l)
ks = [k for k in f if "k_stream" in k][0]
fk, wk = f[ks][0] / len(f[ks][1]), w[ks][0] / len(w[ks][1])
b = json.loads(line)
j = {"kernel": "k_stream", "workload": "configs[1], 620M records, 1 launch", "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
     "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
     "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
     # bench.py matches this record to its own run by the SURVEY-formula bytes of the launch (same table <=> same value)
     "algorithmic_bytes_per_launch": b["roofline"]["survey_formula_bytes_per_launch"], "own_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
     "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), rocprofv3 --kernel-trace, see profiles/%s_pmc_hbm_stream_kernels.csv" % R}
json.dump(j, open(P + R + "_pmc_k_stream.json", "w"), indent=1)
with open(d + "ks_kernel_stats.csv") as fh:
    for row in csv.DictReader(fh):
        if "k_stream" in row["Name"]:
            print("k_stream rocprof avg %.3f ms; bench HIP-event avg %.3f ms; value %.1f; frac %.4f; traffic %d" % (
                float(row["AverageNs"]) / 1e6, b["roofline"]["avg_launch_ms"], b["value"], b["roofline"]["frac"], j["traffic_bytes_per_launch"]))

for src, dst in (("feedbench.log", "_feedbench_8M_records.log"), ("feed_kernel_stats.csv", "_feed_kernel_stats_8M_records.csv"), ("inflatebench.log", "_inflatebench_570MB.log")):
    if os.path.exists(d + src):
        shutil.copy(d + src, P + R + dst)
