#!/usr/bin/env python3
"""Wall-clock of the REAL reference (oracle/_ref/BreakID_ref, one thread) and of the oracle port on BASELINE.json
configs[0]/[1-shape at 1 M records] ("config 1" of SURVEY 8(d)) and configs[3] (panel shape, "config 4"), `-fast` and
default (AHC).  Runs in the build container (the reference does not travel); results go to profiles/ and BASELINE.md.

    python tools/time_reference.py [cfg1] [cfg4] [--timeout-s N] > profiles/r02_reference_timing.json
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakid_amd import bamio, fixtures, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def cfg4():
    import torch
    from breakid_amd import synth_gpu
    contigs, cols = synth_gpu.make_panel(12349, torch.device("cpu"), n_loci=500, depth=2000, window=600, contigs=fixtures.PANEL_CONTIGS)
    return fixtures._from_table("cfg4", contigs, synth_gpu.to_numpy_cols(cols), synth.random_refgene(fixtures.PANEL_CONTIGS, 80, 5), nib=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg1", "cfg4"]
    timeout = 4 * 3600
    for a in sys.argv[1:]:
        if a.startswith("--timeout-s="):
            timeout = int(a.split("=")[1])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "port"])
    out = {"host_cores": os.cpu_count(), "cores_used": 1, "rows": []}
    for name in args:
        fx = fixtures.g3() if name == "cfg1" else cfg4()
        n = int(len(fx.cols["tid"]))
        with tempfile.TemporaryDirectory() as tmp:
            bam = os.path.join(tmp, name + ".bam")
            fx.write_bam(bam)
            side = synth.write_side_files(fx.contigs, tmp, refgene_lines=fx.refgene, max_nib_len=60_000_000)
            subprocess.check_call([os.path.join(REF, "ref_index"), bam])
            env = dict(os.environ, BREAKID_REF_INSTALLDIR=side["install"])
            procs = {}
            for mode in ("fast", "default"):
                cmd = [os.path.join(REF, "BreakID_ref"), "-i", bam, "-o", os.path.join(tmp, "o_" + mode), "-n", side["nib"], "-all"] + (["-fast"] if mode == "fast" else [])
                procs[mode] = (time.time(), subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
            for mode, (t0, p) in procs.items():
                try:
                    p.wait(timeout=max(1, timeout - (time.time() - t0)))
                    wall, note = time.time() - t0, "rc %d" % p.returncode
                except subprocess.TimeoutExpired:
                    p.kill()
                    wall, note = time.time() - t0, "stopped after the time limit (lower bound)"
                t1 = time.time()
                o = pyoracle.Oracle(fx.contigs, fx.cols)
                o.run(20, fast=(mode == "fast"))
                port = time.time() - t1
                o.close()
                row = {"config": name, "records": n, "mode": mode, "reference_wall_s": round(wall, 2), "reference_records_per_s": round(n / wall, 1),
                       "port_wall_s": round(port, 2), "port_records_per_s": round(n / port, 1), "note": note}
                out["rows"].append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
