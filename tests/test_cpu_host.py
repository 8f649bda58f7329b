"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the BAM decoder reproduces the generator's SoA, the product refuses to run without a GPU."""
import os
import re
import tempfile

import numpy as np
import pytest

from breakid_amd import abi, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "breakid_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|void|uint64_t|uint32_t|const char \*)\s*(bk_[a-z_0-9]+)\s*\(", txt, re.M)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    L = capi.lib()
    syms = _declared_symbols()
    assert set(capi.EXPORTS) <= set(syms)
    for s in syms:
        assert hasattr(L, s), s


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.BreakIDError) as e:
        capi.Context([("chr1", 1000)])
    assert e.value.code == abi.BK_ERR_NO_DEVICE


def test_bam_decoder_matches_generator():
    ds = synth.make_g1()
    ds.recs[5].oc = "50M50S"
    ds.recs[5].sa = "chr2,100,+,50S50M,60,0;"
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p)
        contigs, cols = capi.decode_bam(p)
    ref = ds.to_soa()
    assert contigs == ds.contigs
    for k, _ in abi.SOA_COLS_ALL:
        assert np.array_equal(cols[k], ref[k]), k


def test_bam_decoder_threads_and_chunks_agree(monkeypatch):
    """parallel BGZF inflate / chunked record decode (bam_reader.cc): > 1 decode chunk (65536 records each), several BGZF
    blocks, SA/OC blobs crossing chunk borders; 1 thread and 5 threads must give the same table; truncation is an error"""
    contigs = [("chr1", 3_000_000), ("chr2", 2_000_000)]
    ds = synth.make_cfg(5, contigs, 150_000, 40, 30, 200, jitter=200, read_len=100)
    for i in range(0, len(ds.recs), 977):
        ds.recs[i].sa = "chr2,%d,+,40S60M,60,0;" % (100 + i)
        if i % 2:
            ds.recs[i].oc = "60M40S"
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p)
        out = {}
        for th in ("1", "5"):
            monkeypatch.setenv("BREAKID_THREADS", th)
            out[th] = capi.decode_bam(p)[1]
        raw = open(p, "rb").read()
        open(p, "wb").write(raw[: len(raw) // 2])
        with pytest.raises(capi.BreakIDError) as e:
            capi.decode_bam(p)
        assert e.value.code == abi.BK_ERR_IO
    assert len(ref["tid"]) > 2 * 65536
    for k, _ in abi.SOA_COLS_ALL:
        assert np.array_equal(out["1"][k], ref[k]), k
        assert np.array_equal(out["5"][k], ref[k]), k


def test_qname_hash_matches_python():
    L = capi.lib()
    for s in [b"", b"a", b"read/1", b"L12_3", b"x" * 200]:
        assert L.bk_qname_hash(s, len(s)) == synth.fnv1a64(s)


def test_abi_struct_sizes():
    assert abi.PAIR.itemsize == 56 and abi.SPLIT.itemsize == 88 and abi.CLUSTER.itemsize == 72


def test_bench_gpus_flag_is_honoured_or_refused():
    """bench.py --gpus N must never print a line for fewer ranks than it was asked for (this container has no GPU: N = 2 is refused
    before anything is imported that could touch one; a launcher whose WORLD_SIZE disagrees with --gpus is refused as well)"""
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(n, 2))], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to run" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
