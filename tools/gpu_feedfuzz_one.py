"""One case of tools/gpu_feedfuzz.py (python tools/gpu_feedfuzz_one.py <case> <seed0>) - to look at a failure again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = [sys.argv[0]], sys.argv[1:]
import tools.gpu_feedfuzz as ff
case, seed0 = int(args[0]), int(args[1])
print("case %d: %s" % (case, "ok" if ff.one(case, seed0 * 100003 + case) else "FAILED"), flush=True)
