"""SCST step at 8 / 16 / 32 rows (the strong-scaling shards) and the XE step (ragged: rows shrink from 64 to a handful) -- dev tool."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in (8, 16, 32):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", str(b), "--headline-only", "--steps", "10", "--warmup", "3"],
                         capture_output=True, text=True).stdout
    j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print("rows %2d: %.3f ms per SCST step" % (b, j["ms_per_step"]))
print(subprocess.run([sys.executable, os.path.join(root, "tools", "perf_xe.py")], capture_output=True, text=True).stdout.strip().splitlines()[-1])
