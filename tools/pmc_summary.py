"""Per-kernel HBM traffic from two rocprofv3 PMC passes of the bench command (FETCH_SIZE pass, WRITE_SIZE pass):
   python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request on
wide coalesced reads -> doubled; WRITE_SIZE is exact; both are in KiB.  Cross-check: adam_clamp_multi_kernel reads 4 and
writes 3 tensors of the parameter size (61.4 M floats) -> 983 MB / 737 MB expected."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k in f:
        n = f[k][0]
        fetch = 2.0 * f[k][1] / n * 1024.0
        write = (w[k][1] / w[k][0] * 1024.0) if k in w and w[k][0] else 0.0
        out[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": fetch + write}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
                         "--no-cpu-baseline; FETCH_SIZE x2 (gfx950), KiB -> bytes", "kernels": out}, open(sys.argv[3], "w"), indent=1)


main()
