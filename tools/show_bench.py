import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", "value %.0f ms/step %.3f | kernel avg %.1f us launches %d bytes/launch %.1f MB frac_hbm %.3f frac_mfma %.3f" % (
        d["value"], d["ms_per_step"], r["avg_launch_us"], r["launches"], r["bytes_per_launch"] / 1e6, r["frac"], r["mfma_f32_frac"]))
