"""Timeline of the LAST backward pass in a rocprofv3 kernel trace (csv): kernels between the last reinforce_loss_kernel and the
next adam_clamp_multi_kernel, in start order, with stream / queue, start offset, duration and the gap to the previous end."""
import csv, sys
ROLL = len(sys.argv) > 2 and sys.argv[2] == "rollouts"      # the rollout pair in front of the last backward pass instead (a window of its middle)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "reinforce_loss_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
STEP = len(sys.argv) > 2 and sys.argv[2] == "step"            # everything between the previous step's Adam and the last loss kernel (the rollouts), unwindowed
if ROLL:
    i1 = i0
    i0 = max(i for i in range(i1) if "mean_feats_kernel" in rows[i]["Kernel_Name"])
if STEP:
    i1 = i0
    i0 = max(i for i in range(i1) if "adam_clamp_multi" in rows[i]["Kernel_Name"]) + 1
    ROLL = False
if not ROLL and not STEP:
    i1 = next(i for i in range(i0, len(rows)) if "adam_clamp_multi" in rows[i]["Kernel_Name"])
t0 = rows[i0]["s"]
qs = {}
def short(n):
    n = n.replace("icz::(anonymous namespace)::", "").replace("icz::", "").replace("void ", "")
    return n.split("(")[0][:44]
print(("rollouts: %.1f us from the prologue to the loss kernel" if ROLL else "backward: %.1f us from the loss kernel to Adam") % ((rows[i1]["s"] - t0) / 1e3))
last_end = t0
for r in rows[i0:i1 + 1]:
    if ROLL and not (1200e3 <= r["s"] - t0 <= 1520e3):
        last_end = max(last_end, r["e"])
        continue
    q = qs.setdefault(r.get("Queue_Id", r.get("Stream_Id", "?")), len(qs))
    print("%8.1f %7.1f us  q%d  gap %6.1f  %s  grid %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, q, (r["s"] - last_end) / 1e3,
                                                   short(r["Kernel_Name"]), r.get("Grid_Size", "")))
    last_end = max(last_end, r["e"])
