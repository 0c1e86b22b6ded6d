"""timeline.py <kernel_trace.csv> [memory_copy_trace.csv] [n_last]: the last n_last ms-scale window of a rocprofv3 trace as one line per
dispatch / copy: start and duration in microseconds relative to the window's first event"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dbgk::", "")[:50]))
if len(sys.argv) > 2 and sys.argv[2] != "-":
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "")) ))
ev.sort()
# the last region: from the last k_zero_list (dbgk_reset) on
last = max(i for i, e in enumerate(ev) if "k_zero_list" in e[2])
# step back to the reset before: the final reset belongs to the region we want only if followed by work
starts = [i for i, e in enumerate(ev) if "k_zero_list" in e[2]]
i0 = starts[-1]
if len(ev) - i0 < 8 and len(starts) > 1: i0 = starts[-2]
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%10.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
