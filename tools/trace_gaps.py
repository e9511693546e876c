"""Idle time between the two big kernels of a frame, from a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py DIR [--window]"""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:24], r.get("Queue_Id")) for r in rows)
big = [e for e in ev if e[2].startswith("k_raster") or e[2].startswith("k_deferred")]
n = len(big)
gr, gd = [], []
for a, b in zip(big[n // 2:], big[n // 2 + 1:]):
    (gr if a[2].startswith("k_raster") else gd).append((b[0] - a[1]) / 1000)
print("tile pass -> lighting gap us: median %.1f mean %.1f" % (statistics.median(gr), statistics.mean(gr)))
print("lighting -> tile pass gap us: median %.1f mean %.1f" % (statistics.median(gd), statistics.mean(gd)))
if "--window" in sys.argv:
    i = n // 2
    t0 = big[i][0]
    for e in ev:
        if big[i][0] - 5000 <= e[0] <= big[i + 3][1]:
            print("%9.1f %9.1f %s q%s" % ((e[0] - t0) / 1000, (e[1] - t0) / 1000, e[2], e[3]))
