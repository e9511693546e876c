"""Per-kernel averages of rocprofv3 --pmc SQ passes: python tools/summarize_sq.py DIR..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    files = glob.glob(d + "/*/*counter_collection.csv")
    if not files:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k in sorted(acc):
        print("  %-28s" % k[:28], " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), "n=%d" % len(next(iter(acc[k].values()))))
