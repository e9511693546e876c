"""Turns the rocprofv3 PMC passes collected by tools/collect_profiles.sh into the JSON files bench.py reads.

  python tools/summarize_pmc.py traffic FETCH.csv WRITE.csv OUT.json     # HBM bytes per launch (roofline.traffic)
  python tools/summarize_pmc.py sq SQA.csv SQB.csv [SQC.csv ...] OUT.json   # SQ / GRBM counters per launch (roofline_valu)

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled for the lighting passes
(k_deferred, k_deferred_tiled: all their reads are 16 B/lane).  k_raster's reads are narrow gathers: its FETCH_SIZE is
reported uncorrected (a lower bound).  WRITE_SIZE reads 16-B-per-lane streaming stores exactly.
"""
import collections
import csv
import json
import sys


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"):
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":     # the launch's duration in the same pass: cycles / time = the clock it ran at
                d[k]["GRBM_duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in d.items()}


def main():
    mode, a, b, out_path = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[-1]
    A, B = agg(a), agg(b)
    for extra in sys.argv[4:-1]:                      # further SQ passes (instruction classes, GRBM_GUI_ACTIVE): merged into B
        for k, cs in agg(extra).items():
            B.setdefault(k, {}).update(cs)
    out = {}
    if mode == "traffic":
        for k in sorted(set(A) | set(B)):
            fetch_kib, write_kib = A.get(k, {}).get("FETCH_SIZE", 0.0), B.get(k, {}).get("WRITE_SIZE", 0.0)
            factor = 2.0 if k in ("k_deferred", "k_deferred_tiled") else 1.0
            out[k] = {"FETCH_SIZE_KiB": round(fetch_kib, 1), "WRITE_SIZE_KiB": round(write_kib, 1), "fetch_correction": factor,
                      "hbm_bytes_per_launch": int((fetch_kib * factor + write_kib) * 1024)}
    else:
        for k in sorted(set(A) | set(B)):
            out[k] = {c: round(v, 1) for c, v in sorted({**A.get(k, {}), **B.get(k, {})}.items())}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k in ("k_raster", "k_deferred", "k_deferred_tiled")}, indent=1))


if __name__ == "__main__":
    main()
