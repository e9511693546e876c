"""Turns rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json (read by bench.py for `traffic`).

Collection (separate passes, as MI355X_MICROARCH.md §HBM prescribes; on the GPU box):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <out>/pmc_fetch --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <out>/pmc_write --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
Units/corrections: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
of a wide (16 B/lane) coalesced streaming read, so it is doubled for k_deferred (whose reads are all
16 B/lane).  k_raster's reads are narrow gathers: its FETCH_SIZE is reported uncorrected (lower bound).
"""
import collections
import csv
import glob
import json
import sys


def agg(pattern):
    rows = list(csv.DictReader(open(glob.glob(pattern)[0])))
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def main(fetch_dir, write_dir, out_path):
    f = agg(fetch_dir + "/*/*counter_collection.csv")
    w = agg(write_dir + "/*/*counter_collection.csv")
    out = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        name = k.split("<")[0]
        fetch_kib, write_kib = f.get(k, 0.0), w.get(k, 0.0)
        factor = 2.0 if name == "k_deferred" else 1.0
        out[name] = {"FETCH_SIZE_KiB": round(fetch_kib, 1), "WRITE_SIZE_KiB": round(write_kib, 1), "fetch_correction": factor,
                     "hbm_bytes_per_launch": int((fetch_kib * factor + write_kib) * 1024)}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
