#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files (the launches after the first of each kernel).
usage: pmc_summary.py <dir-or-csv> [...]   FETCH_SIZE / WRITE_SIZE are printed in GB (x1024 bytes; FETCH x2 per MI355X_MICROARCH.md's gfx950 note)."""
import collections, csv, glob, os, sys

def files(p):
    return [p] if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True))

for arg in sys.argv[1:]:
    for f in files(arg):
        print(f)
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "airband" not in n or "siggen" in n:
                continue
            n = n.replace("void airband::", "").replace("(anonymous namespace)::", "").split("(")[0]
            per[n][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for n, cs in per.items():
            out = []
            for c, v in cs.items():
                v = v[1:] if len(v) > 1 else v
                m = sum(x for x, _ in v) / len(v)
                if c == "FETCH_SIZE":
                    out.append("%s %.2f GB" % (c, m * 1024 * 2 / 1e9))
                elif c == "WRITE_SIZE":
                    out.append("%s %.2f GB" % (c, m * 1024 / 1e9))
                else:
                    out.append("%s %.4g" % (c, m))
            dur = [t for v in cs.values() for _, t in (v[1:] if len(v) > 1 else v)]
            print("   %-60s %5.2f ms  %s" % (n[:60], sum(dur) / len(dur) / 1e6, "  ".join(out)))
