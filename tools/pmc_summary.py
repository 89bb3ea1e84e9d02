"""Per-kernel averages of rocprofv3 PMC passes.
usage: python tools/pmc_summary.py OUT.csv COUNTER=dir [COUNTER=dir ...]
Each dir is the -d directory of one `rocprofv3 --pmc COUNTER --kernel-trace -- python3 bench.py ...` run."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.split("::")[-1].split("<")[0].strip()


def main():
    out, specs = sys.argv[1], sys.argv[2:]
    table = defaultdict(dict)
    counters = []
    for spec in specs:
        names, d = spec.split("=")   # COUNTER=dir or A+B+C=dir (several counters collected in one pass)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for counter in names.split("+"):
            counters.append(counter)
            acc = defaultdict(lambda: [0.0, 0])
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        k = short(row["Kernel_Name"])
                        acc[k][0] += float(row["Counter_Value"])
                        acc[k][1] += 1
            for k, (s, n) in acc.items():
                table[k][counter] = s / max(n, 1)
                table[k]["dispatches"] = n
    with open(out, "w") as fh:
        fh.write("kernel,dispatches," + ",".join(c + "_avg_per_dispatch" for c in counters) + "\n")
        for k in sorted(table):
            fh.write(f"{k},{table[k].get('dispatches', 0)}," + ",".join(f"{table[k].get(c, float('nan')):.3f}" for c in counters) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
