"""Developer tool: averages every counter of rocprofv3 counter-collection CSVs per kernel whose name contains a substring.
    python tools/pmc_any.py SUBSTRING DIR [DIR ...]"""
import csv, collections, glob, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if sub not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d)
    for k, cs in sorted(acc.items()):
        print(k[:110])
        for c, v in sorted(cs.items()):
            v = v[len(v) // 2:]   # skip warm-up dispatches
            print(f"   {c:36s} {sum(v) / len(v):18.1f}")
