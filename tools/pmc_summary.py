"""Developer tool: averages the counters of the library's conv kernels over the dispatches in rocprofv3 CSVs."""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "conv_" not in k and "gc::wgrad" not in k or "gc::" not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[k]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    print("==", d)
    for k, cs in acc.items():
        print(k[:100])
        for c, v in sorted(cs.items()):
            v = v[len(v) // 2:]   # skip warm-up dispatches
            print(f"   {c:36s} {sum(v) / len(v):16.1f}")
