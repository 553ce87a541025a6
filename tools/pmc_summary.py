"""Summarises rocprofv3 counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_sweep"
for f in sorted(glob.glob(root + "/*/vm/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k: continue
        acc[k][row["Counter_Name"]].append((row["Dispatch_Id"], float(row["Counter_Value"])))
    for k, cs in acc.items():
        print(f"== {f.split('/')[-3]}  {k[:90]}")
        for c, vals in cs.items():
            per = collections.defaultdict(float)
            for d, v in vals: per[d] += v
            vs = list(per.values())
            print(f"   {c:28s} n={len(vs)} mean={sum(vs)/len(vs):.6e}")
