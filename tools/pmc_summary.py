"""Average PMC counters per kernel from a rocprofv3 --pmc csv (counter_collection.csv)."""
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
