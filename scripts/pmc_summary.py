"""Sum rocprofv3 --pmc counter_collection.csv rows per (kernel, counter): total, launches, mean."""
import csv, sys, collections, re
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = re.sub(r"\(.*", "", row.get("Kernel_Name", "?"))[:70]
        c = row.get("Counter_Name", "?")
        tot[(k, c)] += float(row.get("Counter_Value", 0) or 0); cnt[(k, c)] += 1
print("%-70s %-28s %10s %16s %14s" % ("kernel", "counter", "launches", "total", "mean"))
for (k, c), v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%-70s %-28s %10d %16.6g %14.6g" % (k, c, cnt[(k, c)], v, v / cnt[(k, c)]))
