"""Per-kernel means of the counters of one rocprofv3 --pmc csv directory."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
want = sys.argv[2:] 
for k in acc:
    if want and not any(w in k for w in want): continue
    print(k)
    for c in sorted(acc[k]): print("   %-28s %14.1f  (n=%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
