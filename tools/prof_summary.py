"""Prints the top kernels of a rocprofv3 --stats csv directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-70s calls %5s avg_us %9.2f total_ms %8.2f pct %s" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
