"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory: name, calls, avg us, total ms, %."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["TotalDurationNs"]) / 1e6, 1), r["Percentage"])
