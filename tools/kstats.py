"""print kernel name + calls + avg us from a rocprofv3 *_kernel_stats.csv"""
import csv, sys
for row in list(csv.reader(open(sys.argv[1])))[1:]:
    name = row[0].replace("void (anonymous namespace)::", "")[:48]
    if "distribution" in name or "rocclr" in name: continue
    print("  %-50s calls=%s avg_us=%.2f" % (name, row[1], float(row[3]) / 1e3))
