"""Summarise rocprofv3 --pmc CSV passes: per (kernel, counter) average over dispatches."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "sbmc" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            key = (short, row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-50s %-22s avg=%.4g n=%d" % (k, c, s / n, n))
