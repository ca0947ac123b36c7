import csv, glob, sys, collections
# usage: pmc_sum.py <dir> : per kernel and counter, the sum over dispatches / number of dispatches
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    if "prop_kernel" in k or "pool" in k:
        print("%-62s %-14s per launch %.6g (launches %d)" % (k, c, v / len(disp[(k, c)]), len(disp[(k, c)])))
