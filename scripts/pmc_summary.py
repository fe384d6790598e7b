"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per launch of each counter per srd:: kernel."""
import csv, glob, collections, sys
rows = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "srd::" in k and "<true>" not in k and "<1>" not in k and "<3>" not in k:   # skip the instrumented variants
                rows[(k.split("(")[0].replace("void srd::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    print("%-32s %-26s n=%-3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
