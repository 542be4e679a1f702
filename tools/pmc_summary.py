"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of
each kernel. Usage: python tools/pmc_summary.py gpurun_out/pmc_*/ [--kernel k_closest]"""
import csv, glob, os, sys, collections
kernel = None
args = [a for a in sys.argv[1:]]
if "--kernel" in args:
    i = args.index("--kernel"); kernel = args[i + 1]; del args[i:i + 2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if kernel and kernel not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:40s} mean/dispatch {sum(v)/len(v):.6g}  (n={len(v)})")
