import csv, sys, collections
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mg_" not in n: continue
    acc[n.split("(")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in acc.items():
    print(k[:60], {a: f"{b:.3g}" for a, b in v.items()})
