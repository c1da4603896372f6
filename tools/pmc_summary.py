import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "bmu" not in k and "segsum" not in k: continue
            print(k)
            for c, v in sorted(acc[k].items()):
                print("   %-32s %16.0f  (per dispatch %14.0f)" % (c, v, v / cnt[(k, c)]))
