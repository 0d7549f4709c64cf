#!/usr/bin/env python3
"""Mean per-dispatch PMC values of k_encode_mlp from rocprofv3 counter-collection CSVs."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_encode_mlp" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d)
        for k in sorted(agg):
            v = agg[k]
            print(f"  {k:30s} {sum(v) / len(v):.5g}")
