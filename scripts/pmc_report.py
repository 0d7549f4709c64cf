#!/usr/bin/env python3
"""Mean per-dispatch PMC values of k_encode_mlp from rocprofv3 counter-collection CSVs."""
import collections, csv, glob, sys
for d in [a for a in sys.argv[1:] if a != 'k_linear']:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for kern in ("k_encode_mlp", "k_mlp_dgrad", "k_mlp_wgrad_frag", "k_mlp_wgrad<8, 2", "k_mlp_wgrad<4, 8", "k_mlp_wgrad<4, 1", "k_reduce_scatter", "k_linear_f32"):
                if kern in name:
                    agg[(kern if kern != "k_encode_mlp" else name.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        print(d)
        for k in sorted(agg):
            v = agg[k]
            print(f"  {k[0]:42s} {k[1]:28s} mean {sum(v) / len(v):.5g}  (n={len(v)})")
