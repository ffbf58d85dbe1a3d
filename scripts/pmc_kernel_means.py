"""Mean counter value per launch for every kernel whose name contains a pattern, from rocprofv3 --pmc CSVs.
    python scripts/pmc_kernel_means.py <pattern> <csv>... """
import collections, csv, sys
agg = collections.defaultdict(list)
for src in sys.argv[2:]:
    for r in csv.DictReader(open(src)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:62s} {c:44s} {sum(v)/len(v):16.1f}  (n={len(v)})")
