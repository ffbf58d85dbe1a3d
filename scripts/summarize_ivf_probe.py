"""One line per training variant of a scripts/probe_ivf_recall.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["workload"], "| true top-10 inside the query's cluster:", d["true_top10_in_query_cluster"])
for v in d["variants"]:
    print(f'{v["training"]:58s} build {v["build_s"]:5.1f} s  cohesion {v["cluster_cohesion"]:.4f}  lists/cluster {v["lists_per_cluster"]:6.2f}  '
          f'recall@10 at nprobe 1/4/16/64/128 {[x["recall_at_10"] for x in v["sweep"]]}  own-cluster list probed {[x["own_cluster_list_probed"] for x in v["sweep"]]}')
