import sys, os, json; sys.path.insert(0, "/root/repo")
import bench
for cfg in sys.argv[1:]:
    os.environ["LIGERO_BENCH_PROVERS"] = cfg
    r = bench.full_prover_rate(0)
    print(cfg, round(r["value"]), round(r["ms_per_64_proofs"], 2), flush=True)
