#!/usr/bin/env python3
"""Full-size golden roots for the two large BASELINE shapes, from the C restatement (oracle/):

    s20   10 036 x 4096 -> 32 768   (BASELINE.json configs[2])
    s22   20 068 x 8192 -> 65 536   (BASELINE.json configs[3])

Input = bench.synthetic_preenc(seed, rows * k) (seeded uniform field elements; the same generator bench.py and
the -m gpu tests use).  The oracle's streamed variant (orc_encode_commit_streamed: identical byte strings per column,
U never materialised) makes the 42 GB shape feasible on a 64 GB host: about 1 / 5 minutes on 8 cores.
Writes tests/golden/large_roots.json: root, sha256 of the leaf digests and of the inner nodes.

    python tests/golden/make_golden_large.py [s20] [s22]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bench import WORKLOADS, LARGE_SEED, synthetic_preenc  # noqa: E402
from oracle import binding as orc                           # noqa: E402


def main():
    which = sys.argv[1:] or ["s20", "s22"]
    path = os.path.join(HERE, "large_roots.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    threads = min(orc.lib().orc_max_threads(), os.cpu_count() or 1)
    for name in which:
        rows, k, batch = WORKLOADS[name]
        assert batch == 1
        n = 8 * k
        pre = synthetic_preenc(LARGE_SEED, rows * k).reshape(rows, k, 4)
        t0 = time.time()
        r = orc.encode_commit_streamed(pre, k, n, threads=threads, block_rows=64)
        out[name] = {
            "rows": rows, "k": k, "n": n, "seed": LARGE_SEED, "input": "bench.synthetic_preenc(seed, rows * k)",
            "root": r["root"].hex(),
            "leaves_sha256": hashlib.sha256(r["leaves"].tobytes()).hexdigest(),
            "nodes_sha256": hashlib.sha256(r["nodes"].tobytes()).hexdigest(),
            "provenance": "oracle/ligero_oracle.c orc_encode_commit_streamed (model-derived: the Rust reference cannot be built here)",
        }
        print(name, out[name]["root"], f"{time.time() - t0:.0f} s", flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
