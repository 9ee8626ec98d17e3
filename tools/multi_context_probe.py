"""Do independent commits overlap when issued from several contexts (one stream each)?
(a) the batch split over n contexts; (b) n full-batch contexts in flight (steady-state throughput)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import ligero_amd, bench
rows, k = 344, 128
def run(nctx, b, label):
    cs = [ligero_amd.LigeroCommitter(rows=rows, k=k, batch=b) for _ in range(nctx)]
    for i, c in enumerate(cs):
        c.upload(bench.synthetic_preenc(i, b * rows * k).reshape(b * rows, k, 4))
    for _ in range(3):
        for c in cs: c.commit_resident()
    for c in cs: c.sync()
    t0 = time.perf_counter()
    steps = 30
    for _ in range(steps):
        for c in cs: c.commit_resident()
    for c in cs: c.sync()
    dt = (time.perf_counter() - t0) / steps
    print('%s: %d contexts x batch %d: %.3f ms per round = %.3f ms per 64 commits' % (label, nctx, b, dt * 1e3, dt * 1e3 * 64 / (nctx * b)))
    for c in cs: c.close()
for nctx in (1, 2, 4):
    run(nctx, 64 // nctx, 'split')
for nctx in (2, 3):
    run(nctx, 64, 'full ')
run(2, 128, 'full ')
run(1, 128, 'single')
run(1, 256, 'single')
