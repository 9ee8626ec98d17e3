import sys, time, numpy as np
sys.path.insert(0, '.')
import ligero_amd, bench
rows,k = 344,128
for nctx in (1,2,4,8):
    b = 64//nctx
    cs = [ligero_amd.LigeroCommitter(rows=rows,k=k,batch=b) for _ in range(nctx)]
    for i,c in enumerate(cs):
        c.upload(bench.synthetic_preenc(i, b*rows*k).reshape(b*rows,k,4))
    for _ in range(3):
        for c in cs: c.commit_resident()
    for c in cs: c.sync()
    t0=time.perf_counter()
    steps=30
    for _ in range(steps):
        for c in cs: c.commit_resident()
    for c in cs: c.sync()
    dt=(time.perf_counter()-t0)/steps
    print(nctx, 'contexts x batch', b, ': %.3f ms per 64 commits' % (dt*1e3))
    for c in cs: c.close()
