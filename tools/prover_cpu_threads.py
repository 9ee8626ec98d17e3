import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["LIGERO_NO_TORCH_PRELOAD"] = "1"
import numpy as np, bench
from ligero_amd.prover import LigeroBatchProver
inst, idx, vals = bench.poseidon_batch_inputs()
batch = 1024
allv = np.ascontiguousarray(vals[np.arange(batch) % 64])
bp = LigeroBatchProver(inst, batch, device=0, threads=4, device_transcript=True)
bp.prove(idx, allv, copy=False)
def threads_cpu():
    out = {}
    for t in os.listdir("/proc/self/task"):
        f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
        out[t] = ((int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"), open(f"/proc/self/task/{t}/comm").read().strip())
    return out
print({k: v for k, v in os.environ.items() if k.split("_")[0] in ("HSA", "HIP", "ROC", "GPU", "AMD", "ROCR")})
a = threads_cpu(); time.sleep(1.0); b = threads_cpu()
print("idle second:", {t: round(v[0] - a.get(t, (0, ""))[0], 3) for t, v in b.items() if v[0] - a.get(t, (0, ""))[0] > 0.005})
a = threads_cpu(); t0 = time.perf_counter(); m0 = time.thread_time()
bp.submit(idx, allv)
for _ in range(9):
    bp.submit(idx, allv); bp.collect()
bp.collect()
dt = time.perf_counter() - t0; b = threads_cpu()
print(f"wall {dt:.3f} s ({10 * batch / dt:.0f} proofs/s), main thread cpu {time.thread_time() - m0:.3f} s")
for t, (v, name) in b.items():
    d = v - a.get(t, (0, ""))[0]
    if d > 0.005: print(t, name, f"{d:.3f} s")
bp.close()
