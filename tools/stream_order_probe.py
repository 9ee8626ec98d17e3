"""Does the hardware queue a stream lands on move the Poseidon step or the single-commitment figures, and does the library's
choice of streams (pick_pipeline_streams, context.hip; LG_PICK_STREAMS=0 turns it off) take the luck out of it?  Each case is a
child process that first makes S extra streams (torch.cuda.Stream: what a host application would have made before the context
exists, which shifts every later placement), then times 200 batch-64 steps, 200 single commitments in a stream, 100 single
commitments each waited for.
    python tools/stream_order_probe.py 0 1 2 3 5          (S values; each with and without the choice)
Under rocprofv3 --kernel-trace the child's placement is in the trace: tools/queue_map.py."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(extra):
    import torch
    import bench
    import ligero_amd
    keep = [torch.cuda.Stream() for _ in range(extra)]
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    rows, k, batch = 344, 128, 64
    pre = bench.synthetic_preenc(1, batch * rows * k).reshape(-1, k, 4)
    out = {}
    cm = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch, device=0)
    cm.upload(pre)
    for _ in range(10):
        cm.commit_resident()
    cm.sync()
    t = time.perf_counter()
    for _ in range(200):
        cm.commit_resident()
    cm.sync()
    out["step_ms"] = round((time.perf_counter() - t) / 200 * 1e3, 4)
    cm.close()
    one = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=1, device=0)
    one.upload(pre[:rows])
    for _ in range(5):
        one.commit_resident()
    one.sync()
    t = time.perf_counter()
    for _ in range(200):
        one.commit_resident()
    one.sync()
    out["single_ms"] = round((time.perf_counter() - t) / 200 * 1e3, 4)
    t = time.perf_counter()
    for _ in range(100):
        one.commit_resident()
        one.root()
    out["single_latency_ms"] = round((time.perf_counter() - t) / 100 * 1e3, 4)
    one.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    else:
        for extra in sys.argv[1:]:
            for pick in ("0", "1"):
                env = dict(os.environ, LG_PICK_STREAMS=pick)
                r = subprocess.run([sys.executable, __file__, "--child", extra], env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                print(f"extra streams {extra:>2}  chosen streams {'on ' if pick == '1' else 'off'}  {line[-1] if line else r.stderr[-300:]}", flush=True)
