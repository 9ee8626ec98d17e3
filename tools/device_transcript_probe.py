"""proofs/s of the throughput-mode prover with the transcript on the device, for several (provers in flight) x (proofs per
batch) splits on one GPU; `host` = the host-transcript batch prover for comparison.
    python tools/device_transcript_probe.py 2x256 pipe:1x1024 host:4x64 [--steps=6] [--threads=N] [--cpus=C]
PxB = P provers in flight (one thread each) of B proofs per batch; pipe:PxB = each prover keeps two batches in flight.
--cpus C pins the process to C CPUs first (what one rank of an 8-GPU node gets of this box's quota: C = 2).
--resident: the openings stay on the device (lg_prover_set_resident): what the device can prove when PCIe is not the bound."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    steps = int(opt.get("--steps", 6))
    if "--cpus" in opt:
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:int(opt["--cpus"])]))
    import bench
    from ligero_amd.prover import LigeroBatchProver
    inst, idx, vals = bench.poseidon_batch_inputs()
    ncpu = len(os.sched_getaffinity(0)) if "--cpus" in opt else bench.usable_cpus()
    print(f"usable cpus {ncpu}, steps {steps}", flush=True)
    for spec in args:
        host = spec.startswith("host:")
        pipe = spec.startswith("pipe:")      # one prover, two batches in flight: submit(i + 1) before collect(i)
        nprov, batch = (int(x) for x in spec.split(":")[-1].split("x"))
        threads = int(opt.get("--threads", max(1, ncpu // nprov)))
        allv = np.ascontiguousarray(vals[np.arange(batch) % 64])
        provers = [LigeroBatchProver(inst, batch, device=0, threads=threads, device_transcript=not host) for _ in range(nprov)]
        try:
            for bp in provers:
                bp.prove(idx, allv, copy=False)
                if "--resident" in sys.argv and not host:
                    bp.set_resident(True)
                    bp.prove(idx, allv, copy=False)

            def work(bp):
                if pipe:
                    bp.submit(idx, allv)
                    for _ in range(steps - 1):
                        bp.submit(idx, allv)
                        bp.collect()
                    bp.collect()
                    return
                for _ in range(steps):
                    bp.prove(idx, allv, copy=False)
            ts = [threading.Thread(target=work, args=(bp,)) for bp in provers]
            c0 = time.process_time()
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            cpu = time.process_time() - c0
        finally:
            for bp in provers:
                bp.close()
        n = nprov * batch * steps
        print(f"{spec:>12}  threads {threads:2d}  {n / dt:9.0f} proofs/s   {dt / steps * 1e3:8.1f} ms per round of {nprov} x {batch}   "
              f"host CPU {cpu / n * 1e3:6.3f} ms per proof", flush=True)


if __name__ == "__main__":
    main()
