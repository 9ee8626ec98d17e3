"""Test helper: `world` ranks as THREADS of one process, each with an object that quacks like the part of torch.distributed
the sharded paths use (get_world_size / get_rank / get_backend / all_gather_into_tensor / send / recv / broadcast).  The GPU box allows at most six
processes on its card at once, so the world-8 layouts of the driver's scaling run are exercised here with eight sharded
contexts of the real device backend on eight threads of the test process (the library's contexts are independent and its
entry points release the GIL under ctypes); worlds of two and four also run as real gloo process groups elsewhere."""
import queue
import threading

import torch


class _Shared:
    def __init__(self, world, timeout):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world
        self.timeout = timeout
        self.mail = {(a, b): queue.Queue() for a in range(world) for b in range(world)}   # point to point: one FIFO per (src, dst)


class ThreadRankDist:
    """one rank's view of the group"""

    def __init__(self, shared, rank):
        self._s, self._rank = shared, rank

    def get_world_size(self, group=None):
        return self._s.world

    def get_rank(self, group=None):
        return self._rank

    def get_backend(self, group=None):
        return "gloo"                      # host tensors stay on the host, device tensors on the device

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        self._all_gather(out, inp)
        if async_op:
            class _Done:                   # the exchange above is synchronous: nothing left to wait for
                @staticmethod
                def wait():
                    return True
            return _Done()
        return None

    # point to point and broadcast (the row-relay commit): a message is a clone of the tensor, taken once the sender's device
    # work is complete -- the receiver copies it in and waits for its own device
    def send(self, tensor, dst, group=None):
        if tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)
        self._s.mail[(self._rank, dst)].put(tensor.detach().clone())

    def recv(self, tensor, src, group=None):
        msg = self._s.mail[(src, self._rank)].get(timeout=self._s.timeout)
        assert msg.numel() == tensor.numel()
        tensor.view(-1).copy_(msg.view(-1))
        if tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)

    def broadcast(self, tensor, src, group=None):
        s = self._s
        if self._rank == src:
            if tensor.is_cuda:
                torch.cuda.synchronize(tensor.device)
            s.slots[src] = tensor.detach().clone()
        s.barrier.wait()
        if self._rank != src:
            tensor.view(-1).copy_(s.slots[src].view(-1))
            if tensor.is_cuda:
                torch.cuda.synchronize(tensor.device)
        s.barrier.wait()

    def _all_gather(self, out, inp):
        s = self._s
        n = inp.numel()
        assert out.numel() == s.world * n
        s.slots[self._rank] = inp.reshape(-1).clone()       # `inp` may alias a block of `out`
        if inp.is_cuda:
            torch.cuda.synchronize(inp.device)
        s.barrier.wait()
        flat = out.view(-1)
        for r in range(s.world):
            flat[r * n:(r + 1) * n].copy_(s.slots[r])
        if out.is_cuda:
            torch.cuda.synchronize(out.device)
        s.barrier.wait()                                     # nobody overwrites a slot another rank still reads


def run_ranks(world, fn, timeout=300):
    """fn(rank, dist) on `world` threads; returns [fn's result per rank]; the first exception of any rank is re-raised
    (the barrier is broken so the other ranks leave their collective instead of waiting for it)."""
    shared = _Shared(world, timeout)
    out, errs = [None] * world, [None] * world

    def body(rank):
        try:
            out[rank] = fn(rank, ThreadRankDist(shared, rank))
        except BaseException as e:           # noqa: BLE001 -- reported to the caller below
            errs[rank] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout + 60)
    assert not any(t.is_alive() for t in threads), "a rank thread is stuck"
    real = [e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real or any(errs):
        raise (real or [e for e in errs if e is not None])[0]
    return out
