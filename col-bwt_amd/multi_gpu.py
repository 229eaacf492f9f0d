"""Read sharding and the `.pml/.cid` gather for N GPUs of one node.

Reads are independent (the reference walks them one after the other with no
shared mutable state, src/pml_query.cpp:74-86), so the query shards trivially:
contiguous read ranges balanced by base count, the index replicated in every
GPU's HBM, no collective on the data path.  The only exchange is the gather of
the per-base outputs to rank 0 (RCCL over xGMI: `torch.distributed` backend
"nccl"), pipelined in chunks behind the compute so the links work while the
next chunk is being queried.

Everything here is backend-agnostic plumbing (also exercised with gloo on CPU
tensors by tests/test_multi_gpu_gloo.py); the compute is injected as a callable.
"""
import numpy as np


def shard_reads(read_off, world):
    """Contiguous read ranges [(lo, hi)] per rank, balanced by base count."""
    read_off = np.asarray(read_off, dtype=np.uint64)
    n_reads = read_off.size - 1
    total = int(read_off[-1] - read_off[0]) if n_reads > 0 else 0
    cuts = [0]
    for r in range(1, world):
        target = int(read_off[0]) + (total * r) // world
        k = int(np.searchsorted(read_off, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), n_reads))
    cuts.append(n_reads)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def chunk_bounds(n_reads, n_chunks):
    """Read-index boundaries of the pipeline chunks of one rank's shard."""
    n_chunks = max(1, min(n_chunks, max(n_reads, 1)))
    return [n_reads * c // n_chunks for c in range(n_chunks + 1)]


class GatherPipeline:
    """Per-step driver: query chunk c, then gather it while chunk c+1 computes.

    query_chunk(lo, hi) launches the query for reads [lo, hi) of this rank's
    shard on the current stream (asynchronously on a GPU).  `outputs` is a list
    of (local_1d_uint8_tensor, bytes_per_base).  Fixed-length reads only (the
    benchmark shape): chunk byte ranges are lo*m*bpb .. hi*m*bpb.
    On rank 0 `gathered[k]` is a (world, bytes) uint8 tensor per output.
    """

    def __init__(self, dist, rank, world, n_reads, read_len, n_chunks, outputs, device, streams=None):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.m = read_len
        self.bounds = chunk_bounds(n_reads, n_chunks)
        self.outputs = outputs
        self.cuda = device.type == "cuda"
        self.compute_stream, self.comm_stream = streams if streams else (None, None)
        self.gathered = None
        if world > 1 and rank == 0:
            self.gathered = [torch.empty((world, t.numel()), dtype=torch.uint8, device=device) for t, _ in outputs]

    def step(self, query_chunk, on_launch=None):
        torch, dist = self.torch, self.dist
        works = []
        for c in range(len(self.bounds) - 1):
            lo, hi = self.bounds[c], self.bounds[c + 1]
            if on_launch:
                on_launch("before")
            query_chunk(lo, hi)
            if on_launch:
                on_launch("after")
            if self.world == 1:
                continue
            if self.cuda:
                done = torch.cuda.Event()
                done.record(self.compute_stream)
                ctx = torch.cuda.stream(self.comm_stream)
            else:
                done, ctx = None, _NullCtx()
            with ctx:
                if done is not None:
                    self.comm_stream.wait_event(done)
                for k, (src, bpb) in enumerate(self.outputs):
                    a, b = bpb * lo * self.m, bpb * hi * self.m
                    glist = [self.gathered[k][r, a:b] for r in range(self.world)] if self.rank == 0 else None
                    works.append(dist.gather(src[a:b], glist, dst=0, async_op=True))
        for w in works:
            w.wait()
        if self.cuda and self.comm_stream is not None and self.world > 1:
            self.compute_stream.wait_stream(self.comm_stream)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
