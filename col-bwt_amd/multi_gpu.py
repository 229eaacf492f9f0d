"""Read sharding and the `.pml/.cid` gather for N GPUs of one node.

Reads are independent (the reference walks them one after the other with no
shared mutable state, src/pml_query.cpp:74-86), so the query shards trivially:
contiguous read ranges balanced by base count, the index replicated in every
GPU's HBM, no collective on the data path.  The only exchange is the gather of
the per-base outputs to rank 0 (RCCL over xGMI: `torch.distributed` backend
"nccl"), pipelined in chunks behind the compute so the links work while the
next chunk is being queried.  The 16-bit PML values travel as ONE BIT per base
(`PmlCodec`; include/colbwt.h "multi-GPU gather codec"): a read's values are
determined by where they are zero; the col ids travel as codes of the table's
own dictionary of ids (3 bits per base on the C2 index).  3 bytes per base
become 0.5 and rank 0 rebuilds both arrays on its own GPU -- rank 0's ingest
over its 7 xGMI links is what limits the job at N = 8, not the query.

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


def chunk_bounds(n_reads, n_chunks, align=1):
    """Read-index boundaries of the pipeline chunks of one rank's shard; inner
    boundaries are multiples of `align` reads."""
    n_chunks = max(1, min(n_chunks, max(n_reads, 1)))
    b = [min(n_reads, (n_reads * c // n_chunks) // align * align) for c in range(n_chunks)] + [n_reads]
    return [x for k, x in enumerate(b) if k == 0 or k == len(b) - 1 or x > b[k - 1]] if align > 1 else b


class PmlCodec:
    """A per-base array on its way to rank 0 as `bits` bits per base (fixed-length reads): 32 bases
    are `bits` 32-bit words, so every chunk that starts on a 32-base word has a byte range of its
    own in the packed array.  bits = 1: the PML values (a read's values are determined by where they
    are zero); bits = colbwt_cid_code_bits(#ids): the col ids as codes of the table's dictionary.

    The callables get plain integers (element / word indices) and run on the caller's current
    stream: the HIP kernels of csrc/gather_codec.hip on a GPU (bench.py), numpy stand-ins in the
    gloo test.
      pack(lo_base, n_bases)              local values -> local packed words
      unpack(r, first_word, n_words)      rank 0: gathered packed words of rank r -> its values
    `mask` is the local packed array as a uint8 tensor of whole 32-bit words (4 * bits bytes per 32
    bases); on rank 0 `gathered_mask` is (world, mask.numel()) and the rebuilt values live with the
    caller.
    """

    def __init__(self, mask, pack, unpack, gathered_mask=None, bits=1):
        self.mask, self.pack, self.unpack, self.gathered_mask, self.bits = mask, pack, unpack, gathered_mask, bits


class GatherPipeline:
    """Per-step driver: query chunk c, then gather it while the following chunks -- of
    this step and of the next -- compute; a chunk's buffers are only reused once its
    previous gather has finished.  Call finish() before reading the results.

    query_chunk(lo, hi) launches the query for reads [lo, hi) of this rank's
    shard on the current stream (asynchronously on a GPU).  `outputs` is a list
    of (local_1d_uint8_tensor, bytes_per_base).  Fixed-length reads only (the
    benchmark shape): chunk byte ranges are lo*m*bpb .. hi*m*bpb.
    On rank 0 `gathered[k]` is a (world, bytes) uint8 tensor per output.
    """

    def __init__(self, dist, rank, world, n_reads, read_len, n_chunks, outputs, device, streams=None,
                 pml_codec=None, codecs=None):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.m = read_len
        # arrays that travel packed (`codecs`; `pml_codec` = a list of one): a chunk must then start on
        # a 32-base word, which 32 reads of any length do
        self.codecs = list(codecs or []) + ([pml_codec] if pml_codec else [])
        self.bounds = chunk_bounds(n_reads, n_chunks, 32 if self.codecs else 1)
        self.outputs = outputs
        self.cuda = device.type == "cuda"
        self.compute_stream, self.comm_stream = streams if streams else (None, None)
        self.gathered = None
        self.pending = [[] for _ in range(len(self.bounds) - 1)]   # per chunk: gathers still reading it
        if world > 1 and rank == 0:
            self.gathered = [torch.empty((world, t.numel()), dtype=torch.uint8, device=device) for t, _ in outputs]

    def step(self, query_chunk, on_launch=None):
        torch, dist = self.torch, self.dist
        for c in range(len(self.bounds) - 1):
            lo, hi = self.bounds[c], self.bounds[c + 1]
            works = self.pending[c]
            for w in works:              # the compute stream waits for last step's gather of this chunk
                w.wait()
            del works[:]
            if on_launch:
                on_launch("before")
            query_chunk(lo, hi)
            if on_launch:
                on_launch("after")
            if self.world == 1:
                continue
            for codec in self.codecs:
                codec.pack(lo * self.m, (hi - lo) * self.m)
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
                w0, w1 = lo * self.m // 32, (hi * self.m + 31) // 32          # whole words of the chunk
                for codec in self.codecs:
                    a, b = 4 * codec.bits * w0, 4 * codec.bits * w1
                    gm = codec.gathered_mask
                    glist = [gm[r, a:b] for r in range(self.world)] if self.rank == 0 else None
                    work = dist.gather(codec.mask[a:b], glist, dst=0, async_op=True)
                    works.append(work)
                    if self.rank == 0:
                        work.wait()          # orders the comm stream behind the gather (blocks on gloo)
                        for r in range(self.world):
                            codec.unpack(r, w0, w1 - w0)

    def finish(self):
        """Waits for every outstanding gather (and, on a GPU, orders the compute stream behind the
        side stream's decode kernels)."""
        for works in self.pending:
            for w in works:
                w.wait()
            del works[:]
        if self.cuda and self.comm_stream is not None and self.world > 1:
            self.compute_stream.wait_stream(self.comm_stream)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
