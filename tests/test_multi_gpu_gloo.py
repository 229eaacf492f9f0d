"""N > 1 path on CPU: world_size-2 gloo processes run the same sharding and
chunked gather pipeline bench.py uses on RCCL (col-bwt_amd/multi_gpu.py); the
per-rank compute is the oracle here (the checker standing in for the GPU
kernel, which cannot run in this tier), so the test pins the plumbing: shard
boundaries, chunk byte ranges, gather order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _np_unpack(zero_bits, end_bits):
    """Reference of csrc/gather_codec.hip: pml[k] = j - k + (zero[j] ? 0 : 1), j = next stop >= k."""
    n = len(zero_bits)
    out = np.zeros(n, np.uint16)
    j, add = -1, 0
    for k in range(n - 1, -1, -1):
        if zero_bits[k] or end_bits[k]:
            j, add = k, 0 if zero_bits[k] else 1
        out[k] = j - k + add if j >= 0 else 0
    return out


def _worker(rank, world, port, n_reads, m, n_chunks, q, packed=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_oracle, load_package
    pkg = load_package()
    from colbwt_amd import multi_gpu
    import helpers
    oracle = load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    image = pkg.synth_index(3000, mean_len=6, split_permille=100, seed=5).tobytes()
    reads = helpers.backward_walk_reads(image, n_reads * world, m, 0.02, seed=9)     # the whole job
    mine = reads[rank * n_reads:(rank + 1) * n_reads]                                 # this rank's shard
    bases, off = helpers.concat_reads(mine)
    ref = oracle.OracleIndex(image)
    pml = torch.zeros(n_reads * m, dtype=torch.int16)
    cid = torch.zeros(n_reads * m, dtype=torch.uint8)

    def query_chunk(lo, hi):           # stand-in for ColPml.query_device on reads [lo, hi)
        p, c = ref.query_batch(bases[int(off[lo]):int(off[hi])], off[lo:hi + 1] - off[lo])
        pml[lo * m:hi * m] = torch.from_numpy(p.view(np.int16))
        cid[lo * m:hi * m] = torch.from_numpy(c)

    if not packed:
        pipe = multi_gpu.GatherPipeline(dist, rank, world, n_reads, m, n_chunks,
                                        [(pml.view(torch.uint8), 2), (cid, 1)], torch.device("cpu"))
        pipe.step(query_chunk)
        pipe.step(query_chunk)      # buffers are reused across steps
        pipe.finish()
        gp = pipe.gathered[0].reshape(-1).numpy().view(np.uint16) if rank == 0 else None
        gc = pipe.gathered[1].reshape(-1).numpy() if rank == 0 else None
    else:
        # the bit-per-base PML gather, numpy standing in for the codec kernels
        nb = n_reads * m
        words = (nb + 31) // 32
        mask = torch.zeros(4 * words, dtype=torch.uint8)
        gmask = torch.zeros((world, 4 * words), dtype=torch.uint8) if rank == 0 else None
        gpml = np.zeros((world, words * 32), np.uint16)
        end_bits = np.zeros(words * 32, bool)
        end_bits[np.arange(1, n_reads + 1) * m - 1] = True

        def pack(lo_base, n):
            assert lo_base % 32 == 0
            z = (pml[lo_base:lo_base + n].numpy() == 0)
            z = np.concatenate((z, np.zeros((-n) % 32, bool)))
            mask[lo_base // 8:lo_base // 8 + len(z) // 8] = torch.from_numpy(np.packbits(z, bitorder="little"))

        def unpack(r, w0, nw):
            zb = np.unpackbits(gmask[r].numpy(), bitorder="little").astype(bool)
            assert end_bits[min(32 * (w0 + nw), nb) - 1]          # a chunk ends at a read end
            gpml[r, 32 * w0:32 * (w0 + nw)] = _np_unpack(zb[:32 * (w0 + nw)], end_bits[:32 * (w0 + nw)])[32 * w0:]

        codec = multi_gpu.PmlCodec(mask, pack, unpack, gmask)
        # ... and the col ids as codes of the table's dictionary, `bits` bit planes per 32 bases
        # (numpy stand-ins for cid_pack_kernel / cid_unpack_kernel of csrc/gather_codec.hip)
        ids = np.unique(helpers.unpack_col_pml(image)["cid"])
        bits = max(1, int(np.ceil(np.log2(len(ids)))))
        code_of = np.zeros(256, np.uint8)
        code_of[ids] = np.arange(len(ids))
        planes = torch.zeros(4 * bits * words, dtype=torch.uint8)
        gplanes = torch.zeros((world, 4 * bits * words), dtype=torch.uint8) if rank == 0 else None
        gcid = np.zeros((world, words * 32), np.uint8)

        def pack_cid(lo_base, n):
            assert lo_base % 32 == 0
            c = code_of[cid[lo_base:lo_base + n].numpy()]
            c = np.concatenate((c, np.zeros((-n) % 32, np.uint8))).reshape(-1, 32)
            pl = np.stack([np.packbits((c >> p) & 1, axis=1, bitorder="little") for p in range(bits)], axis=1)   # (words, bits, 4)
            w0 = lo_base // 32
            planes[4 * bits * w0:4 * bits * (w0 + len(c))] = torch.from_numpy(pl.reshape(-1))

        def unpack_cid(r, w0, nw):
            pl = gplanes[r].numpy()[4 * bits * w0:4 * bits * (w0 + nw)].reshape(nw, bits, 4)
            codes = sum(np.unpackbits(pl[:, p], axis=1, bitorder="little").astype(np.uint8) << p for p in range(bits))
            gcid[r, 32 * w0:32 * (w0 + nw)] = ids[np.minimum(codes, len(ids) - 1)].reshape(-1)

        cid_codec = multi_gpu.PmlCodec(planes, pack_cid, unpack_cid, gplanes, bits=bits)
        pipe = multi_gpu.GatherPipeline(dist, rank, world, n_reads, m, n_chunks, [], torch.device("cpu"),
                                        codecs=[codec, cid_codec])
        pipe.step(query_chunk)
        pipe.step(query_chunk)      # buffers are reused across steps
        pipe.finish()
        gp = gpml[:, :nb].reshape(-1) if rank == 0 else None
        gc = gcid[:, :nb].reshape(-1) if rank == 0 else None
        assert bits < 8             # the synthetic table's ids: a handful of values
    if rank == 0:
        all_bases, all_off = helpers.concat_reads(reads)
        ep, ec = ref.query_batch(all_bases, all_off)
        q.put(bool(np.array_equal(gp, ep) and np.array_equal(gc, ec)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks,n_reads,packed", [(1, 40, False), (3, 40, False), (1, 40, True), (3, 100, True)])
def test_two_rank_gather_matches_single_rank(n_chunks, n_reads, packed):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_reads, 50, n_chunks, q, packed)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_shard_reads_balances_bases():
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from colbwt_amd import multi_gpu
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 500, size=1000)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    for world in (1, 2, 3, 8):
        sh = multi_gpu.shard_reads(off, world)
        assert sh[0][0] == 0 and sh[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        per = [int(off[hi] - off[lo]) for lo, hi in sh]
        assert max(per) - min(per) <= 2 * 500
    assert multi_gpu.chunk_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert multi_gpu.chunk_bounds(0, 4) == [0, 0]
    assert multi_gpu.chunk_bounds(100, 4, 32) == [0, 32, 64, 100] and multi_gpu.chunk_bounds(40, 3, 32) == [0, 40]
