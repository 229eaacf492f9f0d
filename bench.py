#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PML / col-ID query path on MI355X.

Metric (BASELINE.json): query bases/s on a 64-haplotype-scale synthetic index
(~200 M rows) with 150 bp reads, 10 M reads per GPU, at 1/2/4/8 GPUs (weak
scaling: reads shard across ranks, the index is replicated in every HBM, the
.pml/.cid blocks are gathered to rank 0 over RCCL).

A "step" = one pass of the query over this rank's whole read batch, inputs and
outputs resident in HBM (for N > 1 it includes the RCCL gather, pipelined in
chunks behind the compute).  One JSON line on rank 0.

    python bench.py                       # N=1, C2 workload
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

`--rows/--reads/--read-len` shrink the workload for rehearsals (the line then
names the smaller workload; only the default is the BASELINE configuration).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
GATHER_CEILING_G = 39.5        # G random 32-byte rows (2 x 16 B loads, one line fill)/s this chip sustains on a 16 GiB table
GATHER_CEILING_LINE_G = 47.0   # G random whole 128-byte lines/s, fetched lane-cooperatively into LDS, on a 128 GiB table
ALG_BYTES_PER_BASE = 27


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=200_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-permille", type=int, default=10)
    ap.add_argument("--chunks", type=int, default=0, help="pipeline chunks per step (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--gather", choices=("packed", "raw"), default="packed",
                    help="N > 1: PML values travel to rank 0 as one bit per base (gather codec) or as u16")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    pkg = load_package()
    m = args.read_len
    n_reads = args.reads
    n_bases = n_reads * m

    # ---- the index: synthesised on the host in the reference's on-disk format,
    #      loaded through the C-ABI (upload + device re-layout)
    t0 = time.time()
    image = pkg.synth_index(args.rows, mean_len=8, split_permille=0, seed=42)
    t_gen = time.time() - t0
    # Rank 0 also holds every rank's rebuilt results (world x 3 bytes per base, + packed copies and the
    # check's slices) next to its own batch.  AUTO's deep mismatch entries leave ~80 GB of the 309 GB
    # next to the C2 index: beyond 40 GB of such buffers (N = 8 here) every rank opens with the 64-byte
    # entries instead (38 GB less; the same layout on all ranks, decided from world and batch size alone).
    if world > 1 and world * n_bases * 7 // 2 + 8 * n_bases > (40 << 30):
        os.environ.setdefault("COLBWT_LAYOUT", "5")
    t0 = time.time()
    tbl = pkg.ColPml.from_bytes(image, device=local_rank)
    t_load = time.time() - t0
    info = tbl.info()

    # ---- this rank's reads, sampled on the device by backward walk (seed differs per rank)
    d_bases = torch.zeros(n_bases + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    tbl.synth_reads_device(n_reads, m, args.sub_permille, 43 + rank, d_bases.data_ptr(), d_off.data_ptr(),
                           stream.cuda_stream)
    d_pml = torch.zeros(n_bases + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(n_bases + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    # ---- chunked query + gather pipeline (byte views; u16 PML travels as 2 bytes)
    from colbwt_amd import multi_gpu
    n_chunks = args.chunks or (1 if world == 1 else 4)
    pml_bytes = d_pml.view(torch.uint8)[:2 * n_bases]
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    codecs = []
    g_pml = None                                   # rank 0, packed gather: the rebuilt (world, bases) u16 values
    g_cid = None                                   # ... and, with the col-id dictionary, the rebuilt col ids
    cid_bits = 8
    if world > 1 and args.gather == "packed":
        words = (n_bases + 31) // 32
        d_zero = torch.zeros(words, dtype=torch.int32, device=dev)
        g_zero = torch.zeros((world, 4 * words), dtype=torch.uint8, device=dev) if rank == 0 else None
        if rank == 0:
            d_end = torch.zeros(words, dtype=torch.int32, device=dev)       # same read lengths on every rank
            pkg.read_end_mask_device(d_off.data_ptr(), n_reads, d_end.data_ptr(), stream.cuda_stream)
            g_pml = torch.zeros((world, words * 32), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()

        def pack(lo_base, nb):
            pkg.pml_pack_device(d_pml.data_ptr() + 2 * lo_base, nb, d_zero.data_ptr() + 4 * (lo_base // 32),
                                stream.cuda_stream)

        def unpack(r, w0, nw):
            pkg.pml_unpack_device(g_zero[r].data_ptr(), d_end.data_ptr(), w0, nw, words, g_pml[r].data_ptr(),
                                  comm_stream.cuda_stream)

        codecs.append(multi_gpu.PmlCodec(d_zero.view(torch.uint8), pack, unpack, g_zero))
        # the col ids as codes of the table's dictionary of ids (every rank holds the same table)
        ids = tbl.cid_dictionary()
        cid_bits = pkg.cid_code_bits(len(ids))
        if cid_bits <= 4:
            d_planes = torch.zeros(cid_bits * words, dtype=torch.int32, device=dev)
            g_planes = torch.zeros((world, 4 * cid_bits * words), dtype=torch.uint8, device=dev) if rank == 0 else None
            if rank == 0:
                g_cid = torch.zeros((world, words * 32), dtype=torch.uint8, device=dev)

            def pack_cid(lo_base, nb):
                pkg.cid_pack_device(d_cid.data_ptr() + lo_base, nb, ids, d_planes.data_ptr() + 4 * cid_bits * (lo_base // 32),
                                    stream.cuda_stream)

            def unpack_cid(r, w0, nw):
                pkg.cid_unpack_device(g_planes[r].data_ptr(), w0, nw, ids, g_cid[r].data_ptr(), comm_stream.cuda_stream)

            codecs.append(multi_gpu.PmlCodec(d_planes.view(torch.uint8), pack_cid, unpack_cid, g_planes, bits=cid_bits))
            outputs = []
        else:
            cid_bits = 8
            outputs = [(d_cid[:n_bases], 1)]
    else:
        outputs = [(pml_bytes, 2), (d_cid[:n_bases], 1)]
    pipe = multi_gpu.GatherPipeline(dist, rank, world, n_reads, m, n_chunks, outputs, dev, (stream, comm_stream),
                                    codecs=codecs)
    kernel_events = []

    def query_chunk(lo, hi):
        tbl.query_device(d_bases.data_ptr(), d_off.data_ptr() + 8 * lo, hi - lo, (hi - lo) * m,
                         d_pml.data_ptr(), d_cid.data_ptr(), 2, stream.cuda_stream)

    def timing_hook(when):      # HIP events on the stream the kernel is launched on
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        if when == "before":
            kernel_events.append([e, None])
        else:
            kernel_events[-1][1] = e

    def step(record):
        pipe.step(query_chunk, timing_hook if record else None)

    def fence():
        pipe.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # N > 1 (untimed): the gathered results against a plain u16 gather of the same outputs, an eighth of
    # the bases at a time -- rank 0 already holds world x 4.5 GB of rebuilt results next to its 229 GB index
    gather_ok = None
    if world > 1:
        gather_ok = True if rank == 0 else None
        n_slices = 8
        for sl in range(n_slices):
            lo, hi = n_bases * sl // n_slices, n_bases * (sl + 1) // n_slices
            raw = [torch.empty(hi - lo, dtype=torch.int16, device=dev) for _ in range(world)] if rank == 0 else None
            dist.gather(d_pml[lo:hi].contiguous(), raw, dst=0)
            rawc = [torch.empty(hi - lo, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            dist.gather(d_cid[lo:hi].contiguous(), rawc, dst=0)
            torch.cuda.synchronize()
            if rank == 0:
                got_c = g_cid if g_cid is not None else pipe.gathered[-1]
                got_p = g_pml if codecs else pipe.gathered[0].view(torch.int16)
                gather_ok = gather_ok and all(bool(torch.equal(got_c[r][lo:hi], rawc[r])) and bool(torch.equal(got_p[r][lo:hi], raw[r]))
                                              for r in range(world))
            del raw, rawc

    kernel_ms = [a.elapsed_time(b) for a, b in kernel_events]
    launches = len(kernel_ms)
    avg_launch_ms = sum(kernel_ms) / max(launches, 1)
    bases_per_launch = n_bases / n_chunks

    out = None
    if rank == 0:
        value = world * n_bases * args.steps / elapsed
        achieved = ALG_BYTES_PER_BASE * bases_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        traffic_source = None
        line_fills = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        is_baseline_cfg = (args.rows, args.reads, args.read_len) == (200_000_000, 10_000_000, 150)
        kernel_name = (f"fat2_query_kernel<{info.layout_shape >> 8},u16,{'deep' if info.layout == 6 else 'plain'}>" if info.layout in (5, 6)
                       else f"fat_query_kernel<{info.layout_shape >> 8},u16>" if info.layout == 4
                       else f"sk_query_kernel<{info.layout},u16>" if info.layout >= 2 else "pml_query_kernel<u16>")
        if is_baseline_cfg and n_chunks == 1 and os.path.exists(tpath):
            # PMC counters cannot be read inside this run: `traffic` is what the committed rocprofv3
            # passes measured for THIS kernel -- reported only while the kernel sources still hash to
            # what they were when the profile was taken, null otherwise (tools/profile_round.sh +
            # tools/summarize_profile.py <tag> --set-traffic refresh it).
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from summarize_profile import kernel_sources_sha
            tj = json.load(open(tpath))
            same_kernel = kernel_name.split("<")[0] in (tj.get("kernel") or "") and \
                kernel_name.split("<")[1].split(",")[0] in (tj.get("kernel") or "")
            fresh = tj.get("kernel_sources_sha") == kernel_sources_sha(ROOT)
            traffic_source = {"profile": f"profiles/{tj.get('tag')}_summary.json", "taken_at_commit": tj.get("taken_at_commit"),
                              "kernel": tj.get("kernel"), "kernel_sources_unchanged": fresh, "same_kernel": same_kernel}
            if fresh and same_kernel:
                traffic = tj.get("hbm_bytes_per_launch")
            if fresh and same_kernel and tj.get("read_requests_per_launch"):
                # What actually bounds the kernel: the chip's rate of random
                # 128-byte line fills, measured by tools/gather_littles.sh for this access shape.
                rd, wr = tj["read_requests_per_launch"], tj["write_requests_per_launch"]
                sec = avg_launch_ms * 1e-3
                ceiling = GATHER_CEILING_LINE_G if info.layout in (4, 5, 6) else GATHER_CEILING_G
                line_fills = {"read_requests_per_launch": rd, "write_requests_per_launch": wr,
                              "achieved_G_per_s": rd / sec / 1e9, "write_G_per_s": wr / sec / 1e9,
                              "ceiling_G_per_s": ceiling, "frac": rd / sec / 1e9 / ceiling,
                              "ceiling_source": ("profiles/r02_gather_line_rows.jsonl (dependent random whole lines, "
                                                 "lane-cooperative LDS-DMA, reads only)" if info.layout in (4, 5, 6) else
                                                 "profiles/r01_gather_littles_16GiB.jsonl (dependent random 2x16 B "
                                                 "loads per line, reads only)")}
        out = {
            "metric": "query bases/s", "value": value, "unit": "bases/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": f"{info.r} row synthetic .col_pml (n={info.n}), {n_reads}x{m} bp "
                                   f"backward-walk reads per GPU, {args.sub_permille / 10:.1f}% substitutions"
                                   + ("" if is_baseline_cfg else " [REDUCED rehearsal size]"),
                       "rows": int(info.r), "reads_per_gpu": n_reads, "read_len": m,
                       "parallelism": (f"reads sharded x{world}, index replicated, RCCL gather to rank 0 "
                                       + (f"(PML as 1 bit/base + col ids as {cid_bits} bits/base: {(1 + cid_bits) / 8:.3f} B/base)"
                                          if codecs else "(3 B/base)"))
                       if world > 1 else "single GPU",
                       "gather_matches_plain_gather": gather_ok,
                       "pipeline_chunks": n_chunks,
                       "index_gen_s": round(t_gen, 2), "index_load_s": round(t_load, 2),
                       "index_hbm_bytes": int(info.device_bytes),
                       "hbm_layout": {1: "one-step", 2: "two-step", 3: "three-step",
                                      4: f"line rows K={info.layout_shape >> 8} KS={info.layout_shape & 255}",
                                      5: f"line rows K={info.layout_shape >> 8} + mismatch lines",
                                      6: f"line rows K={info.layout_shape >> 8} + deep mismatch lines"}.get(info.layout, "?"),
                       "hbm_table_rows": int(info.table_rows)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name,
                         "avg_launch_ms": avg_launch_ms,
                         "launches": launches, "alg_bytes_per_base": ALG_BYTES_PER_BASE,
                         "bases_per_launch": bases_per_launch, "line_fills": line_fills},
        }

    # ---- CPU baseline: the oracle (a port, the reference itself cannot travel) on a
    #      bounded sample of rank 0's reads, 1 thread (the reference's threading model,
    #      pml_query.cpp:74) and all host cores; doubles as an on-box parity check.
    if rank == 0 and world == 1 and not args.no_cpu:
        oracle = load_oracle()
        ref = oracle.OracleIndex(image)
        cores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores

        def run_sample(k, threads):
            hb = d_bases[:k * m].cpu().numpy()
            ho = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
            t1 = time.perf_counter()
            p, c = ref.query_batch(hb, ho, threads=threads)
            return time.perf_counter() - t1, p, c

        probe_k = min(n_reads, 4000)
        dt, _, _ = run_sample(probe_k, 1)
        rate1 = probe_k * m / dt
        k1 = int(min(n_reads, max(probe_k, rate1 * args.cpu_seconds * 0.5 / m)))
        dt1, p1, c1 = run_sample(k1, 1)
        kall = int(min(n_reads, max(k1, (k1 * m / dt1) * cores * 0.5 * args.cpu_seconds * 0.5 / m)))
        dta, pa, ca = run_sample(kall, cores)
        gp = d_pml[:kall * m].cpu().numpy().view(np.uint16)
        gc = d_cid[:kall * m].cpu().numpy()
        parity = bool(np.array_equal(gp, pa) and np.array_equal(gc, ca)
                      and np.array_equal(gp[:k1 * m], p1) and np.array_equal(gc[:k1 * m], c1))
        resets = float((pa == 0).mean())
        out["cpu_baseline"] = {
            "value": k1 * m / dt1, "unit": "bases/s", "cores": 1, "kind": "port",
            "sample": f"oracle/colbwt_oracle.c on the first {k1} reads of the same batch ({k1 * m} bases, "
                      f"{dt1:.1f} s), same {info.r}-row index in host memory",
            "all_cores": {"value": kall * m / dta, "cores": cores, "reads": kall, "seconds": round(dta, 2)},
            "gpu_matches_oracle_on_sample": parity, "reset_fraction": resets,
        }
        if not parity:
            out["error"] = "GPU output differs from the oracle on the CPU sample"
        # ---- the kernel's real ceiling, measured on THIS box (it varies by a few percent from
        #      box to box): dependent random 2 x 16-byte loads per 128-byte line of a 16 GiB table
        #      (tools/gather_bench, a child process; skipped together with the CPU leg, e.g. under
        #      rocprofv3).  Replaces the constant taken from profiles/.
        lf = out["roofline"].get("line_fills")
        gb = os.path.join(ROOT, "tools", "gather_bench")
        if lf and os.path.exists(gb):
            try:
                import subprocess
                tbl.close()                      # the calibration table needs the HBM the index holds
                torch.cuda.empty_cache()
                gb_args = ["65536", "196608", "1500", "16", "2", "0"] if info.layout in (4, 5, 6) else ["16384", "524288", "1500", "10", "2", "0"]
                res = subprocess.run([gb] + gb_args, capture_output=True, text=True, timeout=120)
                rates = [json.loads(line)["Gsteps_per_s"] for line in res.stdout.splitlines() if line.startswith("{")]
                if rates:
                    lf["ceiling_G_per_s"] = max(rates)
                    lf["frac"] = lf["achieved_G_per_s"] / lf["ceiling_G_per_s"]
                    lf["ceiling_source"] = "tools/gather_bench on this box, after the timed region " + (
                        "(dependent random whole 128-byte lines, lane-cooperative LDS-DMA, 64 GiB table, reads only)"
                        if info.layout in (4, 5, 6) else
                        "(dependent random loads of one aligned 32-byte row = 2 x 16 B, 16 GiB table, reads only)")
            except Exception as e:      # the calibration is optional: keep the recorded constant
                lf["ceiling_note"] = f"live calibration failed: {e}"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if out is None or "error" not in out else 1


if __name__ == "__main__":
    sys.exit(main())
