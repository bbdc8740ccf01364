"""How the command line's host stage scales over worker PROCESSES x parser THREADS on this host (no GPU used): bench.py's
`_hostcap_worker` body (cli.process_files around an engine that computes nothing) with P processes x t parser threads each.
  python3 scripts/host_scaling.py [reads_total]"""
import multiprocessing as mp
import os
import resource
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def worker(rank, world, din, names, out_dir, threads, barrier, q):
    from nanoreviser_amd import cli
    args = cli.get_args(["-d", din + "/", "-o", out_dir + "/", "-S", "ecoli", "--thread", str(threads)])
    os.makedirs(args.output_dir, exist_ok=True)
    cpus = cli.worker_cpus(rank, world, [0] * world)
    if cpus:
        os.sched_setaffinity(0, cpus)
    os.environ["NRV_PARSER_THREADS_MAX"] = "64"
    cli.process_files(args, names[:8], bench.NullEngine(), lambda m: None, core_share=threads)
    barrier.wait(120)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    st = cli.process_files(args, names, bench.NullEngine(), lambda m: None, core_share=threads)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    q.put((rank, st["bases"], st["parser_workers"], (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), st["host_s"]))


if __name__ == "__main__":
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
    tmp = tempfile.mkdtemp(prefix="nrv_hostscale_")
    din, names = bench._hostcap_files(tmp, total)
    ctx = mp.get_context("spawn")
    print("cores:", bench.host_cores())
    for P, t in ((8, 2), (8, 3), (8, 4), (4, 4), (4, 8), (2, 8), (1, 16), (16, 1), (16, 2)):
        barrier, q = ctx.Barrier(P + 1), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, P, din, names[r::P], os.path.join(tmp, f"o{P}_{t}_{r}"), t, barrier, q)) for r in range(P)]
        for p in procs:
            p.start()
        barrier.wait(180)
        t0 = time.perf_counter()
        res = [q.get(timeout=600) for _ in procs]
        dt = time.perf_counter() - t0
        for p in procs:
            p.join(30)
        b = sum(r[1] for r in res); cpu = sum(r[3] for r in res); hs = sum(r[4] for r in res)
        print(f"{P:2d} processes x {t:2d} parser threads ({sorted({r[2] for r in res})} used): {b / dt / 1e6:6.1f} M bases/s, wall {dt:.2f} s, "
              f"CPU {cpu:.1f} s = {cpu / dt:.1f} cores busy, parser C time {hs:.1f} s ({hs / total * 1e3:.2f} ms per read)", flush=True)
        for r in range(P):
            shutil.rmtree(os.path.join(tmp, f"o{P}_{t}_{r}"), ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)
