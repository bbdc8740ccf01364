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
    # per-thread CPU (user, system) by thread name: a sampler keeps the last /proc/self/task/<tid>/stat of every thread it saw
    import threading
    seen, names_of, stop = {}, {}, threading.Event()
    tck = os.sysconf("SC_CLK_TCK")

    def sample():
        while not stop.is_set():
            for th in threading.enumerate():
                if th.native_id is not None:
                    names_of[th.native_id] = th.name.split("_")[0].rstrip("0123456789-") or th.name
            for tid in os.listdir("/proc/self/task"):
                try:
                    f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
                    seen[int(tid)] = (int(f[11]) / tck, int(f[12]) / tck)
                except Exception:
                    pass
            stop.wait(0.02)
    base = {}
    for tid in os.listdir("/proc/self/task"):
        f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
        base[int(tid)] = (int(f[11]) / tck, int(f[12]) / tck)
    smp = threading.Thread(target=sample, name="sampler", daemon=True)
    smp.start()
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    st = cli.process_files(args, names, bench.NullEngine(), lambda m: None, core_share=threads)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    stop.set()
    smp.join()
    by = {}
    for tid, (u, sy) in seen.items():
        u0, s0 = base.get(tid, (0.0, 0.0))
        nm = names_of.get(tid, "other")
        a = by.setdefault(nm, [0.0, 0.0])
        a[0] += u - u0
        a[1] += sy - s0
    q.put((rank, st["bases"], st["parser_workers"], (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), st["host_s"],
           r1.ru_stime - r0.ru_stime, by))


if __name__ == "__main__":
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
    tmp = tempfile.mkdtemp(prefix="nrv_hostscale_", dir=os.environ.get("HOST_SCALING_TMP"))
    odir = tempfile.mkdtemp(prefix="nrv_hostscale_out_", dir=os.environ.get("HOST_SCALING_OUT", os.environ.get("HOST_SCALING_TMP")))
    din, names = bench._hostcap_files(tmp, total)
    ctx = mp.get_context("spawn")
    print("cores:", bench.host_cores())
    for P, t in [tuple(int(x) for x in c.split("x")) for c in os.environ.get("HOST_SCALING", "8x2,8x3,8x4,4x4,4x8,2x8,1x16,16x1,16x2").split(",")]:
        barrier, q = ctx.Barrier(P + 1), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, P, din, names[r::P], os.path.join(odir, f"o{P}_{t}_{r}"), t, barrier, q)) for r in range(P)]
        for p in procs:
            p.start()
        barrier.wait(180)
        t0 = time.perf_counter()
        res = [q.get(timeout=600) for _ in procs]
        dt = time.perf_counter() - t0
        for p in procs:
            p.join(30)
        b = sum(r[1] for r in res); cpu = sum(r[3] for r in res); hs = sum(r[4] for r in res)
        agg = {}
        for r in res:
            for k, (u, sy) in r[6].items():
                a = agg.setdefault(k, [0.0, 0.0]); a[0] += u; a[1] += sy
        print(f"{P:2d} processes x {t:2d} parser threads ({sorted({r[2] for r in res})} used): {b / dt / 1e6:6.1f} M bases/s, wall {dt:.2f} s, "
              f"CPU {cpu:.1f} s = {cpu / dt:.1f} cores busy (system {sum(r[5] for r in res):.1f} s), parser C time {hs:.1f} s ({hs / total * 1e3:.2f} ms per read); "
              "by thread (user+sys s): " + ", ".join(f"{k} {u:.1f}+{sy:.1f}" for k, (u, sy) in sorted(agg.items(), key=lambda kv: -sum(kv[1]))), flush=True)
        for r in range(P):
            shutil.rmtree(os.path.join(odir, f"o{P}_{t}_{r}"), ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(odir, ignore_errors=True)
