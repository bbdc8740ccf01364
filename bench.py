#!/usr/bin/env python3
"""bench.py - bases revised / s of the MI355X reviser engine on BASELINE.json's workload.

A "step" is one pass of the hot path (signal CNN + 4 Bi-LSTM + head, model1 AND model2) over one
batch of 4096 synthetic independent 13-event windows per GPU (north_star target; generator:
SURVEY.md 8d C4 = nanoreviser_amd/workload.py; E. coli weights + the seeded synthetic (78,16)
`feature` kernel, because the shipped files are T=11 - SURVEY.md F3).  One window == one revised base.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE from the environment), and typed as is - then this process, which
never touches a GPU, starts the N ranks itself as fresh child processes (the reference fans out from
one command the same way, NanoReviser.py:203-219), relays rank 0's JSON line and returns the first
non-zero exit code.

Inputs are resident in HBM before the timed region.  Reads/windows are independent, so ranks own
disjoint shards and the data path has NO collective (weak scaling); the only communication is the
barrier and three scalar reductions of the control plane, on `gloo` with CPU tensors - no RCCL anywhere
(a workload without collectives must not depend on RCCL initialising).  Rank 0 prints ONE JSON line.
`--share-device` maps rank r to device r % device_count, so that a 1-GPU box can rehearse the N-rank
path with the real engine (tests/test_gpu_multirank.py); it is reported in `config` and never the default.

Besides the contract keys the line carries (N = 1, rank 0; --no-extras turns them off):
  roofline            dominant kernel (lstm3), hipEvent pairs on the launch stream INSIDE the timed
                      region on every 8th launch (a bracket costs ~12 us of idle pipe, so bracketing
                      every launch would perturb a 20-step run by 2 %)
  kernel_us           every kernel, from a separate untimed pass after the timed region
  roofline_f32        the same step with plain f32 matrix instructions (IEEE f32, the reference's own
                      precision; ceiling 157.3 TFLOP/s), measured with the SAME protocol as `value`:
                      priming, warm-up, K timed steps, every-8th bracket, untimed all-kernel pass
  host_inclusive      nrv_predict / nrv_predict_read / nrv_predict_reads_raw from HOST memory
                      (H2D + D2H inside the timed call) - never `value`; cli_e2e: the command line itself,
                      fast5 files in, FASTA files out, in a child process (NanoReviser.py:105-183).
                      With --gpus N > 1 the line carries host_inclusive.cli_e2e with n_gpus = N
                      (`NanoReviser.py --gpus N` on N times the files, run while the ranks wait)
  read_mode           config C5: one 200 k-event read, human weights, device-formed windows
  configs             C2 (ecoli, batch 512) and C3 (human, batch 4096) on the replicated fixture reads
  cpu_baseline        oracle/nrv_oracle.c (plain-C f32 port of the reference graph - test
                      infrastructure) on the host cores: all cores, one thread, affinity, and the
                      host stage's us/base.  A reported baseline, never the thing measured as `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md:42 (dense f32 matrix)
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # same table: dense bf16 / f16 matrix
# f32-grade products on the 16-bit matrix pipe (include/nanorev.h, nrv_set_precision):
#   bf16x3: f32 operand = 3 bf16 terms, product = 6 MFMA products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid)
#   f16x2 : scaled f32 operand = 2 f16 terms, product = 3 MFMA products (hi*hi, hi*lo, lo*hi)
# The ceiling for ALGORITHMIC FLOPs in such a mode is the dense 16-bit peak / products.
PRODUCTS = {"bf16x3": 6, "f16x2": 3, "f32": 1}
DTYPE = {"f32": "f32", "bf16x3": "f32 via 3xbf16 split (f32 accumulate)",
         "f16x2": "f32 via scaled 2xf16 split (f32 accumulate)"}

# MAC per (window, timestep, model): SURVEY.md 8a
MAC_CNN = 1200 + 9600 + 25600
MAC_L = {1: 2816, 2: 49152, 3: 327680, 4: 163840}
MAC_HEAD_T = 16384 + 4096 + 192
FLOP_PER_BASE_DEDUP = {11: 24973216, 13: 29487264}     # SURVEY.md 8d, read mode (per-event CNN)


def flop_per_window(T):
    """2 x (model1 + model2) MACs of one independent window (31 234 464 at T=13)."""
    per_t = MAC_CNN + sum(MAC_L.values()) + MAC_HEAD_T
    return 2 * ((per_t * T + 96 * T + 96) + (per_t * T + 96 * T + 80))


def flop_lstm3_launch(T, n_windows, executed=True):
    """FLOPs of ONE lstm3 launch (both directions, both models).  `executed`: the h_0 = 0 recurrent
    product of the first step is not computed by the kernel, so it is not counted either."""
    D, H = 192, 128
    steps_rec = T - 1 if executed else T
    mac = 2 * 4 * H * (T * D + steps_rec * H)           # two directions
    return 2 * mac * 2 * n_windows                       # 2 FLOP/MAC, two models


def mode_peak(precision):
    if precision == "f32":
        return PEAK_F32_MFMA_TFLOPS, "dense f32 MFMA peak"
    k = PRODUCTS[precision]
    return (PEAK_16BIT_MFMA_TFLOPS / k,
            f"dense 16-bit MFMA peak {PEAK_16BIT_MFMA_TFLOPS:.0f} TFLOP/s / {k} MFMA products per f32-grade "
            "product; achieved counts ALGORITHMIC flops")


# ------------------------------------------------------------------------------------------------
# ranks
# ------------------------------------------------------------------------------------------------
class Dist:
    """One process per GPU.  The control plane (one barrier pair, three scalar reductions) runs on gloo with
    CPU tensors in every configuration: the data path has no collective, so nothing here touches RCCL."""

    def __init__(self, backend="gloo"):
        import torch.distributed as dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = dist if self.world > 1 else None
        self.backend = "gloo"
        if self.dist is not None and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)

    def _reduce(self, v, dtype, op):
        import torch
        t = torch.tensor([v], dtype=dtype)
        self.dist.all_reduce(t, op=op)
        return t.item()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_float(self, v):
        import torch
        return float(v) if self.dist is None else float(self._reduce(v, torch.float64, self.dist.ReduceOp.MAX))

    def min_float(self, v):
        import torch
        return float(v) if self.dist is None else float(self._reduce(v, torch.float64, self.dist.ReduceOp.MIN))

    def sum_int(self, v):
        import torch
        return int(v) if self.dist is None else int(self._reduce(v, torch.int64, self.dist.ReduceOp.SUM))

    def gather_obj(self, obj):
        """Every rank's small Python object, in rank order (gloo all_gather_object; [obj] for one rank)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def timed_steps(d, step_fn, sync_fn, steps, warmup):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides;
    returns (max-over-ranks elapsed seconds, this rank's elapsed seconds)."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    d.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    mine = time.perf_counter() - t0
    d.barrier()
    return d.max_float(mine), mine


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` typed as is: start the N ranks as FRESH child processes (this
    parent has made no HIP call and never will), one per GPU, rendezvous on 127.0.0.1.  Rank 0's
    stdout is ours (its JSON line goes straight through); the first failing rank ends the run."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NRV_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in live:                      # one rank died: the others would wait in the barrier forever
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


class DryRunEngine:
    """Stand-in for the engine in `--dry-run` (CPU, gloo): exercises launch, sharding, barriers,
    the max-over-ranks timing and the JSON line; it revises nothing and its rate means nothing."""

    def __init__(self, batch, T, seed):
        self.x = np.random.default_rng(seed).standard_normal((batch, T * 6)).astype(np.float32)
        self.n = 0

    def step(self):
        self.n += int(np.tanh(self.x).argmax(-1).shape[0])


def run_distributed_cpu_selftest(batch=4096, steps=3):
    """CPU / gloo exercise of the N>1 control path inside an already-spawned rank (tests/
    test_shard_gloo.py): the Dist class, timed_steps and the shard arithmetic of the real run over
    the dry-run engine's step loop."""
    from nanoreviser_amd import shard
    d = Dist("gloo")
    lo, hi = shard.shard_range(d.world * batch, d.rank, d.world)
    eng = DryRunEngine(hi - lo, 13, 20260 + d.rank)
    mx, mine = timed_steps(d, eng.step, lambda: None, steps, 1)
    total = d.sum_int(eng.n - (hi - lo))               # the warm-up step is not counted
    d.close()
    return {"total_units": total, "max_ms": mx * 1e3, "my_ms": mine * 1e3, "shard": (lo, hi)}


# ------------------------------------------------------------------------------------------------
# CPU baseline (oracle/ is the checker and the reported baseline, never the product)
# ------------------------------------------------------------------------------------------------
def host_cores():
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    cores = len(aff)
    quota = None
    try:                                   # a cgroup CPU quota caps what the threads can really use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
            cores = max(1, min(cores, int(quota + 0.5)))
    except Exception:
        pass
    return cores, aff, quota


def _mask_str(aff):
    """Affinity list as ranges: [0,1,2,3,8,9] -> '0-3,8-9'."""
    out, i = [], 0
    while i < len(aff):
        j = i
        while j + 1 < len(aff) and aff[j + 1] == aff[j] + 1:
            j += 1
        out.append(str(aff[i]) if i == j else f"{aff[i]}-{aff[j]}")
        i = j + 1
    return ",".join(out)


def host_stage_us_per_base():
    """The product's own host stage (a14/a15 counterparts: event collapse, medians, per-event stats,
    feature rows; windows are cut on the device) on one fixture read, one core."""
    from nanoreviser_amd import hoststage as hs
    p = os.path.join(ROOT, "tests", "golden", "reads", "ch13_read2251.npz")
    if not os.path.exists(p):
        return None
    g = np.load(p)
    cols = [g[k] for k in ("ev_start", "ev_mean", "ev_stdv", "ev_model_state", "ev_move", "raw_signal")]
    best, n = None, 0
    for _ in range(5):
        t0 = time.perf_counter()
        rd = hs.collapse_events(*cols)
        rt = hs.read_tensors_raw(rd)
        dt = time.perf_counter() - t0
        n = len(rt.feat_ev)
        best = dt if best is None else min(best, dt)
    return {"value": best / n * 1e6, "unit": "us/base", "cores": 1,
            "what": "nanoreviser_amd.hoststage collapse_events + read_tensors_raw on fixture read ch13_read2251 "
                    f"({n} bases), best of 5; windows are cut on the device (segment_kernel)",
            "reference_python_loop_us_per_base": 42.0,
            "reference_note": "preprocessing.py:85-170 signal_segmentation measured at survey time in the build "
                              "container (SURVEY.md 6); /root/reference does not exist on the GPU box"}


def cpu_baseline(m1, m2, T, sig, rd, target_s=12.0, host_cap=None):
    """oracle/nrv_oracle.c on all host cores (bounded to ~target_s) and on one thread (~target_s/2)."""
    from oracle import c_oracle as CO
    cores, aff, quota = host_cores()
    f1, f2 = m1.flat(), m2.flat()

    def run(n, threads):
        t0 = time.perf_counter()
        CO.predict(f1, T, 6, sig[:n], rd[:n], threads=threads)
        CO.predict(f2, T, 5, sig[:n], rd[:n], threads=threads)
        return time.perf_counter() - t0

    def bounded(threads, budget):
        n0 = min(len(rd), 8 * threads)
        t_cal = run(n0, threads)
        n = int(min(len(rd), max(n0, n0 * budget / max(t_cal, 1e-3))))
        t = run(n, threads)
        if t < 0.5 * budget and n == len(rd):            # whole step was too quick: repeat it
            reps = int(min(64, max(1, budget / max(t, 1e-3))))
            t0 = time.perf_counter()
            for _ in range(reps):
                run(n, threads)
            t = time.perf_counter() - t0
            n *= reps
        return n, t

    n, t = bounded(cores, target_s)
    n1, t1 = bounded(1, target_s / 2)
    blas = blas_grade_baseline(m1, m2, T, sig, rd, cores, target_s / 2)
    ref = "unavailable on this host"
    try:
        import keras  # noqa: F401
        import tensorflow  # noqa: F401
        ref = "importable (not timed: bench uses the port)"
    except Exception:
        pass
    return {"value": n / t, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"{n} windows drawn from the step's {len(rd)} synthetic windows (T={T}), model1+model2, "
                      f"oracle/nrv_oracle.c with {cores} OpenMP threads, {t:.1f} s",
            "one_thread": {"value": n1 / t1, "unit": "bases/s", "cores": 1,
                           "sample": f"{n1} windows, 1 thread, {t1:.1f} s"},
            "affinity": {"mask": _mask_str(aff), "n": len(aff), "os_cpu_count": os.cpu_count(),
                         "cgroup_cpu_quota": quota},
            "blas_grade": blas,
            "host_capacity": host_cap if host_cap is not None else {"error": "not measured (no helper process)"},
            "host_stage_us_per_base": host_stage_us_per_base(),
            "reference_keras_tf": ref,
            "note": "kind 'port': the reference's path is Python on keras 2.2.4 / tensorflow 1.12, which cannot be "
                    "installed here; the C port restates the same graph.  Reported baseline only - a GPU/CPU ratio "
                    "says nothing about kernel quality, the roofline fraction does."}


# The Bi-LSTM kernel of each mode as rocprofv3 names it, and the sources it is built from: `roofline.traffic` comes
# from a COMMITTED PMC pass (profiles/*_pmc_lstm3.json), so it is only printed while that pass still describes the
# kernel this library runs - same kernel name, same source text (sha256) - and is null, with the reason, otherwise.
KERNEL_SIGNATURE = {"f16x2": "lstm_h2w_kernel<32, 16, 128, 0, 4>",
                    "bf16x3": "lstm_split_kernel<32, 16, 128, 2, 1, 0>",
                    "f32": "lstm_layer_kernel<32, 16, 128, 1, 1, false, 0>"}
KERNEL_SOURCES = {"f16x2": ["nrv_lstm_f16x2w.h", "nrv_lstm_f16x2s.h", "nrv_lstm_f16x2.h"], "bf16x3": ["nrv_lstm_bf16x3.h"],
                  "f32": ["nrv_lstm_f32.h"]}


def kernel_source_sha(precision):
    """sha256 of the kernel's sources as the compiler sees them: `//` comments and blank lines do not count (a comment
    edit must not orphan a PMC pass; none of these files has `//` inside a string or a block comment)."""
    import hashlib
    import re
    hsh = hashlib.sha256()
    for f in KERNEL_SOURCES[precision]:
        with open(os.path.join(ROOT, "nanoreviser_amd", "csrc", f), "r") as fp:
            for line in fp:
                code = re.sub(r"\s*//.*$", "", line.rstrip("\n")).rstrip()
                if code:
                    hsh.update(code.encode() + b"\n")
    return hsh.hexdigest()[:16]


def step_source_sha():
    """sha256 of every kernel source of the step (csrc/nrv_*.h: the kernels and the switches that choose between them; comments
    and blank lines excluded): the step-wide PMC figures (`traffic_step`) describe the library only while none of them has
    changed.  nrv_api.hip (host code: entry points, packing, pipeline) is left out on purpose."""
    import glob
    import hashlib
    import re
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "nanoreviser_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "nrv_*.h"))):
        with open(f, "r") as fp:
            for line in fp:
                code = re.sub(r"\s*//.*$", "", line.rstrip("\n")).rstrip()
                if code:
                    hsh.update(code.encode() + b"\n")
    return hsh.hexdigest()[:16]


class NullEngine:
    """The engine's call surface with no device behind it: every window comes back as class 0 / 0.  What remains of a
    `cli.process_files` run is the HOST side of the command line - fast5 parsing, event collapse, statistics, packing,
    merge, record, file write - i.e. the rate at which this host can feed GPUs."""
    T = 11

    @staticmethod
    def pack_bundle(raw, starts, feat, meta, T):
        from nanoreviser_amd.engine import Reviser
        return Reviser.pack_bundle(raw, starts, feat, meta, T)

    def run_packed_raw(self, packed):
        out = packed[-1]
        for o in out:
            o.fill(0)
        return out

    def predict_reads_raw(self, raws, starts, feats, shifts, scales):
        n = max(sum(len(f) for f in feats) - self.T, 0)
        return (np.zeros((n, 6), np.float32), np.zeros((n, 5), np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))

    def predict_read(self, sig_ev, feat_ev):
        return self.predict_reads_raw(None, None, [feat_ev], None, None)

    def close(self):
        pass


def scratch_base():
    """Where the file-based legs (host_capacity, cli_e2e) keep their inputs and OUTPUTS: a tmpfs (/dev/shm) when there is one with
    room, else the default temporary directory.  Why: on the GPU boxes /tmp is the container's OVERLAY root, whose write path
    costs ten times the system time of any real filesystem once a directory has seen a few thousand creates + renames (r05,
    scripts/host_scaling.py: the same 8-worker host stage 68-78 M bases/s writing to the overlay, 120-131 M writing to tmpfs,
    parser time identical) - a property of the sandbox, not of the host stage.  Returns (dir or None, description)."""
    import shutil
    import tempfile
    if os.environ.get("NRV_BENCH_SCRATCH"):
        return os.environ["NRV_BENCH_SCRATCH"], "NRV_BENCH_SCRATCH"
    try:
        if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > (4 << 30):
            return "/dev/shm", "tmpfs (/dev/shm)"
    except OSError:
        pass
    return None, "default temporary directory (%s)" % tempfile.gettempdir()


def _hostcap_files(tmp, n_reads, copies=None):
    """n_reads hard links of COPIES of the committed fixture fast5 files -> (dir, names).  `copies` distinct inodes per fixture
    (default 32, NRV_HOSTCAP_COPIES): sixteen threads on two sockets reading the same two inodes contend for those inodes' page
    cache in the kernel - a property of this synthetic file set, not of a real run (every read its own file)."""
    import shutil
    copies = int(os.environ.get("NRV_HOSTCAP_COPIES", "32")) if copies is None else copies
    gold = os.path.join(ROOT, "tests", "golden", "fast5")
    src = sorted(f for f in os.listdir(gold) if f.endswith(".fast5"))
    if not src:
        raise RuntimeError("no fixture fast5 files")
    din, dcp = os.path.join(tmp, "in"), os.path.join(tmp, "copies")
    os.makedirs(din, exist_ok=True)
    os.makedirs(dcp, exist_ok=True)
    names = []
    for i in range(n_reads):
        fn = f"r{i:06d}_{i % len(src)}.fast5"
        dst = os.path.join(din, fn)
        if not os.path.exists(dst):
            j = (i // len(src)) % max(1, copies)
            cp = os.path.join(dcp, f"c{j}_{src[i % len(src)]}") if copies > 0 else os.path.join(gold, src[i % len(src)])
            if copies > 0 and not os.path.exists(cp):
                shutil.copy(os.path.join(gold, src[i % len(src)]), cp)
            try:
                os.link(cp, dst)
            except OSError:
                shutil.copy(cp, dst)
        names.append(fn)
    return din, names


def _hostcap_worker(rank, world, din, names, out_dir, threads, barrier, q):
    """One simulated GPU worker of `host_capacity`'s N-worker point: the command line's own worker body
    (cli.process_files with gpu_workers = world) around an engine that computes nothing."""
    try:
        from nanoreviser_amd import cli
        args = cli.get_args(["-d", din + "/", "-o", out_dir + "/", "-S", "ecoli", "--thread", str(threads)])
        os.makedirs(args.output_dir, exist_ok=True)
        # worker r next to GPU r, as `NanoReviser.py --gpus 8` places it; on a box with fewer GPUs worker_cpus finds no NUMA
        # node for the missing ones and cuts the allowed cores into `world` contiguous slices (all eight on GPU 0's
        # node was a placement the command line never uses)
        cpus = cli.worker_cpus(rank, world, list(range(world)))
        share = None
        if cpus:
            cores_all = cli.usable_cores()
            os.sched_setaffinity(0, cpus)
            share = max(1, min(len(cpus), cores_all // world))
        cli.process_files(args, names[:8], NullEngine(), lambda m: None, gpu_workers=world, core_share=share)   # warm
        barrier.wait(120)
        st = cli.process_files(args, names, NullEngine(), lambda m: None, gpu_workers=world, core_share=share)
        q.put((rank, st["bases"], st["parser_workers"], None, sorted(cpus) if cpus else None))
    except BaseException as e:
        try:
            barrier.abort()
        except Exception:
            pass
        q.put((rank, 0, 0, repr(e), None))


def host_capacity(cores, min_s=2.0, max_reads=24000):
    """bases/s through `cli.process_files` with a NullEngine: how many bases per second this host can FEED to GPUs.
    Runs in the helper child (`bench.py --helper`): a fresh process with no OpenMP / torch thread pools alive (the
    r04 figure was taken right behind a 16-thread OpenMP leg and a 16-thread torch leg on a 16-core quota, with a
    57 ms sample: the driver saw 27.6 M where other boxes print 95 M).  Every point is a calibration pass followed
    by ONE timed pass over enough reads for >= min_s seconds (at most max_reads):
      workers_1 / workers_4 / workers_<cores>   parser threads of one GPU worker, cap lifted (what w threads deliver)
      cli_1gpu                                  the pool the command line really runs for one GPU (cli.kNativePoolMax)
      cli_8gpu_workers                          8 worker PROCESSES as `NanoReviser.py --gpus 8` starts them, each with
                                                its share of the cores (cli.parser_pool_size) and its NUMA pinning"""
    import multiprocessing as mp
    import shutil
    import tempfile
    from nanoreviser_amd import cli
    base, base_what = scratch_base()
    tmp = tempfile.mkdtemp(prefix="nrv_hostcap_", dir=base)
    per_read = 6800.0                                    # bases per fixture read, refined by the calibration pass
    out = {"unit": "bases/s", "min_sample_s": min_s, "files_on": base_what,
           "what": "cli.process_files with an engine that computes nothing: fast5 parse, event collapse, statistics, packing, "
                   "merge, FASTA write of the committed fixture reads (hard links); fresh child process, one timed pass per point"}
    try:
        def one(tag, threads, cap):
            nonlocal per_read
            keep = cli.kNativePoolMax
            os.environ.pop("NRV_PARSER_THREADS_MAX", None)
            cli.kNativePoolMax = cap
            try:
                odir = os.path.join(tmp, "out_" + tag)
                args = cli.get_args(["-d", os.path.join(tmp, "in") + "/", "-o", odir + "/", "-S", "ecoli", "--thread", str(threads)])
                os.makedirs(odir, exist_ok=True)
                n_cal = 200 * max(1, min(threads, cap))
                din, names = _hostcap_files(tmp, n_cal)
                t0 = time.perf_counter()
                st = cli.process_files(args, names, NullEngine(), lambda m: None)
                dt = time.perf_counter() - t0
                per_read = st["bases"] / max(st["reads"], 1)
                n = int(min(max_reads, max(n_cal, 1.25 * min_s * st["reads"] / max(dt, 1e-3))))
                din, names = _hostcap_files(tmp, n)
                t0 = time.perf_counter()
                st = cli.process_files(args, names, NullEngine(), lambda m: None)
                dt = time.perf_counter() - t0
                out[tag] = st["bases"] / dt
                out[tag + "_detail"] = {"reads": st["reads"], "seconds": dt, "parser_threads": st["parser_workers"],
                                        "host_ms_per_read": st["host_s"] / max(st["reads"], 1) * 1e3}
                out["host_stage"] = st.get("host_stage")
                shutil.rmtree(odir, ignore_errors=True)
            finally:
                cli.kNativePoolMax = keep
        for w in sorted({1, min(4, cores), cores}):
            one(f"workers_{w}", w, 64)
        one("cli_1gpu", cores, cli.kNativePoolMax)
        # ---- eight GPU workers, as processes: three runs on the scratch filesystem (median + all), one writing to the default
        # temporary directory (the sandbox's overlay root: see scratch_base)
        world = 8
        rate1 = out.get(f"workers_{cores}", 2e7)
        n = int(min(max_reads, max(800, 1.25 * min_s * max(rate1, 8e7) / per_read)))
        din, names = _hostcap_files(tmp, n)
        ctx = mp.get_context("spawn")

        def eight(out_root):
            barrier, q = ctx.Barrier(world + 1), ctx.Queue()
            procs = [ctx.Process(target=_hostcap_worker, args=(r, world, din, names[r::world], os.path.join(out_root, f"out8_{r}"), cores, barrier, q))
                     for r in range(world)]
            for pr in procs:
                pr.start()
            try:
                barrier.wait(180)
                t0 = time.perf_counter()
                res = [q.get(timeout=300) for _ in procs]
                dt = time.perf_counter() - t0
                errs = [e for _, _, _, e, _ in res if e]
                if errs:
                    return {"error": errs[0]}
                return {"rate": sum(b for _, b, _, _, _ in res) / dt, "seconds": dt, "threads": sorted({t for _, _, t, _, _ in res}),
                        "placement": {str(r): (f"{c[0]}-{c[-1]} ({len(c)} cpus)" if c else "not pinned") for r, _, _, _, c in sorted(res)}}
            except Exception as e:
                return {"error": repr(e)}
            finally:
                for pr in procs:
                    pr.join(30)
                    if pr.is_alive():
                        pr.terminate()
                for r in range(world):
                    shutil.rmtree(os.path.join(out_root, f"out8_{r}"), ignore_errors=True)
        runs = [eight(tmp) for _ in range(3)]
        good = sorted(r["rate"] for r in runs if "rate" in r)
        if good:
            out["cli_8gpu_workers"] = good[len(good) // 2]
            out["cli_8gpu_workers_detail"] = {"reads": n, "runs": [r.get("rate", r.get("error")) for r in runs],
                                              "seconds": [r.get("seconds") for r in runs], "worker_processes": world,
                                              "parser_threads_per_worker": runs[0].get("threads"),
                                              "placement": runs[0].get("placement"),
                                              "placement_rule": "cli.worker_cpus(rank, 8, devices 0..7): GPU r's NUMA node where the box has "
                                                                "that GPU, contiguous slices of the allowed cores otherwise"}
        else:
            out["cli_8gpu_workers"] = {"error": runs[0].get("error")}
        if base is not None:
            slow = tempfile.mkdtemp(prefix="nrv_hostcap_out_")
            try:
                r = eight(slow)
                out["cli_8gpu_workers_writing_to_default_tmp"] = r.get("rate", r.get("error"))
                out["default_tmp"] = tempfile.gettempdir()
            finally:
                shutil.rmtree(slow, ignore_errors=True)
        return out
    except Exception as e:
        out["error"] = repr(e)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def blas_grade_baseline(m1, m2, T, sig, rd, cores, budget_s):
    """The same graph with every contraction on the host's BLAS (PyTorch-CPU: oneDNN / MKL GEMMs, float32,
    oracle/torch_cpu.py) - what a Keras-on-TF-MKL stack would roughly deliver (NanoReviser.py:37-38 forces the CPU),
    next to the scalar C port.  Checked against the fp64 oracle on 64 windows before it is timed."""
    try:
        import torch
        from oracle import torch_cpu as TC, nrv_oracle as O
        torch.set_num_threads(max(1, cores))
        a, b = TC.TorchCpuModel(m1.tensors), TC.TorchCpuModel(m2.tensors)
        q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sig[:64], rd[:64], np.float64)
        p1, p2, _, _ = TC.predict_pair(a, b, sig[:64], rd[:64])
        dp = max(float(np.abs(p1 - q1).max()), float(np.abs(p2 - q2).max()))
        if dp > 1e-4:
            return {"error": f"torch-CPU evaluation is {dp:.2e} from the fp64 oracle: not timed"}
        n0 = min(len(rd), 512)
        t0 = time.perf_counter()
        TC.predict_pair(a, b, sig[:n0], rd[:n0])
        t_cal = time.perf_counter() - t0
        n = int(min(len(rd), max(n0, n0 * budget_s / max(t_cal, 1e-3))))
        reps = max(1, int(budget_s / max(t_cal * n / n0, 1e-3))) if n == len(rd) else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            TC.predict_pair(a, b, sig[:n], rd[:n])
        t = time.perf_counter() - t0
        return {"value": n * reps / t, "unit": "bases/s", "cores": cores, "kind": "port (BLAS-grade)",
                "dtype": "f32", "max_abs_dp_vs_fp64_oracle_64_windows": dp,
                "sample": f"{n} windows x {reps}, model1+model2, torch {torch.__version__} CPU with {cores} threads, {t:.1f} s",
                "note": "oneDNN / MKL GEMMs for the convolutions, dense layers and the Bi-LSTM input projections (all T "
                        "steps per GEMM), one GEMM per recurrent step; reported baseline only"}
    except Exception as e:
        return {"error": repr(e)}


def load_traffic(T, batch, precision, profiles_dir=None):
    """(HBM bytes per lstm3 launch, source description) from the newest committed PMC pass that still matches the
    kernel; (None, why) when there is none."""
    import glob
    why = "no committed PMC pass for this mode / shape"
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "r*_pmc_lstm3.json")), reverse=True):
        try:
            j = json.load(open(path)).get(precision)
        except Exception:
            continue
        if not j or j.get("T") != T or j.get("batch") != batch:
            continue
        name = os.path.relpath(path, ROOT) if profiles_dir is None else os.path.basename(path)
        if "kernel_name" in j and KERNEL_SIGNATURE[precision] not in j["kernel_name"]:
            why = f"{name} describes another kernel ({j['kernel_name'][:60]}...): not printed"
            continue
        if "source_sha256_16" in j and j["source_sha256_16"] != kernel_source_sha(precision):
            why = f"{name} was collected on another version of {' + '.join(KERNEL_SOURCES[precision])}: stale, not printed"
            continue
        if "kernel_name" not in j or "source_sha256_16" not in j:
            why = f"{name} carries no kernel name / source hash (rounds 1-3): not trusted for the current kernel"
            continue
        return j.get("hbm_bytes_per_launch"), f"{name} @ {j.get('commit', '?')}: {j.get('source', '')}", j
    return None, why, {}


# ------------------------------------------------------------------------------------------------
# secondary measurements (rank 0, one GPU)
# ------------------------------------------------------------------------------------------------
def _time_calls(fn, sync, reps, warm=1):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def fixture_reads():
    """The five fixture reads (unitest/test_data/fast5, pre-extracted: tests/golden/reads) through the
    product's host stage -> RawReadTensors list."""
    from nanoreviser_amd import hoststage as hs
    idx = json.load(open(os.path.join(ROOT, "tests", "golden", "reads", "index.json")))
    out = []
    for e in idx:
        g = np.load(os.path.join(ROOT, "tests", "golden", "reads", e["key"] + ".npz"))
        rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"], g["ev_move"],
                                g["raw_signal"])
        out.append(hs.read_tensors_raw(rd))
    return out


def extras(args, torch, dev, local_rank, m1, m2, sig, rd, dev_ms_per_step, cli_helper=None):
    from nanoreviser_amd.engine import Reviser
    from nanoreviser_amd.weights import load_species
    from nanoreviser_amd import workload as W
    T, B = args.window, args.batch
    out = {}

    def sync():
        torch.cuda.synchronize()

    def outputs(n):
        return (torch.empty(n, 6, device=dev), torch.empty(n, 5, device=dev),
                torch.empty(n, dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.int8, device=dev))

    # ---- f32 mode (IEEE f32 matrix instructions: the reference's own precision) - same protocol as `value`
    rv = Reviser(m1, m2, device=local_rank, batch=B, precision="f32")
    rv.set_stream(torch.cuda.current_stream().cuda_stream)
    d_sig, d_rd = torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)
    o = outputs(B)
    ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B) + tuple(x.data_ptr() for x in o)
    m = measure(rv, lambda: rv.predict_device(*ptrs), sync, Dist(), args, prime=max(args.prime // 3, 20), rank0=True)
    out.update(roofline_blocks(args, T, B, "f32", m, suffix="_f32"))
    out["roofline_f32"].update({"ms_per_step": m["ms_per_step"], "bases_per_s": B / (m["ms_per_step"] * 1e-3),
                                "steps": args.steps, "warmup": args.warmup, "prime": max(args.prime // 3, 20)})
    rv.close()
    del d_sig, d_rd

    # ---- host-inclusive: inputs and outputs in HOST memory, H2D / D2H inside the timed call
    rv = Reviser(m1, m2, device=local_rank, batch=B, precision=args.precision)
    dev_rate = B / (dev_ms_per_step * 1e-3)
    hi = {}
    for G, key in ((32, "nrv_predict"), (8, "nrv_predict_8_groups")):
        # a call = G launch groups of B windows from HOST arrays (2958 B per base over PCIe at T = 13); what a call costs
        # besides its kernels is one stage of upload at its start (12 MB) and the last download: the larger call is the
        # rate of the entry point, the 8-group call (97 MB, r01-r04's figure) is kept beside it
        hsig, hrd = np.tile(sig, (G, 1, 1)), np.tile(rd, (G, 1, 1))
        dt = _time_calls(lambda: rv.predict_pair(hsig, hrd), lambda: None, 3)
        hi[key] = {"bases_per_s": G * B / dt, "windows": G * B, "host_bytes_per_call": int(hsig.nbytes + hrd.nbytes),
                   "bytes_per_base_over_pcie": T * 56 * 4 + 11 * 4 + 2,
                   "vs_device_resident": (G * B / dt) / dev_rate}
    del hsig, hrd
    out["host_inclusive"] = hi
    rv.close()
    if cli_helper is not None:                               # the command line itself, in a process of its own
        cores, _, _ = host_cores()
        hi["cli_e2e"] = cli_helper.ask({"cmd": "cli", "reps": args.cli_reps, "threads": min(cores, 16), "gpus": 1})
        # BASELINE.json configs[2] through the command line: human weights, 10 000 fast5 files in, 10 000 FASTA files out,
        # built from ALL FIVE of the reference's fixture reads, every file a copy of its own (tmpfs)
        if args.cli_human_reps > 0:
            hi["cli_e2e_human"] = cli_helper.ask({"cmd": "cli", "reps": args.cli_human_reps, "threads": min(cores, 16), "gpus": 1,
                                                  "species": "human", "all_five": True, "copy": True})

    # ---- C5: one long read, human weights, streamed in device-formed window groups
    h1, h2 = load_species("human")
    h1, h2 = h1.with_window(T), h2.with_window(T)
    rv = Reviser(h1, h2, device=local_rank, batch=B, precision=args.precision)
    rv.set_stream(torch.cuda.current_stream().cuda_stream)
    N = 200_000
    sev, fev = W.synth_read(N)
    d_sev, d_fev = torch.from_numpy(sev).to(dev), torch.from_numpy(fev).to(dev)
    o = outputs(N - T)
    rp = (d_sev.data_ptr(), d_fev.data_ptr(), N) + tuple(x.data_ptr() for x in o)
    dt = _time_calls(lambda: rv.predict_read_device(*rp), sync, 5)
    fl = FLOP_PER_BASE_DEDUP.get(T, flop_per_window(T))
    peak, _ = mode_peak(args.precision)
    ach = (N - T) * fl / dt / 1e12
    out["read_mode"] = {"config": f"C5: human weights, one synthetic read of {N} events, T={T}, device-formed windows "
                                  f"in groups of {B}, inputs resident in HBM",
                        "bases_per_s": (N - T) / dt, "flop_per_base_dedup": fl, "achieved_tflops": ach,
                        "frac": ach / peak, "peak": peak}
    dt = _time_calls(lambda: rv.predict_read(sev, fev), lambda: None, 3)
    hi["nrv_predict_read"] = {"bases_per_s": (N - T) / dt, "events": N, "bytes_per_base_over_pcie": 224 + 46,
                              "vs_device_resident": ((N - T) / dt) / out["read_mode"]["bases_per_s"]}
    rv.close()
    del d_sev, d_fev

    # ---- C2 / C3: the fixture reads (T = 11, shipped weights), replicated
    try:
        reads = fixture_reads()
    except Exception as e:                                   # fixtures missing: say so, do not invent
        out["configs"] = {"error": repr(e)}
        return out
    sev = np.concatenate([r.sig_ev for r in reads])
    fev = np.concatenate([r.feat_ev for r in reads])
    nb = sum(len(r.feat_ev) for r in reads)
    cfg = {}
    # BASELINE.json configs[1] "~1k reads, batch=512" and configs[2] "human model, ~10k reads, batch=4096": 200 / 2000 passes
    # over the five fixture reads (one pass = one call = 5 reads; the same read is revised again, results are not cached)
    for name, sp, batch, reps in (("C2", "ecoli", 512, args.c2_reps), ("C3", "human", 4096, args.c3_reps)):
        a, b = load_species(sp)
        rv = Reviser(a, b, device=local_rank, batch=batch, precision=args.precision)
        rv.set_stream(torch.cuda.current_stream().cuda_stream)
        d_sev, d_fev = torch.from_numpy(sev).to(dev), torch.from_numpy(fev).to(dev)
        n = len(fev) - a.T
        o = outputs(n)
        rp = (d_sev.data_ptr(), d_fev.data_ptr(), len(fev)) + tuple(x.data_ptr() for x in o)
        dt = _time_calls(lambda: rv.predict_read_device(*rp), sync, reps)
        raws = [[r.raw for r in reads], [r.starts for r in reads], [r.feat_ev for r in reads],
                [r.shift for r in reads], [r.scale for r in reads]]
        dth = _time_calls(lambda: rv.predict_reads_raw(*raws), lambda: None, max(2, reps // 4))
        # ... and the same entry point as the command line drives it since r06: two calls in flight (nrv_reads_raw_begin / _end),
        # call k+1 enqueued before call k is collected
        npipe = max(4, reps // 4)
        packs = [rv.pack_reads_raw(*raws, a.T) for _ in range(2)]
        rv.end_packed_raw(rv.begin_packed_raw(packs[0]))                                   # warm
        t0 = time.perf_counter()
        tk = rv.begin_packed_raw(packs[0])
        for i in range(1, npipe):
            nxt = rv.begin_packed_raw(packs[i & 1])
            rv.end_packed_raw(tk)
            tk = nxt
        rv.end_packed_raw(tk)
        dtp = (time.perf_counter() - t0) / npipe
        cfg[name] = {"config": f"{sp} weights, T={a.T}, batch={batch} windows per launch group, the five fixture reads "
                               f"({nb} bases) x {reps} = {5 * reps} reads; fast5 parsing excluded",
                     "reads": 5 * reps, "bases": n * reps, "seconds_device_resident": dt * reps,
                     "reads_host_inclusive": 5 * max(2, reps // 4),
                     "bases_per_s_device_resident": n / dt,
                     "bases_per_s_host_inclusive_raw_reads": n / dth,
                     "bases_per_s_host_inclusive_raw_reads_pipelined": n / dtp,
                     "achieved_tflops": n * FLOP_PER_BASE_DEDUP[11] / dt / 1e12}
        rv.close()
    out["configs"] = cfg
    hi["nrv_predict_reads_raw"] = {"bases_per_s": cfg["C3"]["bases_per_s_host_inclusive_raw_reads"],
                                   "bytes_per_base_over_pcie": 46 + 46,
                                   "vs_device_resident": cfg["C3"]["bases_per_s_host_inclusive_raw_reads"]
                                   / cfg["C3"]["bases_per_s_device_resident"]}
    hi["nrv_reads_raw_begin_end"] = {"bases_per_s": cfg["C3"]["bases_per_s_host_inclusive_raw_reads_pipelined"],
                                     "bytes_per_base_over_pcie": 46 + 46, "calls_in_flight": 2,
                                     "vs_device_resident": cfg["C3"]["bases_per_s_host_inclusive_raw_reads_pipelined"]
                                     / cfg["C3"]["bases_per_s_device_resident"],
                                     "what": "the same calls with call k+1 enqueued before call k is collected (how the command line drives the engine)"}
    return out


# ------------------------------------------------------------------------------------------------
# the measurement protocol (used for `value` and for roofline_f32)
# ------------------------------------------------------------------------------------------------
def measure(rv, step, sync, d, args, prime, rank0):
    """Untimed priming (the first unsynchronised bursts of launches of a process pay one-off costs, ~35 ms
    inside the first ~1000 launches, and the chip needs ~0.2 s of load to settle on the clock it sustains),
    W warm-up steps, then EXACTLY K steps between barrier + device sync; the dominant kernel is bracketed by
    hipEvents on every 8th launch INSIDE the timed region; every kernel in an untimed pass behind it."""
    for _ in range(prime):
        step()
    sync()
    # ... and then until the step time has SETTLED: blocks of 200 untimed steps, each timed on its own, until three in a
    # row agree within 1 % (at most 40 blocks, ~3 s).  A box that has just run something else - the GPU test suite, in
    # the driver's sequence - can sit in a lower clock state for seconds: r03m's first bench process on its box timed
    # 0.409 ms per step (dominant kernel 188 us), the second, a minute later, 0.341 (152 us).
    settle = []
    if prime >= 100 and not getattr(args, "no_settle", False):
        while len(settle) < 40:
            t0 = time.perf_counter()
            for _ in range(200):
                step()
            sync()
            settle.append((time.perf_counter() - t0) / 200 * 1e3)
            if len(settle) >= 3 and max(settle[-3:]) <= 1.01 * min(settle[-3:]):
                break
    prof_mode = 0
    if not args.no_prof:
        prof_mode = 1 if args.prof_all else 3            # 3: the dominant kernel, every 8th launch
        rv.prof_enable(prof_mode)
        rv.prof_read()
    for _ in range(args.warmup):
        step()
    sync()
    if prof_mode:
        rv.prof_enable(prof_mode)                        # restart the every-8th tick; drops warm-up samples
        rv.prof_read()
    elapsed, mine = timed_steps(d, step, sync, args.steps, 0)
    prof = rv.prof_read() if prof_mode else {}
    # the spread of THIS run: the same K-step block four more times (barrier + sync around each, max over ranks), so that a
    # reader can tell box-to-box variance (+-4-8 % between boxes) from change; `value` stays the first block's (the contract)
    blocks = [elapsed / args.steps * 1e3]
    for _ in range(max(0, getattr(args, "blocks", 5) - 1)):
        e2, _ = timed_steps(d, step, sync, args.steps, 0)
        blocks.append(e2 / args.steps * 1e3)
    if prof_mode:
        rv.prof_read()                                   # drop the extra blocks' samples: roofline = the contract block's
    kernel_us = {}
    if prof_mode and rank0:
        rv.prof_enable(1)
        rv.prof_read()
        for _ in range(max(8, min(args.steps, 32))):
            step()
        sync()
        kernel_us = {k: (ms / max(c, 1)) * 1e3 for k, (ms, c) in rv.prof_read().items() if c > 0}
    rv.prof_enable(0)
    sync()
    overhead_us = rv.prof_overhead_us() if prof_mode and rank0 else 0.0
    return {"elapsed": elapsed, "mine": mine, "ms_per_step": elapsed / args.steps * 1e3, "prof": prof,
            "kernel_us": kernel_us, "prof_mode": prof_mode, "bracket_overhead_us": overhead_us,
            "settle_ms": [round(x, 4) for x in settle], "blocks_ms": blocks}


KERNEL_NAME = {"bf16x3": "lstm_split_kernel<32,16,128,2,1>", "f32": "lstm_layer_kernel<32,16,128,1,1>",
               "f16x2": "lstm_h2w_kernel<32,16,128> (16x16x32 f16 tiles, eight waves in two groups)"}


def rocprof_duration_us(precision, profiles_dir=None):
    """Average duration of the dominant kernel in the newest COMMITTED rocprofv3 kernel trace of the timed region
    (profiles/r*_kernel_stats_timed_region_<precision>.csv, written by scripts/gpu_prof.sh from this very command) ->
    (microseconds, file name) or (None, reason).  The line's own hipEvent figure and the committed trace then sit side by
    side in `roofline` (frac / frac_rocprof) and cannot drift apart unseen."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), f"r*_kernel_stats_timed_region_{precision}.csv")))
    for path in reversed(files):
        try:
            for row in csv.DictReader(open(path)):
                if KERNEL_SIGNATURE[precision] in row.get("Name", ""):
                    return float(row["AverageNs"]) / 1e3, os.path.basename(path)
        except Exception:
            continue
    return None, "no committed kernel trace names " + KERNEL_SIGNATURE[precision]


def roofline_blocks(args, T, B, precision, m, suffix=""):
    """roofline / roofline_whole_step / kernel_us of one measure() result."""
    prof, kernel_us = m["prof"], m["kernel_us"]
    out = {}
    if not prof:
        return out
    peak, peak_note = mode_peak(precision)
    names = list(prof.keys())
    k3 = names[3]
    if prof[k3][1] > 0:
        avg_s = prof[k3][0] / prof[k3][1] * 1e-3
        where = "hipEvent pairs on the launch stream inside the timed region" + \
            (", every 8th launch bracketed" if m["prof_mode"] == 3 else "")
        n_l = prof[k3][1]
    else:                                             # fewer than one sampled launch: use the untimed pass
        avg_s = kernel_us[k3] * 1e-6
        where, n_l = "hipEvent pairs on the launch stream, untimed pass right after the timed region", 0
    # A bracket is event record -> kernel -> event record: it contains what an EMPTY bracket measures on this stream
    # on top of the kernel's own duration (~5 us; a rocprofv3 kernel trace of the same command shows the kernel alone:
    # profiles/r03*_kernel_stats_timed_region_*.csv).  The roofline figure is the kernel's: bracket minus empty bracket.
    raw_s = avg_s
    avg_s = max(avg_s - m.get("bracket_overhead_us", 0.0) * 1e-6, 1e-9)
    fl = flop_lstm3_launch(T, B, executed=True)
    ach = fl / avg_s / 1e12
    out["roofline" + suffix] = {
        "kernel": f"{KERNEL_NAME[precision]} ({k3})", "bound": "mfma", "achieved": ach, "peak": peak,
        "unit": "TFLOP/s", "frac": ach / peak,
        "traffic": load_traffic(T, B, precision)[0], "traffic_source": load_traffic(T, B, precision)[1],
        "flop_per_launch": fl, "avg_launch_us": avg_s * 1e6, "launches": n_l,
        "avg_launch_us_bracketed": raw_s * 1e6, "empty_bracket_us": m.get("bracket_overhead_us", 0.0),
        "avg_launch_us_untimed_pass": kernel_us.get(k3),
        "timing": where, "peak_note": peak_note,
        "executed_tflops": ach * PRODUCTS[precision],
        "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
    }
    rp_us, rp_src = rocprof_duration_us(precision)
    out["roofline" + suffix].update({"frac_rocprof": (fl / (rp_us * 1e-6) / 1e12 / peak) if rp_us else None,
                                     "avg_launch_us_rocprof": rp_us, "rocprof_source": rp_src})
    # north_star: "rocprof HBM GB/s and MFMA utilisation reported against gfx950 peak" - from the committed PMC pass of this
    # kernel (null, with the reason in traffic_source, when that pass no longer describes the kernel)
    tr, _, rec = load_traffic(T, B, precision)
    r = out["roofline" + suffix]
    r["hbm_gbps"] = tr / avg_s / 1e9 if tr else None
    r["hbm_frac_of_peak"] = tr / avg_s / 8e12 if tr else None
    r["mfma_busy_frac"] = rec.get("mfma_busy_frac")
    r["mfma_busy_note"] = rec.get("mfma_busy_note")
    step_ok = rec.get("step_source_sha256_16") == step_source_sha() if rec.get("traffic_step") else False
    r["traffic_step"] = rec.get("traffic_step") if step_ok else None
    r["traffic_step_by_kernel"] = rec.get("traffic_step_by_kernel") if step_ok else None
    if rec.get("traffic_step") and not step_ok:
        r["traffic_step_note"] = "the committed PMC pass was collected on another version of the step's kernels: not printed"
    elif step_ok:
        r["traffic_step_note"] = (f"HBM-side bytes of all launches of one {B}-window step (PMC, FETCH_SIZE x2 + WRITE_SIZE) against "
                                  f"{B * (T * 56 * 4 + 11 * 4 + 2)} algorithmic bytes in / out")
    if precision == "f16x2":
        # measured, not assumed (profiles/r06_clock_vs_fill.txt)
        out["roofline" + suffix]["bound_note"] = (
            "what binds this launch is its weight stream, not the matrix pipe and not a power plateau (r06: the layers' inner loop "
            "alone, every wave stamped): 2 KB of weights from L2 per wave and 12 products = ~30 B per cycle and CU, the rate the L2 "
            "delivers to a CU; it holds the pipe's fill at 0.70 and costs clock (1.89 GHz on random operands); the same loop "
            "without the stream fills the pipe to 0.94 at 1.82-1.86 GHz = 600-611 TFLOP/s f32-grade (0.72-0.73 of `peak`)")
    whole = flop_per_window(T) * B / (m["ms_per_step"] * 1e-3) / 1e12
    out["roofline_whole_step" + suffix] = {"achieved": whole, "peak": peak, "unit": "TFLOP/s", "frac": whole / peak,
                                           "frac_of_f32_mfma_peak": whole / PEAK_F32_MFMA_TFLOPS,
                                           "flop_per_window": flop_per_window(T)}
    out["kernel_us" + suffix] = kernel_us
    out["kernel_us_sum" + suffix] = sum(kernel_us.values())
    if precision == "f16x2":
        out["kernel_us_note" + suffix] = ("bracketed figures (each contains one empty bracket, roofline.empty_bracket_us); f16x2: "
                                          "the 6->16 Bi-LSTM runs as four extra waves of the signal-branch launch "
                                          "(cnn_r_kernel), so its own slot shows only the empty bracket")
    return out


class Helper:
    """A child of rank 0 that never touches a GPU, started BEFORE this process makes its first HIP call (a process that
    has initialised the GPU must not exec another program; a plain child that was forked off earlier may start whatever
    it likes).  It takes one JSON command per line on stdin and answers with one JSON line:
      {"cmd": "cli", "gpus": N, ...}   the command line itself, fast5 files in, FASTA files out, N GPU workers, as ITS
                                       child process, timed from outside (NanoReviser.py:105-183, 203-219)
      {"cmd": "hostcap", "cores": C}   `host_capacity` in this fresh process (no OpenMP / torch pools alive)"""

    def __init__(self):
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--helper"],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)

    def ask(self, cmd, timeout=2400):       # >= the sum of the child timeouts of one command (cli_e2e: 2 x 800 s)
        import threading
        box = {}

        def talk():
            try:
                self.p.stdin.write(json.dumps(cmd) + "\n")
                self.p.stdin.flush()
                while True:
                    line = self.p.stdout.readline()
                    if not line:
                        box["r"] = {"error": "helper ended without an answer"}
                        return
                    if line.startswith("{"):
                        box["r"] = json.loads(line)
                        return
            except Exception as e:
                box["r"] = {"error": repr(e)}
        t = threading.Thread(target=talk, daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            self.p.terminate()                               # SIGTERM: helper_main's handler ends the command line it started
            try:
                self.p.wait(10)
            except Exception:
                self.p.kill()
            return {"error": f"helper did not answer {cmd.get('cmd')} within {timeout} s"}
        return box.get("r", {"error": "no answer"})

    def close(self):
        if self.p.poll() is None:
            try:
                self.p.communicate(json.dumps({"cmd": "quit"}) + "\n", timeout=10)
            except Exception:
                self.p.kill()


_CHILDREN = []          # process groups the helper has started (the command line and its GPU workers)


def _run_child(cmd, timeout, env):
    """subprocess.run in a process group of its own, remembered so that a helper that is told to stop (or times out)
    takes the command line AND its GPU workers with it (a killed helper left the grandchild on the GPUs)."""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    _CHILDREN.append(p)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
        err = (err or "") + f"\n[bench] timed out after {timeout} s"
        p.returncode = p.returncode if p.returncode else -9
    finally:
        _CHILDREN.remove(p)
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def helper_main():
    """Body of the helper child (no GPU call in this process, ever)."""
    import signal

    def stop(signum, frame):
        for p in list(_CHILDREN):
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
        os._exit(1)
    signal.signal(signal.SIGTERM, stop)
    for line in sys.stdin:
        try:
            cmd = json.loads(line)
        except ValueError:
            continue
        if cmd.get("cmd") == "quit":
            break
        try:
            if cmd.get("cmd") == "cli":
                res = cli_e2e(cmd.get("reps", 4000), cmd.get("threads", 16), cmd.get("gpus", 1), cmd.get("share", False),
                              species=cmd.get("species", "ecoli"), all_five=bool(cmd.get("all_five", False)),
                              copy=bool(cmd.get("copy", False)))
            elif cmd.get("cmd") == "hostcap":
                res = host_capacity(int(cmd.get("cores", 1)), float(cmd.get("min_s", 2.0)))
            else:
                res = {"error": f"unknown command {cmd.get('cmd')!r}"}
        except Exception as e:
            res = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    return 0


def cli_e2e(reps, threads, gpus=1, share=False, species="ecoli", all_five=False, copy=False):
    """`python NanoReviser.py --gpus N -S <species>` on the committed fixture reads x reps x N (weak scaling: the file set
    grows with the GPUs), as a child process of the helper, wall time from outside, start-up included.
    all_five: the reference's five fixture reads (tests/golden/fast5 + fast5_more) instead of the first two.
    copy: every input file is a COPY - its own inode, its own pages - instead of a symbolic link to one of the fixtures
    (two hot files read 4000 times each sit in the CPU caches; 10 000 distinct files do not);
    falls back to links, and says so, when the scratch filesystem has no room for reps x the fixtures' bytes."""
    import glob
    import shutil
    import tempfile
    src = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fast5", "*.fast5")))
    if all_five:
        src = sorted(src + glob.glob(os.path.join(ROOT, "tests", "golden", "fast5_more", "*.fast5")))
    base, base_what = scratch_base()
    work = tempfile.mkdtemp(prefix="nrv_cli_e2e_", dir=base)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NRV_BENCH_CHILD"):
        env.pop(k, None)                                     # the command line is not a rank of this job
    if share:
        env["NRV_SHARE_DEVICE"] = "1"
    try:
        din, dout = os.path.join(work, "in"), os.path.join(work, "out") + "/"
        os.makedirs(din)
        need = sum(os.path.getsize(f) for f in src) * reps * gpus
        how = "symbolic links"
        if copy:
            free = shutil.disk_usage(work).free
            try:
                avail = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) * 1024
            except Exception:
                avail = free
            # on tmpfs the copies are memory: leave two thirds of what is free / available alone
            how = "copies (one inode and one set of pages per read)" if 3 * need < min(free, avail) else \
                f"symbolic links (no room for {need >> 20} MiB of copies: {min(free, avail) >> 20} MiB free)"
        for i, f in enumerate(src):
            for k in range(reps * gpus):
                dst = os.path.join(din, f"r{i}_{k}.fast5")
                if how.startswith("copies"):
                    shutil.copyfile(f, dst)
                else:
                    os.symlink(f, dst)
        res = {"reads": len(src) * reps * gpus, "threads": threads, "n_gpus": gpus, "share_device": bool(share), "files_on": base_what,
               "species": species, "distinct_fixture_reads": len(src), "input_files_are": how, "input_bytes": need}
        for name, n in (("start_up", 2), ("run", None)):     # two reads first: what a run costs before it streams
            if n is not None:
                d2 = os.path.join(work, "in2")
                os.makedirs(d2, exist_ok=True)
                for f in sorted(os.listdir(din))[:n]:
                    if not os.path.exists(os.path.join(d2, f)):
                        os.symlink(os.path.realpath(os.path.join(din, f)), os.path.join(d2, f))
            t0 = time.perf_counter()
            r = _run_child([sys.executable, os.path.join(ROOT, "NanoReviser.py"), "-d", din if n is None else d2,
                            "-o", dout, "-S", species, "--thread", str(threads), "--gpus", str(gpus if n is None else 1)],
                           800, env)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                res["error"] = f"{name}: rc {r.returncode}: {r.stderr[-300:]}"
                break
            line = [ln for ln in r.stdout.splitlines() if "bases/s end to end" in ln]
            if name == "start_up":
                res["start_up_s_two_reads"] = dt
            else:
                nb = int(line[-1].split(" reads, ")[1].split(" bases")[0]) if line else 0
                res.update({"wall_s": dt, "bases": nb, "bases_per_s": nb / dt,
                            "files_written": len([f for f in os.listdir(dout) if f.endswith("_out.fasta")]),
                            "cli_report": line[-1].strip() if line else None})
        res["what"] = (f"python NanoReviser.py -d <{len(src)} committed fixture fast5 x {reps * gpus}, {how.split(' (')[0]}> -o <tmp> -S {species} "
                       f"--thread {threads} --gpus {gpus}: own HDF5 reader, event collapse, device-side segmentation, "
                       "model1+model2, merge, one FASTA per read; wall time of the child process, start-up included"
                       + ("; NRV_SHARE_DEVICE=1: the workers share the devices there are (rehearsal)" if share else ""))
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ------------------------------------------------------------------------------------------------
def device_key(x):
    """What makes two ranks' devices the same device: PCI address + UUID; when a runtime reports neither (all zeros / empty)
    the device INDEX stands in, so that an uninformative identity can never make a real N-GPU run look like one device."""
    informative = x.get("pci") not in (None, "", "0000:00:00") or str(x.get("uuid")) not in ("", "None")
    return (x.get("pci"), str(x.get("uuid"))) if informative else ("index", x.get("device"))


def check_world(args):
    """Before any rendezvous: a launcher's WORLD_SIZE and --gpus must agree."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; they must agree "
                 "(plain `python bench.py --gpus N` starts its own ranks)")


def run_dry(args):
    """--dry-run: the whole control path on CPU (gloo), no engine, no GPU."""
    from nanoreviser_amd import shard
    check_world(args)
    d = Dist("gloo")
    lo, hi = shard.shard_range(d.world * args.batch, d.rank, d.world)
    eng = DryRunEngine(hi - lo, args.window, 20260 + d.rank)
    elapsed, _ = timed_steps(d, eng.step, lambda: None, args.steps, args.warmup)
    total = d.sum_int((hi - lo) * args.steps)
    ranks = d.gather_obj({"rank": d.rank, "pid": os.getpid(), "shard": [lo, hi]})     # the real run gathers device identities this way
    if d.rank == 0:
        print(json.dumps({
            "metric": "bases revised/sec (whole node)", "value": total / elapsed, "unit": "bases/s",
            "n_gpus": args.gpus, "world_size": d.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none", "data": "dry-run (CPU control path only; no engine, rate is meaningless)",
            "config": {"workload": "dry run", "batch_windows_per_gpu": args.batch, "window": args.window,
                       "world_size": d.world, "ranks": ranks}}), flush=True)
    d.close()
    return 0


def run_rank(args):
    check_world(args)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    # the CLI helper must exist before this process makes its first HIP call (see CliHelper)
    cli_helper = None
    if rank == 0 and not args.no_extras and not (args.no_cli_e2e and args.no_cpu_baseline):
        cli_helper = Helper()
    import torch
    from nanoreviser_amd.engine import Reviser
    from nanoreviser_amd.weights import load_species
    from nanoreviser_amd import workload as W

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.share_device:
        device = local_rank % ndev                     # rehearsal: several ranks on one GPU
    elif local_rank >= ndev:
        sys.exit(f"bench.py: rank {local_rank} has no GPU ({ndev} visible); --share-device maps ranks onto "
                 "the devices there are")
    else:
        device = local_rank
    torch.cuda.set_device(device)
    d = Dist()
    # which physical device this rank drives: index, PCI bus id, UUID.  The data path has no collective, so "N ranks
    # seen" cannot be read off RCCL: config.devices is the proof that an N-GPU line ran on N DISTINCT devices.
    pr = torch.cuda.get_device_properties(device)
    ident = {"rank": d.rank, "local_rank": local_rank, "device": device, "name": pr.name,
             "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
             "uuid": str(getattr(pr, "uuid", ""))}
    devices = d.gather_obj(ident)
    n_distinct = len({device_key(x) for x in devices})
    if n_distinct < d.world and not args.share_device:
        # an N-GPU line that ran on fewer than N devices is not an N-GPU measurement (the data path has no collective that
        # would notice): refuse, loudly, before anything is timed (every rank sees the same gathered list and leaves)
        if d.rank == 0:
            print(f"bench.py: {d.world} ranks but only {n_distinct} distinct device(s) {[x['pci'] for x in devices]}; "
                  "--share-device maps ranks onto the devices there are (a rehearsal, never a scaling figure)", file=sys.stderr, flush=True)
        d.close()
        sys.exit(3)

    T, B = args.window, args.batch
    m1, m2 = load_species(args.species)
    m1, m2 = m1.with_window(T), m2.with_window(T)
    rv = Reviser(m1, m2, device=device, batch=B, precision=args.precision)
    stream = torch.cuda.current_stream()
    rv.set_stream(stream.cuda_stream)

    sig, rd = W.synth_windows(B, T, seed=20260 + d.rank)     # each rank owns its own shard
    dev = f"cuda:{device}"
    d_sig, d_rd = torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)
    p1 = torch.empty(B, 6, device=dev)
    p2 = torch.empty(B, 5, device=dev)
    a1 = torch.empty(B, dtype=torch.int8, device=dev)
    a2 = torch.empty(B, dtype=torch.int8, device=dev)
    ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())

    def step():
        rv.predict_device(*ptrs)

    def sync():
        torch.cuda.synchronize()

    # correctness guard before timing anything: 64 windows against the fp64 oracle (the checker)
    from oracle import nrv_oracle as O
    step(); sync()
    q1, q2, b1, b2 = O.predict_pair(m1.tensors, m2.tensors, sig[:64], rd[:64], np.float64)
    dp = max(float(np.abs(p1[:64].cpu().numpy() - q1).max()), float(np.abs(p2[:64].cpu().numpy() - q2).max()))
    if dp > 1e-4 or not (np.array_equal(a1[:64].cpu().numpy(), b1) and np.array_equal(a2[:64].cpu().numpy(), b2)):
        sys.exit(f"bench.py: HIP path disagrees with the oracle (max|dp|={dp:.2e}); refusing to time it")

    m = measure(rv, step, sync, d, args, prime=args.prime, rank0=d.rank == 0)
    elapsed, ms_per_step = m["elapsed"], m["ms_per_step"]
    total = d.sum_int(B * args.steps)
    value = total / elapsed
    rank_ms = m["mine"] / args.steps * 1e3
    rank_ms_min, rank_ms_max = d.min_float(rank_ms), d.max_float(rank_ms)
    per_rank = d.gather_obj({"rank": d.rank, "ms_per_step": rank_ms})                # every rank's own clock, next to its device
    if per_rank and devices:
        by_rank = {x["rank"]: x["ms_per_step"] for x in per_rank}
        for x in devices:
            x["ms_per_step"] = by_rank.get(x["rank"])
    # the f16x2 range guard must not have fired on the benchmark's own input (a re-run would not be in `value`)
    pending, _ = rv.saturated()
    if d.sum_int(pending) != 0:
        sys.exit("bench.py: the f16x2 range guard fired on the synthetic workload; the timed steps are not valid")

    out = {
        "metric": "bases revised/sec (whole node)", "value": value, "unit": "bases/s",
        "n_gpus": args.gpus, "world_size": d.world, "steps": args.steps, "warmup": args.warmup,
        "prime": args.prime,
        "ms_per_step": ms_per_step,
        "ms_per_step_blocks": [round(x, 5) for x in m["blocks_ms"]],
        "ms_per_step_blocks_min_median_max": [round(x, 5) for x in (min(m["blocks_ms"]), sorted(m["blocks_ms"])[len(m["blocks_ms"]) // 2],
                                                                   max(m["blocks_ms"]))],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[args.precision],
        "value_arithmetic": ("`value` is computed on " + DTYPE[args.precision] + "; the figure on the reference's own IEEE-f32 "
                             "arithmetic, same protocol, same run: value_reference / dtype_reference / roofline_reference")
                            if args.precision != "f32" else "`value` is computed on IEEE-f32 matrix instructions, the reference's precision",
        "data": "synthetic",
        "config": {
            "workload": f"{args.species} weights, synthetic independent {T}-event windows (SURVEY 8d C4 "
                        f"generator), batch={B} windows per GPU per step, model1+model2"
                        + ("; T=13 uses the shipped weights + seeded synthetic (78,16) feature kernel" if T != 11 else ""),
            "species": args.species, "window": T, "batch_windows_per_gpu": B,
            "precision": args.precision,
            "parallelism": f"read/window-sharded x{args.gpus}, no collectives; control plane on gloo (CPU tensors)"
                           + (f"; --share-device: {d.world} ranks on {ndev} device(s)" if args.share_device else ""),
            "world_size": d.world, "devices": devices,
            "distinct_devices": len({device_key(x) for x in devices}),
            "parity_guard_max_abs_dp": dp,
            "prime_note": f"{args.prime} untimed priming steps, then untimed blocks of 200 steps until three in a row agree "
                          f"within 1 % (<= 40 blocks), precede the {args.warmup} warm-up steps (clock settling)",
            "settle_ms_per_step": m["settle_ms"],
        },
        "f16x2_range_guard": {"pending_after_timed_region": 0},
        "rank_ms_per_step": {"min": rank_ms_min, "max": rank_ms_max},
    }
    if d.rank == 0:
        out.update(roofline_blocks(args, T, B, args.precision, m))
    rv.close()
    if d.rank == 0 and args.gpus == 1 and not args.no_extras:
        del d_sig, d_rd
        try:
            out.update(extras(args, torch, dev, device, m1, m2, sig, rd, ms_per_step, None if args.no_cli_e2e else cli_helper))
            # the reference-precision figure next to `value` (same protocol, same run): nobody should have to dig for it
            if "host_inclusive" in out and "nrv_predict" in out["host_inclusive"]:
                # SURVEY 8d "H2D/D2H included and also reported kernel-only": the literal drop-in for model.predict
                # (output_handeler.py:250-251), inputs and outputs in HOST memory; `value` stays the device-resident rate
                out["value_host_inclusive"] = out["host_inclusive"]["nrv_predict"]["bases_per_s"]
            if "roofline_f32" in out:
                out["value_f32"] = out["roofline_f32"]["bases_per_s"]
                out["ms_per_step_f32"] = out["roofline_f32"]["ms_per_step"]
                out["dtype_f32"] = DTYPE["f32"]
                # ... and under the names a reader looks for first
                out["dtype_reference"] = "f32"
                out["value_reference"] = out["roofline_f32"]["bases_per_s"]
                out["ms_per_step_reference"] = out["roofline_f32"]["ms_per_step"]
                out["roofline_reference"] = {k: out["roofline_f32"].get(k) for k in
                                             ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_rocprof", "avg_launch_us",
                                              "avg_launch_us_rocprof", "rocprof_source", "traffic")}
            if "cli_e2e_human" in out.get("host_inclusive", {}):
                out["cli_e2e_human"] = out["host_inclusive"]["cli_e2e_human"]
        except Exception as e:                               # never lose the main line to a secondary block
            out["extras_error"] = repr(e)
    if args.gpus > 1 and not args.no_extras and not args.no_cli_e2e:
        # The metric as BASELINE.json words it - fast5 files in, revised reads out, whole node: the command line with N GPU
        # workers (NanoReviser.py:203-219 fans out the same way) on the fixture reads x cli_reps x N, as a child of rank 0's
        # helper, while every rank of THIS job sits in the barrier with its engine closed (the GPUs are the CLI's).
        d.barrier()
        if d.rank == 0 and cli_helper is not None:
            cores, _, _ = host_cores()
            out["host_inclusive"] = {"cli_e2e": cli_helper.ask({"cmd": "cli", "reps": args.cli_reps, "threads": min(cores, 16),
                                                                "gpus": args.gpus, "share": bool(args.share_device)})}
            v = out["host_inclusive"]["cli_e2e"].get("bases_per_s")
            if v:
                out["value_cli_e2e"] = v
        d.barrier()
    host_cap = None
    if d.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and cli_helper is not None:
        host_cap = cli_helper.ask({"cmd": "hostcap", "cores": host_cores()[0], "min_s": args.hostcap_seconds})
    if cli_helper is not None:
        cli_helper.close()
    if d.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(m1, m2, T, sig, rd, host_cap=host_cap)
    if d.rank == 0:
        print(json.dumps(out), flush=True)
    d.close()
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="windows per GPU per step")
    ap.add_argument("--window", type=int, default=13, help="events per window (T)")
    ap.add_argument("--species", default="ecoli")
    ap.add_argument("--precision", default=os.environ.get("NRV_BENCH_PRECISION", "f16x2"),
                    choices=sorted(PRODUCTS), help="matrix arithmetic (include/nanorev.h, nrv_set_precision)")
    ap.add_argument("--prime", type=int, default=300, help="untimed steps before the warm-up (clock settling)")
    ap.add_argument("--no-settle", action="store_true", help="skip the adaptive settling blocks behind --prime")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the contract keys + roofline")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-all", action="store_true",
                    help="time every kernel INSIDE the timed region (7 event records per step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU / gloo run of the launch + sharding + timing control path (no engine)")
    ap.add_argument("--share-device", action="store_true",
                    help="rank r runs on device r %% device_count (rehearse N ranks on fewer GPUs; never the default)")
    ap.add_argument("--no-cli-e2e", action="store_true", help="skip host_inclusive.cli_e2e (the CLI in a child process)")
    ap.add_argument("--cli-reps", type=int, default=4000, help="cli_e2e: copies of each committed fixture read (x N with --gpus N)")
    ap.add_argument("--hostcap-seconds", type=float, default=2.0, help="cpu_baseline.host_capacity: seconds per point")
    ap.add_argument("--blocks", type=int, default=5, help="timed K-step blocks (the first is `value`; all are in ms_per_step_blocks)")
    ap.add_argument("--c2-reps", type=int, default=200, help="configs.C2: passes over the five fixture reads (200 = 1000 reads)")
    ap.add_argument("--c3-reps", type=int, default=2000, help="configs.C3: passes over the five fixture reads (2000 = 10 000 reads)")
    ap.add_argument("--cli-human-reps", type=int, default=2000,
                    help="cli_e2e_human: copies of each of the five fixture reads (2000 = 10 000 files, ~8.8 GB on tmpfs; 0 = skip)")
    ap.add_argument("--helper", action="store_true", help=argparse.SUPPRESS)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)
    if args.helper:
        return helper_main()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)
    return run_dry(args) if args.dry_run else run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
