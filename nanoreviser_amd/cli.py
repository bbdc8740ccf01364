"""Command line of the MI355X reviser: the reference's `NanoReviser.py` contract, fast5 in,
`<read>_out.fasta` / `_out.fastq` out.

Kept from the reference (NanoReviser.py:42-95, 105-236):
  * the flag surface  -d -o -F -S --thread -t -e -g -s --test_mode --model1_predict_dir
    --model2_predict_dir -v ;
  * weight path convention ./model/<S>/<S>_win13_50ep_model{1,2}.h5 (:191-193);
  * one output file per read, `<output_dir><stem>_out.<fmt>` with stem = name up to the first '.'
    (:137, :163) and the record layout of output_handeler.py:26-62;
  * the failure contract: any error while revising a read writes the ORIGINAL basecalls instead
    (:146-152 fasta, :173-179 fastq via extract_fastq) - and, which the reference parses but never
    does (:63-65), the read is listed in the --failed_read file.
Changed on purpose:
  * the network is actually executed (the reference shells out to a missing Guppy binary,
    SURVEY.md F1): per read  events -> bases -> signal windows/features (hoststage) ->
    nrv_predict_read on the GPU -> merge (hoststage.revise_read);
  * no tmp-dir copies (the -t flag is accepted and ignored; the reference's per-slot tmp dirs race,
    :111) and no file is dropped when the count is not a multiple of the pool size (:212);
  * reads are sharded over the visible GPUs, one worker process per GPU, no collectives
    (--gpus, default all); --thread bounds the host-stage worker processes per GPU worker;
  * explicitly given --model{1,2}_predict_dir win over -S (in the reference -S always overrides
    them, which makes those flags dead, :191-193);
  * --resume: a rerun skips the reads whose output file exists, EXCEPT those the earlier run listed in its
    failed-reads file (their output is the unrevised fallback: they are revised again and replaced; the
    reference clears temp_dir and re-revises / overwrites every read, :196-201; SURVEY.md 5
    "skip-if-exists is a free resume"); --resume_keep_failed leaves those files alone and returns 3;
  * each GPU worker is pinned to the cores of its GPU's NUMA node (`worker_cpus`), and a very long
    read (BASELINE config 5) is split over the GPU workers by window range with a T-1-event halo
    (`shard.split_read_windows`, `plan_splits`): still no device-to-device exchange.
FASTQ qualities: the reviser has no Guppy qualities; each revised base gets
Phred = -10 log10(1 - min(p_model1, p_model2)) of its call (capped 1..40), unrevised edge bases '#'.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import h5lite
from . import hostlib
from . import hoststage as hs
from .shard import shard_reads, split_read_windows
from .weights import load_model

VERSION = "1.0"


def get_args(argv: Optional[Sequence[str]] = None):
    p = argparse.ArgumentParser(
        prog="NanoReviser.py", usage="%(prog)s [-d] [-o]",
        description="An Error-correction Tool for Nanopore Sequencing Based on a Deep Learning Algorithm "
                    "(MI355X engine)")
    p.add_argument("-d", "--fast5_base_dir", dest="fast5_base_dir", help="path to the fast5 files")
    p.add_argument("-o", "--output_dir", dest="output_dir", default="./unitest/nanorev_output/",
                   help="path to store the output files")
    p.add_argument("-F", "--output_format", dest="output_format", default="fasta",
                   help="format of the output files, default is fasta")
    p.add_argument("-S", "--species", dest="species", default="human", help="ecoli or human")
    p.add_argument("--thread", dest="thread", type=int, default=100,
                   help="host parsing threads per GPU worker (capped at the core count)")
    p.add_argument("-t", "--tmp_dir", dest="temp_dir", default="./unitest/tmp/", help="accepted, unused")
    p.add_argument("-e", "--failed_read", dest="failed_reads_filename", default="failed_reads.txt",
                   help="document to log the failed reads")
    p.add_argument("-g", "--basecall_group", dest="basecall_group", default="Basecall_1D_000")
    p.add_argument("-s", "--basecall_subgroup", dest="basecall_subgroup", default="BaseCalled_template")
    p.add_argument("--test_mode", action="store_true", default=False, help="just for unitest")
    p.add_argument("--model1_predict_dir", dest="model1_predict_dir", default=None)
    p.add_argument("--model2_predict_dir", dest="model2_predict_dir", default=None)
    p.add_argument("-v", "--virsion", action="store_true", dest="virsion", help="version of NanoReviser")
    p.add_argument("--gpus", type=int, default=0, help="GPUs to use (0 = all visible)")
    p.add_argument("--batch", type=int, default=4096, help="windows per device launch group")
    p.add_argument("--model_dir", default=None, help="root of model/<species>/ (default: next to the package)")
    p.add_argument("--resume", action="store_true", default=False,
                   help="skip every read whose <stem>_out.<format> already exists and is not empty (outputs are written "
                        "through a temporary + rename, so an existing file is a complete one) - except the reads the earlier "
                        "run listed in its failed-reads file: their output holds the ORIGINAL basecalls (a GPU worker died, "
                        "the engine failed), so they are revised again and the file is replaced")
    p.add_argument("--resume_keep_failed", action="store_true", default=False,
                   help="with --resume: do not redo the reads of the failed-reads file either; they stay listed there and "
                        "the exit code is 3 while any of them is left")
    p.add_argument("--split_reads_above", type=float, default=float(os.environ.get("NRV_SPLIT_READS_MB", "4")),
                   help="with more than one GPU: a fast5 file above this many MB (~10 MB per 100 k events) AND above a GPU "
                        "worker's fair share of the input has its window range split over the workers, T-1 events of halo "
                        "per slice (0 = never)")
    a = p.parse_args(argv)
    if a.virsion:
        print(f"The virsion of NanoReviser : {VERSION} ")
        raise SystemExit(0)
    if not (a.fast5_base_dir and a.output_dir):
        p.print_help()
        raise SystemExit(0)
    return a


def model_paths(args):
    """NanoReviser.py:188-194."""
    root = args.model_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "model")
    sp = "ecoli" if (args.test_mode and not args.species) else args.species
    p1 = args.model1_predict_dir or os.path.join(root, sp, f"{sp}_win13_50ep_model1.h5")
    p2 = args.model2_predict_dir or os.path.join(root, sp, f"{sp}_win13_50ep_model2.h5")
    ok = all(os.path.exists(p) or os.path.exists(os.path.splitext(p)[0] + ".f32") for p in (p1, p2))
    if not ok:
        raise RuntimeError("！！！[Error] model file: Please check the dir of models file!!")
    return p1, p2


def out_name(output_dir: str, fast5_fn: str, fmt: str) -> str:
    # NanoReviser.py:137/163: plain string concatenation with output_dir, stem = up to the first '.'
    return output_dir + fast5_fn.split(".")[0] + "_out." + fmt


def parse_read(path: str, group: str, subgroup: str):
    """fast5 -> (ReadData, fastq text or None).  nanorev_fast5_handeler.py:58-150."""
    d = h5lite.read_fast5(path, group, subgroup)
    ev = d["events"]
    start, length = ev["start"], ev["length"]
    try:
        old = tuple(int(x) for x in d["version"].split(".")[:2]) <= (0, 0)
    except ValueError:
        old = False
    if old:                                   # pre-versioned Albacore: times in seconds (:66-71)
        start = start * 4000 - d["raw_attrs"]["start_time"]
        length = length * 4000
    rd = hs.collapse_events(start, ev["mean"], ev["stdv"], ev["model_state"], ev["move"], d["signal"])
    fq = d["fastq"]                           # bytes (fixed-length string dataset) or str (variable-length)
    if isinstance(fq, (bytes, np.bytes_)):
        fq = bytes(fq).decode("utf8", "replace")
    elif fq is not None:
        fq = str(fq)
    return rd, fq


def phred_chars(p1: np.ndarray, p2: np.ndarray, a1=None, a2=None) -> np.ndarray:
    """Per-window quality from the smaller of the two models' top probabilities.  With the argmax
    arrays the top probability is a gather (p.max(-1) over a 6-wide axis is 10x slower in NumPy)."""
    if a1 is not None and a2 is not None and len(a1) == len(p1):
        idx = np.arange(len(a1))
        conf = np.minimum(p1[idx, a1], p2[idx, a2]).astype(np.float64)
    else:
        conf = np.minimum(p1.max(-1), p2.max(-1)).astype(np.float64)
    q = np.clip(np.round(-10.0 * np.log10(np.maximum(1.0 - conf, 1e-4))), 1, 40).astype(np.int64)
    return (q + 33).astype(np.uint8)


def _finish_read(T: int, rt, p1, p2, a1, a2, want_qual: bool = True):
    """Calls of one read -> (revised sequence, per-base quality string or None)."""
    codes = np.frombuffer(np.asarray(rt.bases, dtype="S1").tobytes(), dtype=np.uint8)
    off, n = (T - 1) // 2, len(a1)
    if n == 0:
        return codes.tobytes().decode("ascii"), "#" * len(codes)
    first, second, count = hs.merge_calls(codes[off:off + n], a1, a2)
    if not want_qual:                                # FASTA output: no quality string to build
        z = np.zeros(n, np.uint8)
        seq_mid, _ = hs.expand_calls(first, second, count, z, z)
        seq = codes[:off].tobytes() + seq_mid.tobytes() + codes[off + n:].tobytes()
        return seq.decode("ascii"), None
    qc = phred_chars(p1, p2, a1, a2)
    seq_mid, q_mid = hs.expand_calls(first, second, count, qc, qc)
    edge = np.full(1, ord("#"), np.uint8)
    seq = codes[:off].tobytes() + seq_mid.tobytes() + codes[off + n:].tobytes()
    qual = np.repeat(edge, off).tobytes() + q_mid.tobytes() + np.repeat(edge, len(codes) - off - n).tobytes()
    return seq.decode("ascii"), qual.decode("ascii")


def predict_one(reviser, rt):
    """Device side of one read -> (p1, p2, a1, a2) over its N-T windows."""
    if isinstance(rt, hs.RawReadTensors):
        return reviser.predict_reads_raw([rt.raw], [rt.starts], [rt.feat_ev], [rt.shift], [rt.scale])
    return reviser.predict_read(rt.sig_ev, rt.feat_ev)


def prepare_many(reviser_cls, rts, T):
    """The NumPy half of `predict_many` for raw reads (concatenation, descriptors, output arrays), done outside the
    engine thread; None when the batch cannot take the packed path (mixed tensor kinds, an engine without it)."""
    pack = getattr(reviser_cls, "pack_reads_raw", None)
    if pack is None or not all(isinstance(rt, hs.RawReadTensors) for rt in rts):
        return None
    return pack([rt.raw for rt in rts], [rt.starts for rt in rts], [rt.feat_ev for rt in rts],
                [rt.shift for rt in rts], [rt.scale for rt in rts], T)


def predict_many(reviser, rts, packed=None):
    """Several reads in ONE device call: their per-event arrays are concatenated, the engine forms
    every sliding window of the concatenation, and the T windows that straddle each read boundary
    are simply not used (0.2 % extra work; full launch groups, one host<->device round trip).
    Returns one (p1, p2, a1, a2) per read."""
    if len(rts) == 1 and packed is None:
        return [predict_one(reviser, rts[0])]
    T = reviser.T
    if packed is not None:
        p1, p2, a1, a2 = reviser.run_packed_raw(packed)
    elif all(isinstance(rt, hs.RawReadTensors) for rt in rts):
        # raw samples + event starts cross PCIe; the (N,50) windows are cut on the device
        p1, p2, a1, a2 = reviser.predict_reads_raw([rt.raw for rt in rts], [rt.starts for rt in rts],
                                                   [rt.feat_ev for rt in rts], [rt.shift for rt in rts],
                                                   [rt.scale for rt in rts])
    else:
        sig = np.concatenate([rt.sig_ev for rt in rts])
        feat = np.concatenate([rt.feat_ev for rt in rts])
        p1, p2, a1, a2 = reviser.predict_read(sig, feat)
    out, e0 = [], 0
    for rt in rts:
        N = len(rt.feat_ev)
        n = max(N - T, 0)
        sl = slice(e0, e0 + n)                     # window i of the read == window e0+i of the batch
        out.append((p1[sl], p2[sl], a1[sl], a2[sl]))
        e0 += N
    return out


def revise_one(reviser, rt, want_qual: bool = True):
    """One read through the engine -> (revised sequence, per-base quality string)."""
    return _finish_read(reviser.T, rt, *predict_one(reviser, rt), want_qual=want_qual)


def revise_many(reviser, rts, want_qual: bool = True):
    return [_finish_read(reviser.T, rt, *c, want_qual=want_qual) for rt, c in zip(rts, predict_many(reviser, rts))]


def _load_one(job, native: bool = True):
    """Worker side of the host stage: fast5 -> per-event device inputs (picklable).  The file goes through the native
    reader (libnanorev_host.so: nrvh_load_fast5, one C call with the GIL released) when that is built and knows the
    file's HDF5 subset; anything else - and every failure - through the Python host stage, which is the definition of
    the numbers and words the errors."""
    path, fn, group, subgroup = job[:4]
    t0 = time.perf_counter()
    if native:
        rc, o = hostlib.load_fast5(path, group, subgroup, want_fastq=(job[4] if len(job) > 4 else True))
        if rc == hostlib.OK:
            rt = hs.RawReadTensors(o["raw"], o["starts"], o["feat"], o["bases"], o["shift"], o["scale"])
            return fn, rt, o["fastq"], None, time.perf_counter() - t0
    try:
        rd, fq = parse_read(path, group, subgroup)
    except Exception as e:                           # broken file: nothing to fall back to
        return fn, None, None, repr(e), time.perf_counter() - t0
    try:
        # int16 samples + starts + features (46 B/base through the pipe and over PCIe); the signal
        # windows are cut on the device.  Non-int16 signals (never seen in fast5) take the host path.
        rt = hs.read_tensors_raw(rd) if np.asarray(rd.signal).dtype == np.int16 else hs.read_tensors(rd)
        return fn, rt, fq, None, time.perf_counter() - t0
    except Exception as e:                           # host stage failed: originals can still be written
        return fn, hs.ReadTensors(None, None, rd.bases, 0.0, 0.0), fq, repr(e), time.perf_counter() - t0


class _LightRead:
    """What the finisher needs of a read whose arrays travel inside a bundle: the original bases."""
    __slots__ = ("bases", "n_ev")

    def __init__(self, bases, n_ev):
        self.bases, self.n_ev = bases, n_ev


def _load_bundle(jobs):
    """Worker-process side of the host stage for SEVERAL files: every read through `_load_one`, and the raw reads
    among them concatenated here into the arrays of ONE device call (samples, event starts, event features, one row
    of (raw_len, ev_len, shift, scale) per read), so that the main process neither unpickles thousands of small
    arrays nor concatenates them: a bundle is four big arrays.  Returns (entries, bundle): entries are
    `_load_one`'s tuples, with the tensors of bundled reads replaced by a `_LightRead`; bundle is None when no read
    qualified."""
    t0 = time.perf_counter()
    nb = hostlib.load_bundle([j[0] for j in jobs], jobs[0][2], jobs[0][3], want_fastq=(jobs[0][4] if len(jobs[0]) > 4 else True)) \
        if jobs and all(j[2:] == jobs[0][2:] for j in jobs) else None
    if nb is not None and int((nb["status"] == hostlib.OK).sum()) > 0:
        # ONE native call for the whole task: the reads it took are already concatenated; the others (a layout the native
        # reader does not know, a broken file) go through the Python host stage one by one
        dt = (time.perf_counter() - t0) / len(jobs)
        entries, good, eo = [], [], 0
        for i, j in enumerate(jobs):
            if nb["status"][i] == hostlib.OK:
                el = int(nb["meta"][i, 1])
                entries.append((j[1], _LightRead(nb["bases"][eo:eo + el], el), nb["fastq"][i], None, dt))
                good.append(i)
                eo += el
            else:
                entries.append(tuple(_load_one(j, native=False)))
        if any(isinstance(e[1], hs.RawReadTensors) for e in entries):
            # a Python-path read would have to be spliced into the middle of the concatenation: rare enough to give up
            # the bundle's shortcut for this task (each read then travels on its own, as without a pool)
            out, eo, ro = [], 0, 0
            for i, e in enumerate(entries):
                if i in good:
                    rl, el, sh, sc = nb["meta"][i]
                    rl, el = int(rl), int(el)
                    rt = hs.RawReadTensors(nb["raw"][ro:ro + rl], nb["starts"][eo:eo + el], nb["feat"][eo:eo + el],
                                           nb["bases"][eo:eo + el], float(sh), float(sc))
                    out.append((e[0], rt, e[2], None, e[4]))
                    ro += rl
                    eo += el
                else:
                    out.append(e)
            return _bundle_of(out)
        bundle = {"idx": good, "raw": nb["raw"], "starts": nb["starts"], "feat": nb["feat"],
                  "meta": np.ascontiguousarray(nb["meta"][good]), "bases": nb["bases"]}
        return entries, bundle
    return _bundle_of([_load_one(j) for j in jobs])


def _bundle_of(loaded):
    """`_load_one` tuples -> (entries, bundle): the raw reads among them concatenated into the arrays of one device call."""
    entries, good = [], []
    for fn, rt, fq, err, dt in loaded:
        if err is None and isinstance(rt, hs.RawReadTensors):
            good.append(len(entries))
        entries.append([fn, rt, fq, err, dt])
    if not good:
        return [tuple(e) for e in entries], None
    rts = [entries[i][1] for i in good]
    bundle = {
        "idx": good,
        "raw": np.concatenate([np.ascontiguousarray(r.raw, dtype=np.int16) for r in rts]),
        "starts": np.concatenate([np.ascontiguousarray(r.starts, dtype=np.int32) for r in rts]),
        "feat": np.concatenate([np.asarray(r.feat_ev, np.float32).reshape(-1, 6) for r in rts]),
        "meta": np.array([[len(r.raw), len(r.starts), r.shift, r.scale] for r in rts], np.float64),
    }
    for i, r in zip(good, rts):
        entries[i][1] = _LightRead(r.bases, len(r.starts))
    return [tuple(e) for e in entries], bundle


def _bundle_reads(bundle):
    """The reads of a bundle as RawReadTensors VIEWS (per-read retries after a failed batched call)."""
    out, ro, eo = [], 0, 0
    for rl, el, sh, sc in bundle["meta"]:
        rl, el = int(rl), int(el)
        out.append(hs.RawReadTensors(bundle["raw"][ro:ro + rl], bundle["starts"][eo:eo + el],
                                     bundle["feat"][eo:eo + el], None, float(sh), float(sc)))
        ro += rl
        eo += el
    return out


class _OutSpec:
    """The part of the parsed arguments `write_read` needs (picklable, for the worker processes)."""

    def __init__(self, args):
        self.output_dir, self.output_format = args.output_dir, args.output_format


def _finish_in_worker(spec, T, fn, bases, a1, a2, qc):
    """Worker-process side of the merge: calls of one read + its original bases -> revised read -> its output file
    (the 0.5 ms of NumPy, string building and file IO per read that the main process' finisher thread used to spend
    under the GIL).  qc: per-window Phred characters (FASTQ) or None (FASTA).  Returns (bases written, error)."""
    try:
        codes = np.frombuffer(np.asarray(bases, dtype="S1").tobytes(), dtype=np.uint8)
        off, n = (T - 1) // 2, len(a1)
        if n == 0:
            seq, qual = codes.tobytes().decode("ascii"), "#" * len(codes)
        else:
            first, second, count = hs.merge_calls(codes[off:off + n], a1, a2)
            z = qc if qc is not None else np.zeros(n, np.uint8)
            seq_mid, q_mid = hs.expand_calls(first, second, count, z, z)
            seq = (codes[:off].tobytes() + seq_mid.tobytes() + codes[off + n:].tobytes()).decode("ascii")
            qual = None
            if qc is not None:
                edge = np.full(1, ord("#"), np.uint8)
                qual = (np.repeat(edge, off).tobytes() + q_mid.tobytes()
                        + np.repeat(edge, len(codes) - off - n).tobytes()).decode("ascii")
        write_read(spec, fn, seq, qual)
        return len(seq), None
    except Exception as e:
        return 0, repr(e)


def _finish_native(spec, T, fn, bases, a1, a2, qc):
    """`_finish_in_worker` through libnanorev_host.so (nrvh_finish_read: merge, record, temporary + rename in one C call
    with the GIL released) - for the THREAD pool of process_files.  Same return value."""
    try:
        os.makedirs(spec.output_dir, exist_ok=True)
        nb = hostlib.finish_read(bases, a1, a2, T, qc, fn.split("/")[-1].replace(" ", "|||"),
                                 out_name(spec.output_dir, fn, spec.output_format), spec.output_format == "fastq")
        if nb is None:
            return _finish_in_worker(spec, T, fn, bases, a1, a2, qc)
        return nb, None
    except Exception as e:
        return 0, repr(e)


def _finish_bundle_native(spec, T, fns, bases, ev_len, a1, a2, qc):
    """All reads of one device call through nrvh_finish_bundle (one C call, GIL released): [(bases written, error)]."""
    try:
        os.makedirs(spec.output_dir, exist_ok=True)
        r = hostlib.finish_bundle(bases, ev_len, a1, a2, T, qc, [fn.split("/")[-1].replace(" ", "|||") for fn in fns],
                                  [out_name(spec.output_dir, fn, spec.output_format) for fn in fns], spec.output_format == "fastq")
        if r is None:
            raise RuntimeError("libnanorev_host.so went away")
        nw, st = r
        return [(int(n), None if c == hostlib.OK else f"native finisher: error {int(c)}") for n, c in zip(nw, st)]
    except Exception as e:
        return [(0, repr(e))] * len(fns)


def usable_cores() -> int:
    """Cores this process can really use: its affinity mask, cut down to the cgroup's CPU quota (a container that shows
    256 CPUs and grants 16 runs sixteen parser workers, not the 32 that --thread's default of 100 would otherwise give:
    r03, 4000 reads: 8 workers 11.1, 16 workers 10.4 M bases/s end to end)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    return max(1, cores)


def write_read(args, fast5_fn: str, seq: str, qual: Optional[str]):
    os.makedirs(args.output_dir, exist_ok=True)
    if args.output_format == "fastq":
        text = hs.fastq_record(fast5_fn, list(seq), list(qual if qual is not None else "#" * len(seq)))
    else:
        text = hs.fasta_record(fast5_fn, list(seq))
    # atomic: a worker killed mid-write leaves a stray temporary, never a truncated <read>_out file
    dst = out_name(args.output_dir, fast5_fn, args.output_format)
    tmp = f"{dst}.tmp{os.getpid()}"
    with open(tmp, "w") as fp:
        fp.write(text)
    os.replace(tmp, dst)


kNativePoolMax = 4      # parser THREADS per GPU worker with the native host stage: 4 deliver 23 M bases/s (r04, 16-core cgroup),
                        # twice what one GPU takes; more only take the GIL from the engine thread (r04, 4000 reads end to
                        # end: 2 / 4 / 8 / 16 threads = 11.2 / 11.3 / 10.2 / 9.3 M bases/s).  NRV_PARSER_THREADS_MAX overrides
                        # it (INTEGRATION.md "Process-wide settings"); the cap was tuned at kSwitchInterval.
kSwitchInterval = 2e-4  # CPython switch interval while the parser threads run (see process_files)


def native_pool_max() -> int:
    try:
        return max(1, int(os.environ.get("NRV_PARSER_THREADS_MAX", kNativePoolMax)))
    except ValueError:
        return kNativePoolMax


def parser_pool_size(threads: int, cores: int, gpu_workers: int, n_files: int, native: bool = False) -> int:
    """Parser workers of ONE GPU worker: --thread, capped at this worker's share of the cores the process may use
    (cores // gpu_workers, at least 1), at 32 (processes on the Python host stage) or kNativePoolMax (threads on the
    native one) and at the number of files."""
    share = max(1, cores // max(1, gpu_workers))
    return max(1, min(int(threads), share, native_pool_max() if native else 32, max(1, n_files)))


def process_files(args, files: Sequence[str], reviser, log: Callable[[str], None],
                  on_file: Optional[Callable[[str, bool], None]] = None, gpu_workers: int = 1,
                  core_share: Optional[int] = None) -> dict:
    """Revise `files` (names inside args.fast5_base_dir) with one engine.  The host stage (HDF5
    parsing, event collapse, signal segmentation) runs in worker PROCESSES (--thread of them, capped
    at the core count) that stay a bounded number of reads ahead of the device.
    on_file(fn, revised): called once per file when its output is final (revised, or the original
    basecalls after a failure; revised=False also for files that could not be parsed at all).
    gpu_workers: how many such calls run on this host at the same time (one per GPU, `run_workers`): the cores are
    SHARED, so each call's parser pool gets usable_cores() // gpu_workers of them - eight workers on a 16-core cgroup
    start 8 x 2 parser processes, not 8 x 16.  core_share: this call's cores, when the caller has already
    divided them (a GPU worker pinned to its NUMA slice: `_worker`)."""
    stats = {"reads": 0, "bases": 0, "failed": [], "host_s": 0.0, "engine_s": 0.0}
    # NRV_CLI_TRACE=1: where the wall time of this call goes (first device call, engine idle gaps, tail), to the log
    trace, t_start = ([] if os.environ.get("NRV_CLI_TRACE") else None), time.perf_counter()
    if trace is not None:
        try:                                          # age of this process: interpreter start, imports, main() so far
            with open("/proc/self/stat") as f:
                ticks = int(f.read().rsplit(")", 1)[1].split()[19])
            with open("/proc/uptime") as f:
                trace.append(("mark", 0.0, f"process age {float(f.read().split()[0]) - ticks / os.sysconf('SC_CLK_TCK'):.2f} s at"))
        except Exception:
            pass

    def mark(what):
        if trace is not None:
            trace.append(("mark", time.perf_counter() - t_start, what))
    note = on_file or (lambda fn, ok: None)
    # `reviser` may be a zero-argument factory: the engine is then created by the engine thread as its first task,
    # i.e. WHILE the parser pool starts and the first reads are parsed (nrv_create: HIP context + weight packing,
    # 0.15-0.3 s that used to precede everything else)
    lazy = callable(reviser) and not hasattr(reviser, "predict_read")
    box = {"rv": None if lazy else reviser}
    # NRV_CLI_ENGINES=n: n engines on the same device, each driven by its own thread, so that the pipeline fill / drain
    # of one engine's device call overlaps the other's kernels.  Measured (r03, 4000 reads, 16 launch groups per call):
    # 1 engine 11.0, 2 engines 10.1-10.4, 3 engines 10.0 M bases/s end to end - the device is already the limit and the
    # two engines' persistent kernels get in each other's way - so the default stays ONE; the knob is kept, tested.
    import queue
    import threading
    n_eng = max(1, int(os.environ.get("NRV_CLI_ENGINES", "1"))) if lazy else 1
    free = queue.Queue()                              # engines not inside a call
    engines = []
    if not lazy:
        free.put(reviser)
        engines.append(reviser)
    stats_lock = threading.Lock()
    native_threads = hostlib.load() is not None and os.environ.get("NRV_HOST_THREADS", "1") != "0"
    nworkers = parser_pool_size(int(args.thread), usable_cores(), gpu_workers, len(files), native=native_threads) \
        if core_share is None else parser_pool_size(int(args.thread), int(core_share), 1, len(files), native=native_threads)
    stats["parser_workers"] = nworkers
    jobs = [(os.path.join(args.fast5_base_dir, fn), fn, args.basecall_group, args.basecall_subgroup,
             args.output_format == "fastq") for fn in files]

    # Files per worker task: one task = one device call (>= kBatchEvents events at ~6.5 k events per read); the
    # worker hands back the reads of a task already concatenated (`_load_bundle`).
    # 16 launch groups per device call (r03, 4000 reads: 8 -> 7.6, 16 -> 10.2, 32 -> 10.3 M bases/s end to end: a call's
    # pipeline fill / drain is paid once per call), but never so many reads per task that the workers run dry
    kBatchEvents = int(os.environ.get("NRV_CLI_GROUPS", "16")) * max(int(getattr(args, "batch", 4096)), 1024)
    per_task = max(1, min(32, kBatchEvents // 6500, -(-len(jobs) // (2 * nworkers))))

    pool = None
    # With the native host stage (libnanorev_host.so) a task is ONE C call per bundle of files and one per finished read,
    # both with the GIL released: the pool is THREADS of this process - nothing is spawned, nothing is pickled, a bundle's
    # arrays are handed over by reference.  Without the library (or NRV_HOST_THREADS=0) the same tasks run in worker
    # PROCESSES on the Python host stage, as in round 3.
    stats["host_stage"] = "native, threads" if native_threads else ("python, processes" if hostlib.load() is None else "native, processes")
    old_switch = None
    if nworkers > 1 and len(files) >= 4 and native_threads:
        from concurrent.futures import ThreadPoolExecutor as _TPE
        pool = _TPE(nworkers)
        # A pool thread coming back from its C call needs the GIL for ~0.1 ms of bookkeeping; with CPython's default
        # switch interval (5 ms) it waited that long behind whichever thread held it, and four parser threads delivered
        # what one does (r04: 3.2 M bases/s with 1, 4 or 8 threads; 10 / 13 M with 4 / 8 at 0.2 ms).  Restored on return.
        # A process-global setting: only touched when the embedding caller has left it at CPython's default (a caller
        # that tuned it keeps its value), NRV_SWITCH_INTERVAL=0 leaves it alone altogether, and it is restored on return.
        try:
            want = float(os.environ.get("NRV_SWITCH_INTERVAL", kSwitchInterval))
        except ValueError:
            want = kSwitchInterval
        if want > 0 and abs(sys.getswitchinterval() - 0.005) < 1e-9:
            old_switch = sys.getswitchinterval()
            sys.setswitchinterval(want)
    elif nworkers > 1 and len(files) >= 4:
        import multiprocessing as mp
        # The workers never call BLAS; without a cap every one of them starts, at `import numpy`, an OpenBLAS pool sized
        # for the machine (256 CPUs on the GPU boxes, 16 in the cgroup): the first parsed reads came back after 0.73 s
        # instead of 0.15 s (r03, scripts/gpu_cli_trace.sh).  Inherited through the environment of the spawned children.
        for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ.setdefault(v, "1")
        pool = ProcessPoolExecutor(nworkers, mp_context=mp.get_context("spawn"))

    def results():
        """Yields (entries, bundle): `_load_one` tuples, and for pooled tasks their pre-concatenated raw reads."""
        if pool is None:
            for j in jobs:
                yield [_load_one(j)], None
            return
        from collections import deque
        pend = deque()
        for k in range(0, len(jobs), per_task):
            pend.append(pool.submit(_load_bundle, jobs[k:k + per_task]))
            if k == 0:
                mark("first task submitted (workers started)")
            if len(pend) >= 3 * nworkers:
                if len(pend) == 3 * nworkers and k == (3 * nworkers - 1) * per_task:
                    mark("3 x workers tasks submitted")
                yield pend.popleft().result()
                if k == (3 * nworkers - 1) * per_task:
                    mark("first bundle back")
        while pend:
            yield pend.popleft().result()

    def fallback(fn, rt, fq, e):                      # NanoReviser.py:146-152 / :173-179
        stats["failed"].append(fn)
        log(f"[！！！Error] revising {fn.split('.')[0]}: {e}; writing the original basecalls")
        try:
            if args.output_format == "fastq" and fq is not None:
                b, q = hs.trim_fastq(fq)
                write_read(args, fn, b, q)
            else:
                orig = "".join(x.decode() for x in np.asarray(rt.bases).tolist())
                write_read(args, fn, orig, None)
        except Exception as e2:
            log(f"[！！！Error] stroring : {fn.split('.')[0]}_out.{args.output_format}...... {e2}")
        note(fn, False)

    want_qual = args.output_format == "fastq"

    spec = _OutSpec(args)
    finishing = []                                    # (future, fn, rt, fq) of merges handed to the pool

    def finished(fn, nb):
        stats["bases"] += nb
        if not args.test_mode:
            log(f"[p:::] {fn.split('.')[0]}_out.{args.output_format} was saved......")
        else:
            log("INFO Congratulations, NanoReviser is installed properly")
        note(fn, True)

    def finish_batch(batch, calls):
        """Finisher thread.  With a worker pool the merge + file write of each read is a pool task (the calls and
        the original bases are ~20 KB per read); without one it happens here."""
        reviser = box["rv"]
        for (fn, rt, fq), c in zip(batch, calls):
            if isinstance(c, Exception):
                fallback(fn, rt, fq, c)
                continue
            try:
                if pool is not None:
                    p1, p2, a1, a2 = c
                    qc = phred_chars(p1, p2, a1, a2) if want_qual and len(a1) else None
                    finishing.append((pool.submit(_finish_native if native_threads else _finish_in_worker, spec, reviser.T, fn,
                                                  np.asarray(rt.bases), np.asarray(a1), np.asarray(a2), qc), fn, rt, fq))
                    continue
                seq, qual = _finish_read(reviser.T, rt, *c, want_qual=want_qual)
                write_read(args, fn, seq, qual)
                finished(fn, len(seq))
            except Exception as e:
                fallback(fn, rt, fq, e)

    def collect_finished(block):
        """Results of the pooled merges (main thread).  An entry is one read, or (fn is a list) all reads of a device call."""
        while finishing and (block or finishing[0][0].done()):
            fut, fn, rt, fq = finishing.pop(0)
            many = isinstance(fn, list)
            try:
                res = fut.result()
            except Exception as e:                    # the pool broke (a parser process died)
                res = [(0, repr(e))] * len(fn) if many else (0, repr(e))
            for (nb, err), f1, r1, q1 in (zip(res, fn, rt, fq) if many else [(res, fn, rt, fq)]):
                if err is None:
                    finished(f1, nb)
                else:
                    fallback(f1, r1, q1, err)

    def finish_bundle(batch, bundle, outs):
        """Finisher thread, native host stage: merge + record + file of ALL reads of a device call as one pool task."""
        reviser = box["rv"]
        try:
            p1, p2, a1, a2 = outs
            qc = phred_chars(p1, p2, a1, a2) if want_qual and len(a1) else None
            fut = pool.submit(_finish_bundle_native, spec, reviser.T, [fn for fn, _, _ in batch], bundle["bases"],
                              bundle["meta"][:, 1].astype(np.int64), a1, a2, qc)
            finishing.append((fut, [fn for fn, _, _ in batch], [rt for _, rt, _ in batch], [fq for _, _, fq in batch]))
        except Exception as e:
            for fn, rt, fq in batch:
                fallback(fn, rt, fq, e)

    def run_batch(batch, packed=None, bundle=None):
        """Engine thread: one device call for the batch; merging and writing go to the finisher."""
        reviser = free.get()                          # an engine that is not inside a call (all have the same T, weights)
        try:
            return run_batch_on(reviser, batch, packed, bundle)
        finally:
            free.put(reviser)

    # Device calls PIPELINED two deep (r06; one engine, native bundles; NRV_CLI_PIPELINE=0: one call at a time as before): the
    # engine thread enqueues bundle k+1 (nrv_reads_raw_begin: inputs copied, every stage enqueued) BEFORE it collects bundle k
    # (nrv_reads_raw_end), so the device never waits for a call's fill, drain or the Python between two calls.
    pipelined = n_eng == 1 and os.environ.get("NRV_CLI_PIPELINE", "1") != "0"
    pending = {}                                      # id(engine) -> the call in flight on it: (ticket, batch, bundle, trace index, t0)

    def _done():
        from concurrent.futures import Future
        f = Future()
        f.set_result(None)
        return f

    def collect(reviser, item):
        """Second half of a pipelined call: wait for it, hand its reads to the finisher; a failure is isolated read by read
        on the synchronous path, exactly as for an unpipelined call."""
        tk, batch, bundle, ti, t_begin = item
        t0 = time.perf_counter()
        try:
            outs = reviser.end_packed_raw(tk)
        except Exception:
            # the per-read retries are other entry points of the handle: first collect the call that was enqueued behind this one
            nxt = pending.pop(id(reviser), None)
            if nxt is not None:
                collect(reviser, nxt)
            calls = []
            for rt in _bundle_reads(bundle):
                try:
                    calls.append(predict_one(reviser, rt))
                except Exception as e:
                    calls.append(e)
            with stats_lock:
                stats["engine_s"] += time.perf_counter() - t0
            return fin.submit(finish_batch, batch, calls)         # (the finisher is one thread: behind nxt's hand-over)
        with stats_lock:
            stats["engine_s"] += time.perf_counter() - t0
        if ti is not None:
            trace[ti] = ("call", t_begin - t_start, time.perf_counter() - t_begin)
        return fin.submit(finish_bundle, batch, bundle, outs)

    def flush_pipeline():
        """Engine thread, behind the last batch: whatever is still in flight."""
        last = _done()
        for rv in list(engines):
            item = pending.pop(id(rv), None)
            if item is not None:
                last = collect(rv, item)
        return last

    def run_batch_on(reviser, batch, packed, bundle):
        t0 = time.perf_counter()
        ti = None
        if trace is not None:
            ti = len(trace)
            trace.append(("call", t0 - t_start, 0.0))
        rts = None
        prev = pending.pop(id(reviser), None)
        if (pipelined and bundle is not None and packed is not None and native_threads and pool is not None
                and hasattr(reviser, "begin_packed_raw") and "bases" in bundle
                and len(bundle["bases"]) == int(bundle["meta"][:, 1].sum())):
            try:
                tk = reviser.begin_packed_raw(packed)
            except Exception:
                tk = None                             # e.g. a handle that could not grow its buffers: the synchronous path decides
            if tk is not None:
                pending[id(reviser)] = (tk, batch, bundle, ti, t0)
                with stats_lock:
                    stats["engine_s"] += time.perf_counter() - t0
                return collect(reviser, prev) if prev is not None else _done()
        if prev is not None:                          # between a call's two halves nothing else may run on its handle
            collect(reviser, prev)

        def reads():                                  # the reads as tensors: only the per-read paths need them
            return _bundle_reads(bundle) if bundle is not None else [rt for _, rt, _ in batch]
        try:
            if bundle is not None and packed is not None:
                p1, p2, a1, a2 = reviser.run_packed_raw(packed)
                if native_threads and pool is not None and "bases" in bundle and len(bundle["bases"]) == int(bundle["meta"][:, 1].sum()):
                    with stats_lock:
                        stats["engine_s"] += time.perf_counter() - t0
                    if ti is not None:
                        trace[ti] = ("call", t0 - t_start, time.perf_counter() - t0)
                    return fin.submit(finish_bundle, batch, bundle, (p1, p2, a1, a2))
                calls, e0, T = [], 0, reviser.T
                for rl, el, _, _ in bundle["meta"]:        # window i of a read == window e0 + i of the bundle
                    n = max(int(el) - T, 0)
                    calls.append((p1[e0:e0 + n], p2[e0:e0 + n], a1[e0:e0 + n], a2[e0:e0 + n]))
                    e0 += int(el)
            else:
                rts = reads()
                calls = predict_many(reviser, rts, packed)
        except Exception:
            calls = []
            for rt in (rts if rts is not None else reads()):   # isolate the failing read(s)
                try:
                    calls.append(predict_one(reviser, rt))
                except Exception as e:
                    calls.append(e)
        with stats_lock:
            stats["engine_s"] += time.perf_counter() - t0
        if ti is not None:
            trace[ti] = ("call", t0 - t_start, time.perf_counter() - t0)
        return fin.submit(finish_batch, batch, calls)

    # reads are grouped into device calls of >= kBatchEvents events; the engine runs in its own
    # thread (the C-ABI call releases the GIL) so unpickling the next reads overlaps the device
    from concurrent.futures import ThreadPoolExecutor
    from collections import deque
    inflight = deque()
    import contextlib
    with contextlib.ExitStack() as stack:
        if old_switch is not None:
            stack.callback(lambda: sys.setswitchinterval(old_switch))
        if pool is not None:                          # whatever happens below, the parser pool does not outlive us
            stack.callback(lambda: pool.shutdown(wait=False, cancel_futures=True))
        eng = stack.enter_context(ThreadPoolExecutor(n_eng))
        fin = stack.enter_context(ThreadPoolExecutor(1))

        def create(first):
            try:
                rv = reviser()
            except BaseException as e:
                if first:
                    raise                             # no engine at all: loud (submit() re-raises it)
                log(f"[s:::] a second engine on this device could not be created ({e!r}): continuing with one")
                return
            if first:
                box["rv"] = rv
            engines.append(rv)
            free.put(rv)
            mark("engine created")

        def create_next():
            try:
                created.result()                      # one at a time, the first one first
            except BaseException:
                return
            create(False)
        created = eng.submit(create, True) if lazy else None
        for _ in range(n_eng - 1):
            eng.submit(create_next)

        def submit(batch, bundle=None):
            # the NumPy half of the device call (descriptors, output arrays; for unbundled reads also the concatenation)
            # happens HERE, in the main thread, while the engine thread is inside the previous batch's call
            if created is not None:
                created.result()                      # raises what the factory raised: loud, no fallback engine
            rv = box["rv"]
            packed = None
            try:
                if bundle is not None and hasattr(rv, "run_packed_raw"):
                    packed = type(rv).pack_bundle(bundle["raw"], bundle["starts"], bundle["feat"], bundle["meta"], rv.T)
                elif bundle is None and len(batch) > 1:
                    packed = prepare_many(type(rv), [rt for _, rt, _ in batch], rv.T)
            except Exception:
                packed = None
            return eng.submit(run_batch, batch, packed, bundle)

        def drain(limit):
            while len(inflight) > limit:
                inflight.popleft().result().result()

        batch, nev = [], 0
        for entries, bundle in results():
            bundled = []
            for k, (fn, rt, fq, err, dt) in enumerate(entries):
                stats["host_s"] += dt
                stats["reads"] += 1
                if rt is None:
                    log(f"！！！[Error] fast5 file: {fn.split('.')[0]} {err}")
                    stats["failed"].append(fn)
                    note(fn, False)
                    continue
                if err is not None:
                    fallback(fn, rt, fq, err)
                    continue
                if bundle is not None and k in bundle["idx"]:
                    bundled.append((fn, rt, fq))
                    continue
                batch.append((fn, rt, fq))           # unbundled reads (sequential mode, non-int16 signals)
                nev += len(rt.feat_ev)
                if nev >= kBatchEvents:
                    inflight.append(submit(batch))
                    batch, nev = [], 0
                    drain(n_eng + 1)
            if bundled:
                inflight.append(submit(bundled, bundle))
                drain(n_eng + 1)
            collect_finished(False)
        if batch:
            inflight.append(submit(batch))
        if pipelined:
            inflight.append(eng.submit(flush_pipeline))
        drain(0)
        if created is not None:
            created.result()
        collect_finished(True)
        if pool is not None:                          # all work is in: an orderly end of the workers (their queues' semaphores
            pool.shutdown(wait=True)                  # are released here, not left to the resource tracker)
    calls = sorted((c for c in trace if c[0] == "call"), key=lambda c: c[1]) if trace else []
    if trace and not calls:                           # every file failed before a device call: only the marks
        log("[trace] " + "; ".join(f"{w} {t:.3f} s" for k, t, w in trace if k == "mark") + "; no device call")
    if calls:
        t_end = time.perf_counter() - t_start
        busy, idle, end = sum(c[2] for c in calls), 0.0, calls[0][1]
        for _, t, d in calls:                         # time with NO engine inside a call, between the first and the last call
            idle += max(0.0, t - end)
            end = max(end, t + d)
        log("[trace] " + "; ".join(f"{w} {t:.3f} s" for k, t, w in trace if k == "mark"))
        log(f"[trace] process_files {t_end:.3f} s: first device call at {calls[0][1]:.3f} s, {len(calls)} calls on "
            f"{len(engines)} engine(s), {busy:.3f} s summed, no call running for {idle:.3f} s, "
            f"tail after the last call {t_end - end:.3f} s; first call {calls[0][2]:.3f} s, "
            f"median call {sorted(c[2] for c in calls)[len(calls) // 2]:.4f} s")
    if any(callable(getattr(rv, "saturated", None)) for rv in engines):   # f16x2 range guard: stages re-run on the f32 kernels
        stats["range_reruns"] = sum(int(rv.saturated()[1]) for rv in engines if callable(getattr(rv, "saturated", None)))
        if stats["range_reruns"]:
            log(f"[s:::] {stats['range_reruns']} device stage(s) held out-of-range signal (spikes / tiny MAD) "
                "and were computed on the f32 kernels")
    return stats


def _default_factory(args, device: int):
    from .engine import Reviser
    p1, p2 = model_paths(args)
    return Reviser(load_model(p1), load_model(p2), device=device, batch=args.batch)


def _cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist format)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_local_cpus(device: int, sysfs: str = "/sys") -> Optional[List[int]]:
    """CPUs of the NUMA node HIP device `device` hangs on, or None when that cannot be told.  No HIP call: the KFD topology
    lists the GPU nodes (simd_count > 0) in the order the runtime enumerates them; *_VISIBLE_DEVICES given as a plain
    list of integers is honoured, any other form gives None."""
    try:
        vis = None
        for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            if os.environ.get(v, "") != "":
                if vis is not None:
                    return None                        # two layers of re-mapping: not guessed
                vis = [int(x) for x in os.environ[v].split(",")]
        if vis is not None:
            device = vis[device]
        base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        pr = gpus[device]
        loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        dev = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        try:
            cpus = _cpulist(open(os.path.join(dev, "local_cpulist")).read())
        except OSError:
            cpus = []
        if not cpus:
            node = int(open(os.path.join(dev, "numa_node")).read())
            if node < 0:
                return None
            cpus = _cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        return cpus or None
    except Exception:
        return None


def primary_threads(cpus: Sequence[int], sysfs: str = "/sys") -> List[int]:
    """One logical CPU per physical core: the lowest-numbered hardware thread of every core that has a thread in `cpus`.
    Two busy hardware threads of one core each run at ~60 % speed and are charged full CPU time: under a cgroup CPU quota
    the same host stage delivered 70-80 M or 115-125 M bases/s depending on where the scheduler happened to put the parser
    threads (r05, 2 x 64 cores x 2 threads, quota 16: scripts/host_scaling.py).  Without topology information: `cpus`."""
    keep, have = [], set(cpus)
    try:
        for c in sorted(have):
            sib = _cpulist(open(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list")).read())
            if c == min(x for x in sib if x in have):
                keep.append(c)
    except Exception:
        return sorted(have)
    return keep or sorted(have)


def worker_cpus(rank: int, world: int, devices: Optional[Sequence[int]] = None, allowed: Optional[Sequence[int]] = None,
                sysfs: str = "/sys") -> Optional[List[int]]:
    """The cores GPU worker `rank` of `world` pins itself to (NanoReviser.py:203-219 leaves its Pool wherever the kernel
    puts it), or None = leave the affinity alone.  The cores this process may use - one hardware thread of each
    (`primary_threads`) - are divided so that each worker gets cores of ITS GPU's NUMA node (fast5 buffers, pinned staging memory and the HIP runtime's threads then sit next to the
    GPU's PCIe root); workers whose GPUs share a node share its cores evenly; without NUMA information the allowed
    cores are cut into `world` contiguous slices.  NRV_CPU_AFFINITY=0 turns it off; fewer than two cores per worker: off."""
    if os.environ.get("NRV_CPU_AFFINITY", "1") == "0" or world < 1:
        return None
    try:
        allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    except AttributeError:
        return None
    if len(allowed) < 2 * world:
        return None
    prim = primary_threads(allowed, sysfs)             # one hardware thread per core, when that still leaves two per worker
    if len(prim) >= 2 * world:
        allowed = prim
    devices = list(devices) if devices is not None else list(range(world))
    aset = set(allowed)
    local = []
    for r in range(world):
        c = gpu_local_cpus(devices[r], sysfs)
        c = sorted(aset.intersection(c)) if c else None
        local.append(tuple(c) if c else None)
    if any(c is None for c in local):
        lo, hi = rank * len(allowed) // world, (rank + 1) * len(allowed) // world
        return allowed[lo:hi]
    mine = local[rank]
    peers = [r for r in range(world) if local[r] == mine]
    k, n = peers.index(rank), len(peers)
    part = list(mine[k * len(mine) // n:(k + 1) * len(mine) // n])
    return part if len(part) >= 2 else None


def probe_rename_rate(out_dir: str, pairs: int = 2000) -> Optional[float]:
    """create + write + rename pairs per second the OUTPUT directory's filesystem sustains (one output per read appears
    that way: write_read / nrvh_finish_read), measured on `pairs` small files in a scratch sub-directory.  Why: r05 found
    the sandbox's overlay root to cost ten times the system time of a real filesystem once a directory has seen a few
    thousand creates + renames - the same eight workers fed 68-78 M bases/s writing there and 120-131 M writing to tmpfs,
    i.e. less than eight GPUs consume.  None when the directory cannot be probed."""
    d = os.path.join(out_dir, f".nrv_probe_{os.getpid()}")
    try:
        os.makedirs(d, exist_ok=True)
        payload = b">probe\n" + b"ACGT" * 2048                 # ~8 KB: a typical record
        t0 = time.perf_counter()
        for i in range(pairs):
            tmp = os.path.join(d, f"p{i}.tmp")
            with open(tmp, "wb") as fp:
                fp.write(payload)
            os.replace(tmp, os.path.join(d, f"p{i}_out.fasta"))
        dt = time.perf_counter() - t0
        return pairs / max(dt, 1e-9)
    except OSError:
        return None
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def check_output_rate(args, world: int, n_reads: int, log=print) -> Optional[float]:
    """With several GPU workers and enough reads for it to matter: warn when the output directory cannot take the files the
    GPUs will produce.  One MI355X revises ~12.5 M bases/s = ~1600 reads of 8 k bases per second; ONE thread's probe rate
    must cover the whole run's share with a factor of two to spare (the finishers are a few threads per worker).
    NRV_OUTPUT_PROBE=0 turns the probe off, =1 forces it."""
    mode = os.environ.get("NRV_OUTPUT_PROBE", "")
    if mode == "0" or (mode != "1" and (world < 2 or n_reads < 4000)):
        return None
    rate = probe_rename_rate(args.output_dir)
    need = 1600.0 * world
    if rate is not None and rate < 2.0 * need:
        log(f"[！！！Warning] the output directory sustains {rate:.0f} file creations + renames per second (one thread, "
            f"2000 files); {world} GPU workers produce up to ~{need:.0f} reads per second.  The run will be bound by this "
            "filesystem, not by the GPUs: put -o on a local disk or a tmpfs (overlay / network filesystems are the usual cause).")
    return rate


def share_device() -> bool:
    """NRV_SHARE_DEVICE=1: worker rank r drives device r % device_count - a 1-GPU box rehearsing the N-worker command
    line with the real engine (bench.py --share-device, tests/test_gpu_multirank.py).  Never the default."""
    return os.environ.get("NRV_SHARE_DEVICE", "0") == "1"


def plan_splits(names: Sequence[str], sizes: Sequence[int], world: int, split_mb: float):
    """Which files are revised whole and which by window range.  A fast5 is split only when it would unbalance the run - it is
    larger than a worker's fair share of all bytes - and is worth it (above `split_mb` MB: every worker that holds a slice parses
    the whole file, ~3 ms per MB); it is cut into min(world, ceil(size / max(fair share, split_mb))) slices.  File size is the only
    thing known before parsing (~100 bytes per event).  Returns (units, unit_sizes): a unit is a file name or (file name, slice, slices)."""
    units, usz = [], []
    lim = int(split_mb * (1 << 20)) if split_mb and split_mb > 0 else 0
    fair = sum(int(x) for x in sizes) / max(1, world)
    cut = max(lim, int(fair))
    for fn, sz in zip(names, sizes):
        parts = min(world, -(-int(sz) // cut)) if lim and world > 1 and sz > cut else 1
        if parts <= 1:
            units.append(fn)
            usz.append(int(sz))
        else:
            for k in range(parts):
                units.append((fn, k, parts))
                usz.append(int(sz) // parts)
    return units, usz


def revise_part(args, reviser, fn: str, k: int, parts: int):
    """Slice k of `parts` of ONE read (BASELINE config 5): the read is parsed here (every worker that holds a
    slice parses the file: the host stage is cheap beside 100 k+ windows), `shard.split_read_windows` gives the event
    range [lo, hi) whose windows are exactly windows [lo, hi - T) of the read, and those events alone go to the device -
    whole raw samples, sliced starts / features, so every window is cut and computed as in the unsplit read.
    Returns (payload, error): the slice's calls (+ the original bases and the Fastq record with slice 0)."""
    job = (os.path.join(args.fast5_base_dir, fn), fn, args.basecall_group, args.basecall_subgroup, True)
    _, rt, fq, err, _ = _load_one(job)
    if rt is None or err is not None:
        return None, err or "unreadable"
    T, N = int(reviser.T), len(rt.feat_ev)
    lo, hi = split_read_windows(N, T, parts)[k]
    if hi > lo:
        if isinstance(rt, hs.RawReadTensors):
            p1, p2, a1, a2 = reviser.predict_reads_raw([rt.raw], [rt.starts[lo:hi]], [rt.feat_ev[lo:hi]], [rt.shift], [rt.scale])
        else:
            p1, p2, a1, a2 = reviser.predict_read(rt.sig_ev[lo:hi], rt.feat_ev[lo:hi])
    else:
        p1, p2 = np.zeros((0, 6), np.float32), np.zeros((0, 5), np.float32)
        a1 = a2 = np.zeros(0, np.int8)
    qc = phred_chars(p1, p2, a1, a2) if args.output_format == "fastq" and len(a1) else None
    out = {"T": T, "n_ev": N, "lo": lo, "a1": np.array(a1, np.int8), "a2": np.array(a2, np.int8), "qc": qc}
    if k == 0:
        out["bases"], out["fq"] = np.asarray(rt.bases), fq
    return out, None


def _worker(rank: int, world: int, args, files: List[str], q, factory=None, parts=()):
    """One GPU worker.  Streams a ("file", rank, fn, revised) record through the queue as each file's output
    becomes final, so that the parent knows exactly which files are done should this process die; the slices of
    split reads (`parts`: (fn, slice, slices)) go first - they are the long poles - and come back as ("part", ...)."""
    try:
        cores_all = usable_cores()
        ndev = None
        if share_device():
            from .engine import device_count
            ndev = max(1, device_count())
        device = rank % ndev if ndev else rank
        cpus = worker_cpus(rank, world, [r % ndev if ndev else r for r in range(world)])
        share = None
        if cpus:
            try:
                os.sched_setaffinity(0, cpus)          # before the engine: the HIP runtime's threads inherit it
                share = max(1, min(len(cpus), cores_all // max(1, world)))
            except OSError:
                cpus = None
        made = []

        def make():                                   # one engine per call; process_files may call it from its engine thread
            made.append((factory or _default_factory)(args, device))
            return made[-1]
        rv = make() if parts else None                # slices need the engine now; whole files let it come up beside the parsers
        for fn, k, n in parts:
            try:
                payload, err = revise_part(args, rv, fn, k, n)
            except Exception as e:
                payload, err = None, repr(e)
            q.put(("part", rank, fn, k, payload, err))
        st = process_files(args, files, rv if rv is not None else make, print,
                           on_file=lambda fn, ok: q.put(("file", rank, fn, bool(ok))), gpu_workers=world, core_share=share)
        st["cpus"] = len(cpus) if cpus else 0
        for e in made:
            e.close()
        q.put(("done", rank, st, None))
    except BaseException as e:           # engine could not be created: loud, no silent fallback
        q.put(("done", rank, None, repr(e)))


def write_originals(args, files: Sequence[str], log: Callable[[str], None]) -> List[str]:
    """The failure contract for reads whose WORKER is gone (NanoReviser.py:146-152 / :173-179): every
    file in `files` - the ones the dead worker never reported as final - gets its original basecalls
    written by this process (atomically, replacing whatever an earlier run left under that name).
    Returns the files handled (all of them count as failed reads)."""
    done = []
    for fn in files:
        done.append(fn)
        try:
            rd, fq = parse_read(os.path.join(args.fast5_base_dir, fn), args.basecall_group, args.basecall_subgroup)
            if args.output_format == "fastq" and fq is not None:
                b, q = hs.trim_fastq(fq)
                write_read(args, fn, b, q)
            else:
                write_read(args, fn, "".join(x.decode() for x in np.asarray(rd.bases).tolist()), None)
        except Exception as e:
            log(f"！！！[Error] fast5 file: {fn.split('.')[0]} {e!r}")
    return done


def finish_split_reads(args, split_fns: dict, parts_got: dict, log: Callable[[str], None]):
    """Parent side of the split reads: the slices' calls in slice order are the read's calls (window i of slice k is
    window lo_k + i of the read), merged and written exactly as an unsplit read's.  A read with a slice missing (its
    worker died) or failed gets its original basecalls (NanoReviser.py:146-152).  Returns (bases written, failed names)."""
    spec, nb, failed = _OutSpec(args), 0, []
    for fn, n in sorted(split_fns.items()):
        got = parts_got.get(fn, {})
        pl = [got.get(k, (None, "slice never came back"))[0] for k in range(n)]
        err = next((e for k in range(n) for e in [got.get(k, (None, "slice never came back"))[1]] if e), None)
        ok = n > 0 and err is None and all(p is not None for p in pl) and "bases" in pl[0]
        if ok:                                         # every worker parsed the same file: same event count, slices abut
            T, N = pl[0]["T"], pl[0]["n_ev"]
            want = split_read_windows(N, T, n)
            ok = all(p["n_ev"] == N and p["T"] == T and p["lo"] == want[k][0] and len(p["a1"]) == max(want[k][1] - want[k][0] - T, 0)
                     for k, p in enumerate(pl))
            err = None if ok else "slices do not tile the read"
        if ok:
            a1, a2 = np.concatenate([p["a1"] for p in pl]), np.concatenate([p["a2"] for p in pl])
            qc = np.concatenate([p["qc"] if p["qc"] is not None else np.zeros(0, np.uint8) for p in pl]) \
                if args.output_format == "fastq" and len(a1) else None
            w, e2 = (_finish_native if hostlib.load() is not None else _finish_in_worker)(spec, T, fn, pl[0]["bases"], a1, a2, qc)
            if e2 is None:
                nb += w
                log(f"[p:::] {fn.split('.')[0]}_out.{args.output_format} was saved...... ({n} slices)")
                continue
            err = e2
        log(f"[！！！Error] revising {fn.split('.')[0]}: {err}; writing the original basecalls")
        write_originals(args, [fn], log)
        failed.append(fn)
    return nb, failed


def run_workers(args, shards: List[List[str]], factory=None, poll_s: float = 0.2, part_shards=None, parts_out=None):
    """One worker process per GPU (NanoReviser.py:203-219 fans out the same way, one Pool task per
    file).  The parent never blocks on the result queue alone: a worker that dies hard - a HIP memory
    fault aborts the process, a segfault, the OOM killer - posts nothing, so liveness is polled next to
    the queue.  The other workers finish their own shards (reads are independent).
    Returns [(rank, stats or None, error or None, files)] with files = {fn: revised} for every file the
    worker reported final before it ended (file existence is never used as evidence).
    part_shards[r]: the (fn, slice, slices) units of worker r; their results are collected into parts_out
    {fn: {slice: (payload, error)}} for `finish_split_reads`."""
    import multiprocessing as mp
    import queue as _q
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = len(shards)
    procs = [ctx.Process(target=_worker, args=(r, world, args, shards[r], q, factory,
                                               tuple(part_shards[r]) if part_shards else ())) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    files = [dict() for _ in range(world)]

    def drain(timeout):
        try:
            while True:
                msg = q.get(timeout=timeout)
                if msg[0] == "file":
                    files[msg[1]][msg[2]] = msg[3]
                elif msg[0] == "part":
                    if parts_out is not None:
                        parts_out.setdefault(msg[2], {})[msg[3]] = (msg[4], msg[5])
                else:
                    _, rank, st, err = msg
                    res[rank] = (rank, st, err)
                timeout = 0.0
        except _q.Empty:
            pass

    while len(res) < world:
        drain(poll_s)
        for r, pr in enumerate(procs):
            if r not in res and not pr.is_alive():
                drain(0.5)                            # its last messages may still be in the pipe
                if r not in res:
                    res[r] = (r, None, f"worker {r} died without reporting (exit code {pr.exitcode})")
    for pr in procs:
        pr.join(30)
        if pr.is_alive():
            pr.terminate()
    return [res[r] + (files[r],) for r in range(world)]


def main(argv: Optional[Sequence[str]] = None, reviser_factory=None, standalone: bool = False,
         worker_factory=None, world: Optional[int] = None) -> int:
    """standalone=True (the NanoReviser.py entry point): this process is only the command line, so the
    engine is loaded without importing torch first (~1.5 s of start-up).  Library callers keep the
    default: engine.load_library imports torch first so that one HIP runtime serves both.
    worker_factory (a picklable module-level callable (args, device) -> engine) and world replace the
    engine constructor and the device count in the multi-process path (tests).
    Returns 0, 2 (nothing could run) or 3 (a GPU worker was lost; its reads were written unrevised)."""
    args = get_args(argv)
    rc = 0
    if args.output_format not in ("fasta", "fastq"):
        print("[！！！Error] output_format must be fasta or fastq", file=sys.stderr)
        return 2
    model_paths(args)                     # existence check up front (NanoReviser.py:194)
    names = sorted(f for f in os.listdir(args.fast5_base_dir) if f.endswith(".fast5"))
    os.makedirs(args.output_dir, exist_ok=True)
    t0 = time.time()
    kept_failed, n_resumed = [], 0
    if args.resume:
        # A finished read is a non-empty <stem>_out.<fmt>: outputs only ever appear by rename of a complete temporary.
        # Reads an earlier run wrote UNREVISED (fallback after a lost GPU worker / an engine failure / a failed split-read
        # merge) are in its failed-reads file: they are taken out of `skip` and revised again - write_read replaces the
        # output atomically.  --resume_keep_failed: they stay as they are, stay listed, and the run returns 3.
        def done(f):
            try:
                return os.path.getsize(out_name(args.output_dir, f, args.output_format)) > 0
            except OSError:
                return False
        skip = {f for f in names if done(f)}
        try:
            with open(os.path.join(args.output_dir, args.failed_reads_filename)) as fp:
                earlier_failed = [ln.strip() for ln in fp if ln.strip() in skip]
        except OSError:
            earlier_failed = []
        if args.resume_keep_failed:
            kept_failed = earlier_failed
            if kept_failed:
                rc = 3
        else:
            skip -= set(earlier_failed)
        n_resumed, names = len(skip), [f for f in names if f not in skip]
        print(f"[s:::] --resume: {n_resumed} reads already have their output, {len(names)} to do"
              + (f" ({len(earlier_failed)} of them written unrevised by the earlier run"
                 + (", left as they are)" if args.resume_keep_failed else ", revised again)") if earlier_failed else ""))
    if reviser_factory is not None:       # in-process (tests / embedding): one engine, no sharding
        rv = reviser_factory(args, 0)
        stats = [process_files(args, names, rv, print)]
    else:
        # the command line needs no torch: without it a process starts ~1.5 s sooner.  (engine.py imports
        # torch first only so that a LATER torch import in the same process finds one HIP runtime.)
        if standalone and "torch" not in sys.modules:
            os.environ.setdefault("NRV_NO_TORCH", "1")
        if world is not None:
            ndev = int(world)
        else:
            from .engine import device_count
            ndev = device_count()
        if ndev == 0:
            print("[！！！Error] no MI355X / HIP device visible: the reviser has no CPU path", file=sys.stderr)
            return 2
        # workers: --gpus (default: every visible device), never more than there are units of work; NRV_SHARE_DEVICE=1
        # (rehearsal) lets --gpus exceed the device count, worker r then drives device r % count
        want = min(args.gpus or ndev, args.gpus if (share_device() and args.gpus) else ndev)
        sizes = [os.path.getsize(os.path.join(args.fast5_base_dir, f)) for f in names]
        units, usz = plan_splits(names, sizes, want, args.split_reads_above)
        world = max(1, min(want, len(units)))
        parts = shard_reads(usz, world)
        check_output_rate(args, world, len(names), lambda m: print(m, file=sys.stderr))
        if world == 1:
            made = []

            def make():                               # runs in the engine thread, beside the parser pool's start-up
                made.append((worker_factory or _default_factory)(args, 0))
                return made[-1]                       # called once per engine (process_files: NRV_CLI_ENGINES)
            stats = [process_files(args, names, make, print)]
            for rv in made:
                rv.close()
        else:
            shards = [[units[i] for i in parts[r] if isinstance(units[i], str)] for r in range(world)]
            part_shards = [[units[i] for i in parts[r] if not isinstance(units[i], str)] for r in range(world)]
            split_fns = {u[0]: u[2] for u in units if not isinstance(u, str)}      # file -> slices
            parts_got = {}
            res = run_workers(args, shards, worker_factory, part_shards=part_shards, parts_out=parts_got)
            stats = [s for _, s, _, _ in res if s is not None]
            if split_fns:
                nb_split, failed_split = finish_split_reads(args, split_fns, parts_got, print)
                stats.append({"reads": len(split_fns), "bases": nb_split, "failed": failed_split, "host_s": 0.0, "engine_s": 0.0})
                if failed_split:
                    rc = 3
            for r, s_, e, final in res:
                if s_ is None:                        # the shard's files still get an output + a failed_reads entry
                    print(f"[！！！Error] GPU worker {r} failed: {e}; writing the original basecalls of its "
                          f"unfinished reads", file=sys.stderr)
                    # what the worker reported final stays (its failed reads keep their failed_reads entry);
                    # everything else - started or not, whatever lies on disk - is written unrevised
                    lost = write_originals(args, [f for f in shards[r] if f not in final], print)
                    lost += [f for f, ok in final.items() if not ok]
                    stats.append({"reads": len(shards[r]), "bases": 0, "failed": lost, "host_s": 0.0, "engine_s": 0.0})
                    rc = 3
    failed = kept_failed + [f for s in stats for f in s["failed"]]
    with open(os.path.join(args.output_dir, args.failed_reads_filename), "w") as fp:
        fp.write("".join(f + "\n" for f in failed))
    dt = time.time() - t0
    if not args.test_mode:
        nb = sum(s["bases"] for s in stats)
        print("[s:::] All subprocesses done.")
        print("[s:::] NanoReviser time consuming:%.2f seconds" % dt)
        print(f"[s:::] {sum(s['reads'] for s in stats)} reads, {nb} bases, {len(failed)} failed, "
              f"{nb / max(dt, 1e-9):.0f} bases/s end to end "
              f"(host stage {sum(s['host_s'] for s in stats):.1f} s summed over workers, "
              f"engine thread {sum(s['engine_s'] for s in stats):.1f} s)")
    return rc


if __name__ == "__main__":
    sys.exit(main(standalone=True))
