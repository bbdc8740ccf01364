#!/opt/conda/bin/python3.9
"""Generate host-stage golden vectors by RUNNING THE REFERENCE'S OWN CODE here.

Build-container only: needs /root/reference and h5py (/opt/conda/bin/python3.9).
Run as:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tools/make_goldens.py

What runs, unmodified, from /root/reference:
  * nanorevutils/preprocessing.py (NumPy only) imported as a module:
      signal_segmentation :85-170, get_base_color :173-175, get_base_label :178-180
  * nanorevutils/nanorev_fast5_handeler.py cannot be imported (albacore at module top),
    so the two FunctionDefs get_read_data :39-150 and extract_fastq :152-171 are pulled
    out of its AST and exec'd with h5py/np/LooseVersion in scope;
  * nanorevutils/output_handeler.py cannot be imported (keras at module top); the
    FunctionDefs get_base_1 :104-122, prep_read_fasta :26-45, prep_read_fastq :48-62 and
    the `label_to_base` assignment :83 are AST-extracted the same way.
Only their INPUTS and OUTPUTS are written (tests/golden/), never their source.

Per fixture read (unitest/test_data/fast5/*.fast5) -> tests/golden/reads/<key>.npz:
  inputs : Events columns (mean,start,stdv,length,model_state,move), raw Signal int16,
           Fastq record, albacore version attr, file name
  outputs: get_read_data -> abs_event_start,start,length,bases,ab_mean,ab_std
           signal_segmentation -> shift, scale, mean[N], std[N] (f64), and for the
           (N,50) f64 window matrix its sha256 + an evenly spaced row subset (the full
           matrix is 16 MB for the five reads; it is a pure function of the inputs)
           extract_fastq -> trimmed bases/quals
Merge / writer vectors -> tests/golden/merge_vectors.json
"""
import ast
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import h5py
from distutils.version import LooseVersion

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def extract(path, names, scope):
    src = open(path).read()
    tree = ast.parse(src)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked.append(node)
        elif isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id in names for t in node.targets):
            picked.append(node)
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, path, "exec"), scope)
    return scope


def main():
    sys.path.insert(0, REF)
    from nanorevutils import preprocessing as pp   # reference module, unmodified

    fh = extract(os.path.join(REF, "nanorevutils", "nanorev_fast5_handeler.py"),
                 {"get_read_data", "extract_fastq"},
                 {"h5py": h5py, "np": np, "LooseVersion": LooseVersion})
    oh = extract(os.path.join(REF, "nanorevutils", "output_handeler.py"),
                 {"get_base_1", "label_to_base", "prep_read_fasta", "prep_read_fastq"},
                 {"os": os})

    os.makedirs(os.path.join(OUT, "reads"), exist_ok=True)
    d = os.path.join(REF, "unitest", "test_data", "fast5")
    index = []
    for fn in sorted(os.listdir(d)):
        path = os.path.join(d, fn)
        key = "_".join(fn.split("_")[-3:-1])          # ch10_read5252
        f = h5py.File(path, "r")
        grp = "/Analyses/Basecall_1D_000"
        ev = f[grp + "/BaseCalled_template/Events"][()]
        fastq = f[grp + "/BaseCalled_template/Fastq"][()]
        version = f[grp].attrs["version"]
        rname = list(f["/Raw/Reads/"].keys())[0]
        raw = f["/Raw/Reads/" + rname + "/Signal"][()]
        f.close()

        (abs_start, start, length, bases, signal, ab_mean, ab_std) = fh["get_read_data"](
            path, "Basecall_1D_000", "BaseCalled_template")
        assert np.array_equal(signal, raw)
        # NanoReviser.py:120-125
        sig = signal[int(abs_start):]
        sig_list, s_mean, s_std, shift, scale = pp.signal_segmentation(
            sig, start, int(length[-1]))
        sig_list = np.ascontiguousarray(sig_list, dtype=np.float64)
        N = len(bases)
        assert sig_list.shape == (N, 50), sig_list.shape
        rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(N - 40, N),
                                         np.arange(0, N, 61)]))
        fq_bases, fq_qual = fh["extract_fastq"](path, None)
        colors = np.array([pp.get_base_color(b) for b in bases], dtype=np.int32)

        np.savez_compressed(
            os.path.join(OUT, "reads", key + ".npz"),
            file_name=np.bytes_(fn), albacore_version=np.bytes_(version if isinstance(version, bytes) else str(version).encode()),
            ev_mean=ev["mean"], ev_start=ev["start"], ev_stdv=ev["stdv"],
            ev_length=ev["length"], ev_model_state=ev["model_state"], ev_move=ev["move"],
            raw_signal=raw.astype(np.int16), fastq=np.bytes_(fastq),
            rd_abs_event_start=np.int64(abs_start), rd_start=np.asarray(start, dtype=np.int64),
            rd_length=np.asarray(length, dtype=np.float64),
            rd_bases=np.array(bases, dtype="S1"),
            rd_ab_mean=np.asarray(ab_mean, dtype=np.float32),
            rd_ab_std=np.asarray(ab_std, dtype=np.float32),
            rd_colors=colors,
            seg_shift=np.float64(shift), seg_scale=np.float64(scale),
            seg_mean=np.asarray(s_mean, dtype=np.float64),
            seg_std=np.asarray(s_std, dtype=np.float64),
            seg_sig_sha256=np.bytes_(hashlib.sha256(sig_list.tobytes()).hexdigest().encode()),
            seg_sig_rows=rows.astype(np.int64), seg_sig_vals=sig_list[rows],
            fq_bases=np.bytes_(fq_bases.encode()), fq_qual=np.bytes_(fq_qual.encode()),
        )
        index.append({"key": key, "file": fn, "n_bases": int(N), "n_events": int(len(ev)),
                      "n_raw": int(len(raw))})
        print(key, N, len(ev), len(raw), shift, scale)
    with open(os.path.join(OUT, "reads", "index.json"), "w") as o:
        json.dump(index, o, indent=1)

    # ---- merge + writers (output_handeler.py:104-122, :26-62) -----------------------
    rng = np.random.default_rng(7)
    cases = []

    def run_merge(ev_bases, y1, y2):
        out = oh["get_base_1"](list(ev_bases), np.asarray(y1), np.asarray(y2))
        cases.append({"event_bases": "".join(ev_bases), "y_pre": [int(v) for v in y1],
                      "y_pre2": [int(v) for v in y2], "result": out})

    # y_pre are model1 labels 0..5; y_pre2 as a caller has to pass them: label+1 (1..6)
    for n in (1, 2, 5, 17, 64, 257):
        for _ in range(6):
            evb = rng.choice(list("ACGT"), n)
            run_merge(evb, rng.integers(0, 6, n), rng.integers(1, 7, n))
    # structured cases: all agree, all deletions, all insertions, mismatched lengths
    evb = list("ACGTACGTAC")
    lab = {"A": 5, "G": 4, "T": 3, "C": 2}
    y = [lab[b] for b in evb]
    run_merge(evb, y, [v + 1 for v in y])
    run_merge(evb, [0] * 10, [v + 1 for v in y])
    run_merge(evb, [1] * 10, [2] * 10)
    run_merge(evb, y[:7], [v + 1 for v in y])          # zip truncation
    run_merge(evb, [1] + y[1:], [2] + [v + 1 for v in y[1:]])   # seeded '-' is filtered

    writers = []
    with tempfile.TemporaryDirectory() as td:
        for name, bases, qul in (("/a/b/read one.fast5", list("ACGTTGCA"), list("!!##$$%%")),
                                 ("plain.fast5", list("A"), list("I"))):
            p = os.path.join(td, "o.fasta")
            oh["prep_read_fasta"](name, p, bases)
            fa = open(p).read()
            p = os.path.join(td, "o.fastq")
            oh["prep_read_fastq"](name, p, bases, qul)
            fq = open(p).read()
            writers.append({"fast5_fn": name, "bases": "".join(bases), "qul": "".join(qul),
                            "fasta": fa, "fastq": fq})

    labels = {"base_color": {b: pp.get_base_color(b) for b in "ACGTN-D"},
              "base_label": {b: pp.get_base_label(b) for b in "ACGTN-D"},
              "label_to_base": {str(k): v for k, v in oh["label_to_base"].items()}}
    with open(os.path.join(OUT, "merge_vectors.json"), "w") as o:
        json.dump({"get_base_1": cases, "writers": writers, "labels": labels}, o, indent=1)
    print("merge cases:", len(cases))


if __name__ == "__main__":
    main()
