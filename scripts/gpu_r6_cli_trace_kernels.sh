#!/bin/bash
# Round 6: a rocprofv3 kernel trace of the COMMAND LINE (4000 human-model reads, one GPU, device calls pipelined): how busy the
# device is between the first and the last kernel, and what the five launches cost per 4096-window group in READ mode (the signal
# branch runs once per event there, not once per window and step).
O=gpurun_out/${1:-r06t}; mkdir -p $O
export TMPDIR=/tmp
D=/dev/shm/nrv_kt_in; OUT=/dev/shm/nrv_kt_out/
rm -rf $D $OUT $O/kt; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 800); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
# (NanoReviser.py leaves through os._exit, which would skip the profiler's flush: the same main() with an orderly exit)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 -c "import sys; from nanoreviser_amd.cli import main; sys.exit(main(['-d', '$D', '-o', '$OUT', '-S', 'human', '--thread', '16'], standalone=True))" > $O/cli_under_rocprof.log 2>&1
grep "bases/s end to end" $O/cli_under_rocprof.log
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
rows = []
for f in glob.glob(o + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
first, last = rows[0][0], max(e for _, e, _ in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(list)
for s, e, k in rows:
    per[k.split("(")[0].replace("void nrv::", "")[:60]].append(e - s)
with open(o + "/cli_kernel_trace_summary.txt", "w") as fp:
    def out(x):
        print(x); fp.write(x + "\n")
    out(f"{len(rows)} kernel dispatches; first kernel start .. last kernel end {(last - first) / 1e6:.1f} ms; some kernel running {busy / 1e6:.1f} ms = {busy / (last - first):.3f} of that span")
    tot = sum(sum(v) for v in per.values())
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        out(f"  {k:62s} x{len(v):6d}  median {v2[len(v2) // 2] / 1e3:8.2f} us  mean {sum(v) / len(v) / 1e3:8.2f} us  {100 * sum(v) / tot:5.1f} % of kernel time")
PY
rm -rf $D $OUT $O/kt
