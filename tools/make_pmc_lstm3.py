#!/usr/bin/env python3
"""profiles/<tag>_pmc_lstm3.json - what bench.py's `roofline.traffic` reads - from the per-mode PMC summaries of
scripts/gpu_prof.sh (tools/parse_pmc.py).  Records, next to the HBM bytes per launch of the 192->128 Bi-LSTM, the
kernel's name as rocprofv3 saw it, the sha256 of the sources it was built from and the commit, so that bench.py can
refuse a figure that no longer describes the kernel it runs.
  python3 tools/make_pmc_lstm3.py r04 profiles/r04a_pmc_summary_f16x2.json [profiles/r04a_pmc_summary_f32.json ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag, files = sys.argv[1], sys.argv[2:]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "nanoreviser_amd/csrc"], capture_output=True, text=True).stdout.strip()
out = {}
for f in files:
    prec = os.path.basename(f).rsplit("_", 1)[1].split(".")[0]
    allk = json.load(open(f))
    j = allk["lstm3"]
    out[prec] = {"T": 13, "batch": 4096, "kernel_name": j.get("kernel_name", ""),
                 "source_sha256_16": bench.kernel_source_sha(prec), "commit": commit + ("+" if dirty else ""),
                 "hbm_bytes_per_launch": j["hbm_bytes_per_launch"], "hbm_read_bytes_corrected": j["hbm_read_bytes_corrected"],
                 "hbm_write_bytes": j["hbm_write_bytes"],
                 "source": f"{os.path.relpath(f, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 "
                           "per MI355X_MICROARCH.md HBM section)"}
    # MFMA utilisation of the dominant kernel: cycles a SIMD's matrix pipe was busy / cycles the launch lasted x SIMDs
    # (SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in j and j.get("GRBM_GUI_ACTIVE"):
        out[prec]["mfma_busy_frac"] = j["SQ_VALU_MFMA_BUSY_CYCLES"] / (j["GRBM_GUI_ACTIVE"] / 8 * 1024)
        out[prec]["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), rocprofv3 --pmc pass "
                                       "of this kernel (the profiler lowers the clock: a fraction of cycles, not of peak FLOP/s)")
    # the whole step: HBM-side bytes of every launch
    per = {k: v["hbm_bytes_per_launch"] for k, v in allk.items() if isinstance(v, dict) and v.get("hbm_bytes_per_launch")}
    out[prec]["traffic_step"] = sum(per.values())
    out[prec]["traffic_step_by_kernel"] = per
    out[prec]["step_source_sha256_16"] = bench.step_source_sha()
    assert bench.KERNEL_SIGNATURE[prec] in out[prec]["kernel_name"], (prec, out[prec]["kernel_name"])
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_lstm3.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps(out, indent=1))
