#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

Usage (on the GPU box, separate passes as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (guide, section HBM): FETCH_SIZE reports half of the bytes
of wide (16 B/lane) coalesced streaming reads; kernels whose global reads are dwordx4 get the x2 correction
(flag per kernel below), dword-load kernels are reported uncorrected (uncalibrated width, stated as such)."""
import collections
import csv
import glob
import json
import sys

WIDE_READ_KERNELS = ("d3_fwd_k", "d3_pull_k", "d3_wgrad_k", "dgrad_loop_k", "igemm_k<3, 1, 1, 0, 4, 160", "igemm_k<3, 1, 1, 0, 8, 80",
                     "igemm_k<3, 1, 1, 0, 4, 80", "wgrad_dense_q_k", "grad_finalize_k", "adamw_k", "reduce_rows")
# the transition kernels (p1_*, c3_*) gather 4..16 B per lane in 64-byte runs, not wide coalesced streams: reported
# uncorrected (uncalibrated width, stated as such in the summary)


def load(d, counter):
    path = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        agg[k] += float(r["Counter_Value"]) * 1024.0
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            calls[k] += 1
    return agg, calls


def main():
    fetch, calls = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        wide = any(t in k for t in WIDE_READ_KERNELS)
        f = fetch[k] * (2.0 if wide else 1.0)
        out[k] = {"launches": calls[k], "fetch_bytes_per_launch": f / calls[k],
                  "write_bytes_per_launch": write.get(k, 0.0) / calls[k],
                  "fetch_correction": "x2 (16 B/lane streaming reads)" if wide else "none (dword loads, uncalibrated)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    top = sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{k[:70]:70s} launches {v['launches']:4d}  fetch {v['fetch_bytes_per_launch'] / 1e6:8.1f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:8.1f} MB per launch")


if __name__ == "__main__":
    main()
