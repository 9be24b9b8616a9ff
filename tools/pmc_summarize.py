#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/pmc_dominant.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline-events
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline-events
    python tools/pmc_summarize.py gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv 'conv3x3_f16x3<1,8,false>'

Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE is doubled on gfx950 (wide
coalesced reads are reported at half); separate passes because the two counters do not share a pass reliably."""
import csv
import json
import os
import sys


def per_launch(path, counter, kernel):
    vals = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void aipt::", "").split("(")[0].replace(" ", "")
        if name == kernel and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"{path}: no {counter} rows for kernel {kernel}")
    return sum(vals) / len(vals), len(vals)


def main():
    fcsv, wcsv, kernel = sys.argv[1:4]
    f, nf = per_launch(fcsv, "FETCH_SIZE", kernel)
    w, nw = per_launch(wcsv, "WRITE_SIZE", kernel)
    fetch = 2.0 * f * 1024.0
    write = w * 1024.0
    out = {
        "kernel": kernel,
        "hbm_bytes_per_launch": fetch + write,
        "fetch_bytes_per_launch": fetch,
        "write_bytes_per_launch": write,
        "launches_averaged": [nf, nw],
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --warmup 2 "
                  "--no-cpu-baseline --no-roofline-events); KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                  "(gfx950 reports half of wide coalesced reads); WRITE_SIZE uncalibrated; averaged over all launches "
                  "of the kernel in the run (tools/pmc_summarize.py)",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_dominant.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
