#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/pmc_dominant.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline-events
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline-events
    python tools/pmc_summarize.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv \\
        'trace_bounce<false,true>' 'conv3x3_f16x3<1,8,false,false>' ...

(the SAME bench.py command line the roofline block is measured with, so "per launch" means the same launch: with the default
--batch 4 a bounce launch covers four frames).  Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in
KiB; FETCH_SIZE is doubled on gfx950 (wide coalesced reads are reported at half); separate passes because the two counters do
not share a pass.  bench.py reads {"kernels": {name: {"hbm_bytes_per_launch": ...}}} for roofline.traffic."""
import csv
import json
import os
import sys


def short(name):
    n = name.replace("void aipt::", "").replace("aipt::", "")
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace(" ", "")


def per_launch(path, counter):
    """kernel -> counter values of its launches.  The bounce kernels run batched (the timed calls) AND one frame at a time (bench.py
    re-renders its frames for validation) under one name since round 6: only the launches with the largest grid -- the batched
    ones -- are averaged, so that "per launch" keeps meaning a launch of the timed region."""
    acc, grid = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
            grid.setdefault(k, []).append(int(r["Grid_Size"]))
    for k in acc:
        if k.startswith("trace_bounce"):
            top = max(grid[k])
            acc[k] = [v for v, g in zip(acc[k], grid[k]) if g == top]
    return acc


def main():
    args = sys.argv[1:]
    outpath = None
    if "--out" in args:                                # default: profiles/pmc_dominant.json of this checkout
        i = args.index("--out")
        outpath = args[i + 1]
        del args[i:i + 2]
    fpl = None
    if "--frames-per-launch" in args:                  # frames every batched trace launch of the PMC command held (bench.py
        i = args.index("--frames-per-launch")          # reports roofline_other.traffic only for runs with the same count)
        fpl = int(args[i + 1])
        del args[i:i + 2]
    fcsv, wcsv = args[0:2]
    kernels = args[2:]
    f, w = per_launch(fcsv, "FETCH_SIZE"), per_launch(wcsv, "WRITE_SIZE")
    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the default bench.py command "
                     f"with one denoiser stream (AIPT_DN_PIPELINE=0 ... --steps {fpl or 20} --warmup {fpl or 20} --no-cpu-baseline --no-roofline-events: "
                     "a chip-wide counter belongs to one kernel only while one kernel runs); KiB -> bytes; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; the BVH walk's 16-byte gathers are not "
                     "the calibrated pattern, so its figure is an upper bound); WRITE_SIZE uncalibrated; averaged over all "
                     "launches of the kernel in the run (tools/pmc_summarize.py)",
           "kernels": {}}
    # forward passes of the run = launches of the output layer's second pass (one per denoised frame): gives every conv kernel its
    # launches per frame, which bench.py compares with the launch mix of ITS run before quoting the traffic
    nframes = len(f.get("conv3x3_quad<3,3,false>", [])) or None
    out["frames_denoised"] = nframes
    for k in kernels:
        if k not in f or k not in w:
            print(f"no rows for kernel {k}; have {sorted(f)}", file=sys.stderr)
            continue
        fetch = 2.0 * 1024.0 * sum(f[k]) / len(f[k])
        write = 1024.0 * sum(w[k]) / len(w[k])
        out["kernels"][k] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch,
                             "write_bytes_per_launch": write, "launches_averaged": [len(f[k]), len(w[k])]}
        if nframes and k.startswith("conv3x3"):
            out["kernels"][k]["launches_per_frame"] = round(len(f[k]) / nframes, 4)
        if fpl and k.startswith("trace_bounce") and k.endswith(",true>"):       # the pooled (batched) instantiation
            out["kernels"][k]["frames_per_launch"] = fpl
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(outpath or os.path.join(root, "profiles", "pmc_dominant.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["kernels"]), file=sys.stderr)


if __name__ == "__main__":
    main()
