#!/usr/bin/env python3
"""Per-kernel averages of every counter in a set of rocprofv3 --pmc passes.

    python tools/pmc_table.py gpurun_out/pmc_*/p_counter_collection.csv [--kernel SUBSTR] [--json out.json]

Rows: kernel instantiation (namespace and argument list stripped); columns: counter averages per launch (KiB counters
FETCH_SIZE / WRITE_SIZE are converted to bytes; FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md prescribes),
plus launches, mean duration (us) from the dispatch timestamps of the same pass, VGPRs, LDS and scratch."""
import collections
import csv
import json
import sys


def short(name):
    n = name.replace("void aipt::", "").replace("aipt::", "")
    depth = 0
    for i, ch in enumerate(n):                       # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace(" ", "")


def main():
    args = sys.argv[1:]
    want, jout = None, None
    paths = []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            want = args.pop(0)
        elif a == "--json":
            jout = args.pop(0)
        else:
            paths.append(a)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for p in paths:
        seen = set()
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if want and want not in k:
                continue
            if k.startswith("trace_"):                 # batched and single-frame launches share a name: one row per grid size
                k = f"{k} @ {int(r['Grid_Size']) // int(r['Workgroup_Size'])} workgroups"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (p, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                acc[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                           lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), wg=int(r["Workgroup_Size"]))
    out = {}
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_dur_us"])):
        row = dict(meta[k])
        for c, v in cs.items():
            m = sum(v) / len(v)
            if c == "FETCH_SIZE":
                row["FETCH_bytes"] = 2.0 * 1024.0 * m
            elif c == "WRITE_SIZE":
                row["WRITE_bytes"] = 1024.0 * m
            elif c == "_dur_us":
                row["dur_us_under_pmc"] = round(m, 2)
            else:
                row[c] = m
            row.setdefault("launches", {})[c] = len(v)
        out[k] = row
    if jout:
        json.dump(out, open(jout, "w"), indent=1)
    for k, row in out.items():
        print(k)
        for c, v in row.items():
            if c != "launches":
                print(f"    {c:32s} {v:,.2f}" if isinstance(v, float) else f"    {c:32s} {v}")


if __name__ == "__main__":
    main()
