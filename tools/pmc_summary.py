"""Summarises rocprofv3 --pmc runs (counter_collection.csv) of the K1 kernels into the JSON
that bench.py reads for `roofline.traffic`.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the
128-B read requests of a coalesced stream at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE
is exact.  FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots)."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(chamfer_\w+?_kernel)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            agg[(m.group(1) + (m.group(2) or ""), int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(
                float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex chamfer "
                  "-- python bench.py --workload c3 --no-graph --steps 2 --warmup 1 --no-cpu-baseline",
           "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "launches": []}
    for key in sorted(set(fetch) | set(write)):
        name, grid, wg = key
        f, w = fetch.get(key), write.get(key)
        rec = {"kernel": name, "grid_threads": grid, "workgroup": wg, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
        m = re.search(r"<(\d+), (\d+)>", name)
        if m:      # forward: grid = 4096 * B * W / R threads at N = M = 2048
            rec["cloud_pairs"] = grid * int(m.group(1)) // (4096 * int(m.group(2)))
            rec["algorithmic_bytes"] = rec["cloud_pairs"] * 81920
        else:      # backward: 2 sides * B * 2048 threads
            rec["cloud_pairs"] = grid // 4096
            rec["algorithmic_bytes"] = rec["cloud_pairs"] * 131072
        if f is not None and w is not None:
            rec["hbm_bytes_per_launch"] = (2 * f + w) * 1024
        out["launches"].append(rec)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
