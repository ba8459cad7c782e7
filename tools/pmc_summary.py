"""Summarises rocprofv3 --pmc runs (counter_collection.csv) of the K1 kernels into the JSON
that bench.py reads for `roofline.traffic` (profiles/k1_traffic.json).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-B read
requests of a coalesced stream at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots).

One K1 forward op is the tile kernel followed by the finalize kernel (fpsg_chamfer_fwd_tiled) and, in the step, the
loss sums' one-workgroup last stage (fpsg_chamfer_fwd_tiled_losses), or
one two-pass kernel (fpsg_chamfer_fwd) for few pairs; the backward op is one kernel.  Dispatches
are grouped into ops in dispatch order; the number of cloud pairs of an op comes from the grid of
its LAST kernel at N = M = 2048 (finalize: 4096 threads per pair; sorted backward: 2048 per pair;
two-pass forward <R,W>: 4096*W/R per pair)."""
import collections
import csv
import glob
import json
import re
import sys


def dispatches(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(chamfer_\w+?_kernel)(<[^>]*>)?", r["Kernel_Name"])
            if m:
                rows.append((int(r["Dispatch_Id"]), m.group(1), m.group(2) or "", int(r["Grid_Size"]),
                             float(r["Counter_Value"])))
    rows.sort()
    return rows


pair_bwd = set()     # cloud-pair counts whose backward formed the per-pair constants itself (no upstream-gradient rows read)


def ops(rows):
    """-> {(op, cloud_pairs): [KiB per op, ...]}"""
    out = collections.defaultdict(list)
    pending = last_fwd = None
    pair_bwd.clear()
    for _, name, tmpl, grid, val in rows:
        if name == "chamfer_tile_kernel":
            pending = val
        elif name == "chamfer_finalize_kernel":
            out[("chamfer_fwd", grid // 4096)].append(val + (pending or 0.0))
            pending = None
            last_fwd = ("chamfer_fwd", grid // 4096)
        elif name == "chamfer_loss_reduce_kernel" and last_fwd is not None:
            out[last_fwd][-1] += val            # the loss sums' last stage belongs to the forward op before it
            last_fwd = None
        elif name == "chamfer_fwd_kernel":
            R, W = (int(v) for v in re.findall(r"\d+", tmpl))
            out[("chamfer_fwd", grid * R // (4096 * W))].append(val)
        elif name == "chamfer_bwd_sorted_kernel":
            out[("chamfer_bwd", grid // 2048)].append(val)
            if tmpl.replace(" ", "").endswith(",true>"):      # <REG_SORT, PAIR = true>: fpsg_chamfer_bwd_losses
                pair_bwd.add(grid // 2048)
        elif name == "chamfer_bwd_kernel":
            out[("chamfer_bwd", grid // 4096)].append(val)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch, write = ops(dispatches(sys.argv[1], "FETCH_SIZE")), ops(dispatches(sys.argv[2], "WRITE_SIZE"))
    out = {"how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex chamfer "
                  "-- python3 bench.py --no-extra --no-cpu-baseline --steps 2 --warmup 1 (tools/pmc_k1.sh)",
           "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, summed over the kernels of one op",
           "ops": []}
    for key in sorted(set(fetch) | set(write)):
        op, pairs = key
        f, w = fetch.get(key), write.get(key)
        rec = {"op": op, "cloud_pairs": pairs, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
               "algorithmic_bytes": pairs * (81920 if op == "chamfer_fwd" else (114688 if pairs in pair_bwd else 131072))}
        if f is not None and w is not None:
            rec["hbm_bytes_per_op"] = (2 * f + w) * 1024
            rec["ratio_to_algorithmic"] = rec["hbm_bytes_per_op"] / rec["algorithmic_bytes"]
        out["ops"].append(rec)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
