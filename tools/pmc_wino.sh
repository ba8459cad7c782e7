#!/bin/bash
# HBM traffic of the K6 transform kernels at the episode's shapes (run on the GPU box): two rocprofv3 PMC
# passes over tools/bench_wino_transforms.py, one line per (kernel, grid) with the mean KiB per dispatch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_wino_$C
  rocprofv3 --pmc $C --kernel-include-regex wino_ --output-format csv -d $R/gpurun_out/pmc_wino_$C -o w -- \
    python3 $R/tools/bench_wino_transforms.py > $R/gpurun_out/pmc_wino_$C.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, re
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/pmc_wino_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                name = re.sub(r"^.*?(wino_\w+<[^>]*>).*$", r"\1", r["Kernel_Name"])
                d[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    acc[C] = {k: sum(v) / len(v) for k, v in d.items()}
print("# kernel, grid threads, HBM read MB (2*FETCH_SIZE KiB), HBM write MB")
for k in sorted(acc["FETCH_SIZE"]):
    print(f"{k[0]:40s} {k[1]:10d}  read {2 * acc['FETCH_SIZE'][k] * 1024 / 1e6:8.1f} MB  write {acc['WRITE_SIZE'].get(k, 0) * 1024 / 1e6:8.1f} MB")
PY
