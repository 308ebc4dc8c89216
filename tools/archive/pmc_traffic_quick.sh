#!/bin/bash
# HBM-side bytes of the dominant intra kernel only (two PMC passes of tools/pmc_workload.py): through gpurun from the repo root
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/tq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W="python3 $R/tools/pmc_workload.py"
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ TCC_EA0_WRREQ_64B -d $OUT/w --output-format csv -- $W > /dev/null 2> $OUT/w.err
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/r --output-format csv -- $W > /dev/null 2> $OUT/r.err
cd $R
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "tq")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "rocclr" in k: continue
        name = k.split("(")[0].split("::")[-1][:28] + "@" + r["Grid_Size"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
P, NMB = 352 * 288, 396
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rd = 32 * m.get("TCC_EA0_RDREQ_32B", 0) + 64 * m.get("TCC_EA0_RDREQ_64B", 0) + 128 * m.get("TCC_EA0_RDREQ_128B", 0)
    wr = m.get("WRITE_SIZE", 0) * 1024
    line = f"{k:44s} n={len(next(iter(d.values()))):3d} read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB"
    if "intra_luma" in k:
        fr = {57600: 300, 153600: 300, 21120: 30}.get(int(k.split('@')[1]))
        if fr: line += f"   algorithmic read {fr*P/1e6:.2f} write {fr*(3*P+8*NMB)/1e6:.2f}  total x{(rd+wr)/(fr*(4*P+8*NMB)):.3f}"
    print(line)
PY
