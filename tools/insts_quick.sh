#!/bin/bash
# Vector-instruction counts per kernel launch (one PMC pass of tools/pmc_workload.py): tools/insts_quick.sh <tag> [ICSP_LIB path]
# through gpurun from the repo root -> gpurun_out/r05/insts_<tag>.txt
set -u
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/iq_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "${2:-}" ] && export ICSP_LIB=$2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d $OUT --output-format csv -- python3 $R/tools/pmc_workload.py > /dev/null 2> $OUT.err
cd $R
python3 - $OUT > gpurun_out/r05/insts_$TAG.txt <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rocclr" in n: continue
        m = re.search(r"(k_\w+(<[^>]*>)?)", n)
        agg[(m.group(1) if m else n[:40], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    f64 = m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0)
    print(f"{k:34s} grid {g:8d} launches {len(d['SQ_WAVES']):3d} waves {int(w):6d}  valu/launch {m.get('SQ_INSTS_VALU',0):12.0f}  valu/wave {m.get('SQ_INSTS_VALU',0)/w:7.1f}  f64/wave {f64/w:6.1f}  cvt/wave {m.get('SQ_INSTS_VALU_CVT',0)/w:5.1f}  lds/wave {m.get('SQ_INSTS_LDS',0)/w:6.1f}  salu/wave {m.get('SQ_INSTS_SALU',0)/w:6.1f}")
PY
rm -rf $OUT
cat gpurun_out/r05/insts_$TAG.txt
