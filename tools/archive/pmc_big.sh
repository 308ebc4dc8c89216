#!/bin/bash
# PMC on the loaded-chip workload (3390 frames, --intraPeriod 10): what limits k_me / k_residual8 there?
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcbig}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
W="python3 $GRAFT_REPO_ROOT/tools/quick.py 10 16 3390 stefanlike 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a --output-format csv -- $W > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM -d $OUT/b --output-format csv -- $W > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("k_me<false", "k_residual8", "k_serial_fused", "k_intra_luma32", "k_intra_luma8"):
            if k in n:
                agg[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    wc = max(m.get("SQ_WAVE_CYCLES", 1), 1)
    print(k, g, "launches", len(d["SQ_WAVES"]), "waves", int(w), {
        "valu/wave": round(m.get("SQ_INSTS_VALU", 0) / w), "lds/wave": round(m.get("SQ_INSTS_LDS", 0) / w), "salu/wave": round(m.get("SQ_INSTS_SALU", 0) / w),
        "vmem/wave": round(m.get("SQ_INSTS_VMEM", 0) / w, 1),
        "wavecyc/wave(x4)": round(wc / w), "busy_cyc": int(m.get("SQ_BUSY_CYCLES", 0)), "gui": int(m.get("GRBM_GUI_ACTIVE", 0)),
        "valu_active/wavecyc": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "lds_active/wavecyc": round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3),
        "wait_any": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3), "wait_inst_lds": round(m.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
        "lds_conflict/idx_active": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3), "lds_idx_active": int(m.get("SQ_LDS_IDX_ACTIVE", 0))})
PY
rm -rf $OUT/a $OUT/b
