#!/bin/bash
# Round profile on the GPU box (run from the repo root through gpurun): kernel-trace stats of the bench command, then PMC
# passes (counters in runs of their own, the program directly after `--`), then the traffic calibration.  Output under
# gpurun_out/$1/ ; tools/summarize_profiles.py condenses it into profiles/.
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ICSP_BENCH_DETAIL=$OUT/bench_detail_profiled.json
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --legs ippp > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
# how the dominant kernel's launches overlap in the timed regions of that run (the chip-level figure of the bench line)
python3 $R/tools/overlap_summary.py $OUT/stats $OUT/bench_profiled.json $OUT/overlap.json > $OUT/overlap.txt 2>&1
# the loaded legs: kernel-trace stats and three counter passes each (SKIP_LEGS=1 leaves them out)
if [ -z "${SKIP_LEGS:-}" ]; then
for leg in config4 config4_allintra config5; do
  L="python3 $R/tools/leg_workload.py $leg"
  rocprofv3 --kernel-trace --stats -d $OUT/leg_${leg}_stats --output-format csv -- $L > $OUT/leg_$leg.json 2> $OUT/leg_$leg.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/leg_${leg}_pmc_a --output-format csv -- $L > /dev/null 2> $OUT/leg_${leg}_a.err
  rocprofv3 --pmc WRITE_SIZE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/leg_${leg}_pmc_b --output-format csv -- $L > /dev/null 2> $OUT/leg_${leg}_b.err
  rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/leg_${leg}_pmc_c --output-format csv -- $L > /dev/null 2> $OUT/leg_${leg}_c.err
  find $OUT/leg_${leg}_stats -name "*kernel_trace.csv" -delete
done
fi
W="python3 $R/tools/pmc_workload.py"
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_a --output-format csv -- $W > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --pmc WRITE_SIZE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc_b --output-format csv -- $W > /dev/null 2> $OUT/pmc_b.err
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $OUT/pmc_c --output-format csv -- $W > /dev/null 2> $OUT/pmc_c.err
rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_HIT TCC_MISS -d $OUT/pmc_d --output-format csv -- $W > /dev/null 2> $OUT/pmc_d.err
for p in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  n=$(echo $p | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $p -d $OUT/cal_$n --output-format csv -- $R/tools/calib_traffic.bin > $OUT/cal_$n.out 2> $OUT/cal_$n.err
done
cd $R
python3 tools/summarize_profiles.py $OUT $TAG > $OUT/summary.txt 2>&1
tail -30 $OUT/summary.txt
# the condensed files as made HERE, from this run alone (gpurun merges into gpurun_out/<tag>/ of the caller, where files of earlier
# runs may still lie: summarising there again would average over them)
mkdir -p $OUT/condensed
cp $OUT/overlap.json profiles/${TAG}_overlap.json 2>/dev/null
cp profiles/${TAG}_kernel_stats_*.csv profiles/${TAG}_bench_profiled.json profiles/${TAG}_overlap.json profiles/traffic.json $OUT/condensed/
# keep what is merged back small: the condensed files only
rm -rf $OUT/stats/*/*_agent_info.csv
find $OUT -name "*.csv" -size +8M -delete
