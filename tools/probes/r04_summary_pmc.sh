#!/bin/bash
# round 4: SQ counters of the classifier's summary kernel (one rocprofv3 --pmc pass per counter set, no trace summaries beside them)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_summary_pmc; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd /tmp && AB_CLASSES=${CLASSES:-65} timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/$O/sum_set$i -o sum -- python3 $ROOT/tools/probes/ab_summary_wave.py > $ROOT/$O/log_$i.txt 2>&1)
done
python3 tools/sq_counters_report.py $O | tee $O/report.txt
