#!/bin/bash
# PMC pass (MFMA busy, waits, LDS) of the batch-128 training step -> gpurun_out/train_pmc/pmc_busy.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/busy -o p -- python3 $R/tools/bench_train.py 128 3 256 > $O/bench.json 2> $O/err.log < /dev/null
cd $R
python3 tools/pmc_summary.py $O/busy > $O/pmc_busy.txt 2> $O/pmc.err < /dev/null
head -12 $O/pmc_busy.txt | cut -c1-330
