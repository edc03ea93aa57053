cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 1 0; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/pmcA_$m -o r -- python tools/pmc_pipe.py $m > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcB_$m -o r -- python tools/pmc_pipe.py $m > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/pmcA_$m > gpurun_out/pmcA_$m.txt 2>&1
python tools/pmc_table.py gpurun_out/pmcB_$m > gpurun_out/pmcB_$m.txt 2>&1
rm -rf gpurun_out/pmcA_$m gpurun_out/pmcB_$m
done
