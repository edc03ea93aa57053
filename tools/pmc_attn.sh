cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/pmcA -o r -- python tools/pmc_attn2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcB -o r -- python tools/pmc_attn2.py > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/pmcA > gpurun_out/pmc_attn_A.txt 2>&1
python tools/pmc_table.py gpurun_out/pmcB > gpurun_out/pmc_attn_B.txt 2>&1
rm -rf gpurun_out/pmcA gpurun_out/pmcB
