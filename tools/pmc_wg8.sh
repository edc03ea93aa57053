cd /tmp && export TMPDIR=/tmp
R=/root/repo
for w in ${WLIST:-1 0}; do
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=/tmp/pmc_$w_$(echo $grp | md5sum | cut -c1-6)
  rm -rf $d
  timeout 300 rocprofv3 --pmc $grp -d $d -o x --output-format csv -- python $R/tools/pmc_wg8.py $w > /dev/null 2>&1
  echo "== WG8=$w  $grp"
  python $R/tools/pmc_table.py $d 2>/dev/null | grep -v "^kernel" | cut -c1-200
  python $R/tools/pmc_table.py $d 2>/dev/null | grep "^kernel" | cut -c1-200
done
done
