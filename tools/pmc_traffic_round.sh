cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-decode --no-ragged --no-extra --steps 4 --warmup 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -o r -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -o r -- $B > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/p_fetch gpurun_out/p_write > gpurun_out/r05_pmc_conv_traffic.json 2>&1
rm -rf gpurun_out/p_fetch gpurun_out/p_write
head -12 gpurun_out/r05_pmc_conv_traffic.json
