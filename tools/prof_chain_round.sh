cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_trace -o r -- python bench.py --no-cpu-baseline --no-decode --no-ragged --no-extra --steps 4 --warmup 3 > /dev/null 2>&1
python tools/prof_chain.py gpurun_out/p_trace > gpurun_out/r05_chain.txt 2>&1
python tools/prof_phases.py gpurun_out/p_trace > gpurun_out/r05b_step_phases.txt 2>&1
rm -rf gpurun_out/p_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_g -o g -- python tools/prof_greedy.py run > /dev/null 2>&1
python tools/prof_greedy.py gpurun_out/p_g > gpurun_out/r05_greedy_bs1_nodes.txt 2>&1
rm -rf gpurun_out/p_g
