# Everything profiles/ holds for a round, from the same command the driver times (python bench.py), on the GPU box:
#   kernel trace + stats (CSV), per-kernel summary, HBM traffic of the conv kernels (two --pmc passes), PMC of the conv / attention loops
# usage: bash tools/profile_round.sh r02      (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-decode --no-ragged --no-extra --steps 4 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_trace -o r -- $B > gpurun_out/${TAG}_trace_run.log 2>&1
python tools/prof_summary.py gpurun_out/p_trace 13 > gpurun_out/${TAG}_bench_kernel_trace_summary.md 2>&1
python tools/prof_gaps.py gpurun_out/p_trace > gpurun_out/${TAG}_step_gpu_idle_gaps.txt 2>&1
python tools/prof_phases.py gpurun_out/p_trace > gpurun_out/${TAG}_step_phases.txt 2>&1
python tools/prof_chain.py gpurun_out/p_trace > gpurun_out/${TAG}_step_chain.txt 2>&1
python tools/prof_native.py gpurun_out/p_trace > gpurun_out/${TAG}_step_torch_native_kernels.txt 2>&1
cp gpurun_out/p_trace/r_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -o r -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -o r -- $B > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/p_fetch gpurun_out/p_write > gpurun_out/${TAG}_pmc_conv_traffic.json 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/p_a -o r -- python tools/pmc_pipe.py 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_b -o r -- python tools/pmc_pipe.py 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/p_e -o r -- python tools/pmc_c3.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_f -o r -- python tools/pmc_c3.py > /dev/null 2>&1
(echo "# rocprofv3 --pmc, tools/pmc_c3.py: streaming 3x3 kernel c3r_kernel, B=32: layer1 conv2 forward (grid 262144), layer2 conv2 forward and backward-data (grid 131072); values per launch"; python tools/pmc_table.py gpurun_out/p_e; python tools/pmc_table.py gpurun_out/p_f) > gpurun_out/${TAG}_pmc_conv3x3_stream.txt 2>&1
(echo "# rocprofv3 --pmc, tools/pmc_pipe.py: forward of layer3 3x3 (grid 122880: pipe_kernel<CONV,160,256>), layer4 3x3 (102400: <192,128>), layer2 3x3 (307200: glds_kernel<CONV,128,128>), B=32"; python tools/pmc_table.py gpurun_out/p_a; python tools/pmc_table.py gpurun_out/p_b) > gpurun_out/${TAG}_pmc_conv_loops.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/p_c -o r -- python tools/pmc_attn2.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_d -o r -- python tools/pmc_attn2.py > /dev/null 2>&1
(echo "# rocprofv3 --pmc, tools/pmc_attn2.py: DETR encoder attention core, B=32 x 8 heads, 300x300, dh 32, dropout 0.1, no key-padding mask: attn_q_kernel<..,0,..> forward, attn_bwd1_kernel backward (dQ, dK, dV in one launch; round 3: <..,1,..> dQ + attn_kv2_kernel dK/dV) (4 launches each; values are per launch)"; python tools/pmc_table.py gpurun_out/p_c; python tools/pmc_table.py gpurun_out/p_d; python tools/pmc_attn.py; python tools/bench_attn.py; python tools/bench_ln.py; python tools/bench_c1s_linear.py; python tools/bench_attn_qkv.py; python tools/bench_linear_ln.py; python tools/bench_skinny_pf.py; python tools/ab_decode.py) > gpurun_out/${TAG}_pmc_attention.txt 2>&1
rm -rf gpurun_out/p_trace gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_a gpurun_out/p_b gpurun_out/p_c gpurun_out/p_d gpurun_out/p_e gpurun_out/p_f
python tools/bench_body.py > gpurun_out/${TAG}_body_per_launch.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
