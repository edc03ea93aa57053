"""Every GPV_* switch the tree reads, as the markdown table of INTEGRATION.md section 5:
   * host side (gpv-1_amd/*.py, bench.py): os.environ reads -- policy of the Python layer;
   * native side (gpv-1_amd/csrc): tune_env(...) reads -- compiled to their defaults in libgpv_hip.so, live only in the -DGPV_TUNING build
     (make -C gpv-1_amd/csrc tuning, loaded with GPV_TUNING_LIB=1 by tools/).
usage: python tools/list_knobs.py            (prints the table)
       python tools/list_knobs.py --names    (one name per line: tests/test_abi_cpu.py checks INTEGRATION.md against it)"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# one line for the reads whose source has no comment next to them
NOTES = {
    'GPV_ADAMW_VEC': 'fused AdamW with 16-byte accesses (0: the scalar kernel)',
    'GPV_ATTN_KV_OLD': 'dK / dV on the round-2 kernel (A/B of attn_kv2_kernel)',
    'GPV_ATTN_QKV': 'DETR self-attention with the q | k | v projections inside the attention launch (gpv_attention_qkv_fwd), from 64 (image, head) pairs up',
    'GPV_ATTN_SPLIT': 'query tiles of one (image, head) split over several workgroups (few-head shapes)',
    'GPV_BERT_VOCAB': 'path of bert-base-uncased vocab.txt (string queries)',
    'GPV_BERT_WEIGHTS': 'path of the frozen BERT weights (.bin / .safetensors); random init without',
    'GPV_BERT_NO_PIPE_SMALL': 'the frozen BERT branch without gemm_pipe.hip small-M configurations (shorter as dependent graph nodes)',
    'GPV_BOUNDARY': 'roi: RoI features from the DETR decoder output (the reference); other values are debugging cuts',
    'GPV_C1C_BLOCKS': 'workgroups of the chained 1x1 kernel (0: heuristic)',
    'GPV_C1S': 'streaming 1x1 policy: 0 never, 1 heuristic, 2 wherever legal (-1: leave the default)',
    'GPV_C1S_BLOCKS': 'workgroups of the streaming 1x1 kernel (0: one or two per CU by LDS size)',
    'GPV_C3D2_ROWS': 'output rows a wave walks in the stride-2 streaming 3x3 backward-data (0: heuristic)',
    'GPV_C3R_ROWS': 'output rows a wave walks in the streaming 3x3 kernel (0: heuristic)',
    'GPV_C3S': 'streaming 3x3 policy: 0 never, 1 heuristic, 2 wherever legal (-1: leave the default)',
    'GPV_C3S_BLOCKS': 'workgroups of the streaming 3x3 kernel (0: heuristic)',
    'GPV_C3S_WAVES': 'waves per workgroup of the streaming 3x3 kernel (0: 8)',
    'GPV_CAPTURE_TRACE': 'print every hipGraph capture / replay decision of the trainer',
    'GPV_COATT_BRANCH': 'co-attention language stream on a side stream / graph branch beside the vision stream (ops.Branch)',
    'GPV_COATT_QKV': 'co-attention q | k | v as one GEMM over concatenated weights',
    'GPV_CONV_SPLIT': 'split-K for the register-staged conv weight gradient (0: never)',
    'GPV_DEBUG_SYNC': 'synchronise and check after every launch (also disables the graphs in bench.py)',
    'GPV_DX_MIRROR': 'keep W^T mirrors of the Linear weights for the backward-data GEMMs',
    'GPV_FUSED_CRITERION': 'caption criterion inside the captured body',
    'GPV_FUSED_STEM': 'conv1 + bn1 + relu + maxpool as gpv_stem_pool',
    'GPV_FUSED_TAIL': 'conv3 + downsample (+ next conv1) of a stage head as one launch',
    'GPV_GLDS_DEPI': 'register epilogue of the direct-to-LDS GEMM (0: the fp32 LDS image)',
    'GPV_GRAD_CHAIN': 'layer-input gradients summed in GEMM epilogues instead of autograd adds',
    'GPV_GRAPHS_STRICT': 'a failed capture raises instead of falling back to eager steps',
    'GPV_HEAD_LATE': 'DETR head parameters in the last all-reduce bucket (they finish first in the backward)',
    'GPV_HIP_LIB': 'path of libgpv_hip.so (default: next to the package)',
    'GPV_INFER_BERT_BRANCH': 'frozen BERT on a graph branch during inference',
    'GPV_NO_GRAPHS': 'bench.py: eager steps only',
    'GPV_OVERLAP': 'all-reduce buckets overlapped with the backbone backward (0: after the pass)',
    'GPV_PREP_BRANCH': 'conv weight casts / copies on a branch of F1',
    'GPV_MASK_BITS': 'ReLU masks of the layer2 / layer3 block outputs as one bit per element (written by the conv3 launch, read by the next conv1 backward-data)',
    'GPV_FRESH_STREAMS': 'the side streams of a capture owner (a trainer body: BERT / weight / ops.Branch branches; an inference graph: its ops.Branch branch) are hipStreams of its own, destroyed with its graphs (ops.owned_stream); 0: torch.cuda.Stream(), i.e. one of 32 pooled streams handed out round-robin -- recycled across destroyed graphs: a segmentation fault after 8 - 16 body evictions (tools/soak_evict.py)',
    'GPV_DEBUG_STREAMS': 'debugging the capture lifetimes: owned streams are kept instead of destroyed, ops.foreign_capturing() reports side streams of OTHER owners found in capture mode before a capture ends, ops.report_unpinned_leaves() lists the AccumulateGrad nodes of a body that are not the trainer-pinned ones',
    'GPV_TEXT_EARLY': 'teacher forcing: target embedding, vocabulary classifiers and the first text-decoder self-attention forked beside the co-attention stage',
    'GPV_PROJ_LN_MIN_ROWS': 'fewest rows for which the projection rides in the LayerNorm launch (below: GEMM + LayerNorm, faster as graph nodes up to ~1200 rows)',
    'GPV_PROJ_LN': 'attention out-projection inside the LayerNorm launch (gpv_linear_layernorm_fwd)',
    'GPV_RCCL_HIGH_PRIO': 'RCCL stream created with high priority',
    'GPV_S2_DGRAD_SPLIT': '1x1 stride-2 backward-data as GEMM on the sampled pixels + fill kernel',
    'GPV_STEM_BLOCKS': 'workgroups of the fused stem (0: heuristic)',
    'GPV_STEM_ROWS': 'conv rows a wave walks in the fused stem (0: heuristic)',
    'GPV_TWO_PER_CU': 'two half-width tiles per CU in the direct-to-LDS GEMM (0: one 8-wave tile)',
    'GPV_WG8H_C': 'ramp (in k-tiles) the half-width weight-gradient launch adds per round when it picks its work-unit length',
    'GPV_WG8_KT': 'k-tiles per work unit of the 256 x 256 eight-phase weight-gradient launch',
    'GPV_WGRAD_FLUSH': 'where the deferred Linear weight gradients are issued (detr: behind the DETR transformer backward)',
    'GPV_WGRAD_SPLIT': 'grouped Linear weight gradients in three calls along the backward (0: one)',
    'GPV_WGRAD_STREAM': 'ungrouped conv weight gradients on a side stream',
    'GPV_WG_ABL': 'timing ablations of the 128 x 128 weight-gradient core (WRONG results; tuning build only)',
    'GPV_ZERO_IN_GRAPH': 'gradient buffer cleared on a branch of F2 inside the graph',
}


def comment_of(line, mark, before=()):
    i = line.find(mark)
    c = line[i + len(mark):].strip() if i >= 0 else ''
    if not c:                                   # no trailing comment: the comment-only lines right above the read
        up = []
        for b in reversed(before):
            t = b.strip()
            if t.startswith(mark):
                up.insert(0, t[len(mark):].strip())
            else:
                break
        c = ' '.join(up)
    c = re.sub(r'\s+', ' ', c).replace('|', '\\|')
    return (c[:157] + '...') if len(c) > 160 else c


def collect():
    rows = {}
    for f in sorted(glob.glob(os.path.join(ROOT, 'gpv-1_amd', '*.py')) + [os.path.join(ROOT, 'bench.py')]):
        lines = open(f).read().splitlines()
        for no, line in enumerate(lines, 1):
            for m in re.finditer(r"environ\.get\('(GPV_[A-Z0-9_]+)'(?:, *'([^']*)')?", line):
                name, dflt = m.group(1), m.group(2)
                rows.setdefault(name, ('host', dflt if dflt is not None else '(unset)', '%s:%d' % (os.path.relpath(f, ROOT), no), comment_of(line, '#', lines[max(0, no - 4):no - 1])))
    for f in sorted(glob.glob(os.path.join(ROOT, 'gpv-1_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'gpv-1_amd', 'csrc', '*.h'))):
        lines = open(f).read().splitlines()
        for no, line in enumerate(lines, 1):
            for m in re.finditer(r'tune_env\("(GPV_[A-Z0-9_]+)", *(-?[0-9]+)\)', line):
                rows.setdefault(m.group(1), ('tuning build', m.group(2), '%s:%d' % (os.path.relpath(f, ROOT), no), comment_of(line, '//', lines[max(0, no - 4):no - 1])))
    return rows


if __name__ == '__main__':
    rows = collect()
    if '--names' in sys.argv:
        print('\n'.join(sorted(rows)))
    else:
        print('| knob | read by | default | where | note (the source line\'s comment) |')
        print('|---|---|---|---|---|')
        for name in sorted(rows):
            kind, dflt, where, note = rows[name]
            note = note or NOTES.get(name, '').replace('|', '\\|')
            print('| `%s` | %s | `%s` | `%s` | %s |' % (name, kind, dflt, where, note))
