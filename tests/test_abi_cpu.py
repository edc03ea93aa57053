"""C-ABI checks that need no GPU: the shared library loads, exports exactly the entry points include/gpv_hip.h declares,
and the ctypes mirrors in gpv-1_amd/hip.py have the C layout (sizes and field offsets, via a gcc-compiled probe)."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'gpv_hip.h')


def declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'#ifdef GPV_TUNING.*?#endif', '', src, flags=re.S)       # tuning-build-only entry points are not the production ABI
    return sorted(set(re.findall(r'\bint\s+(gpv_\w+)\s*\(', src)))


def lib_path():
    sys.path.insert(0, ROOT)
    import gpv1_amd.hip as hip
    if not os.path.exists(hip._LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return hip._LIB_PATH


def test_library_exports_every_declared_entry_point():
    import gpv1_amd.hip as hip
    names = declared()
    assert len(names) == 52, names
    assert sorted(hip.EXPORTS) == names
    lib = ctypes.CDLL(lib_path())
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/gpv_hip.h but not exported'
    lib.gpv_abi_version.restype = ctypes.c_int
    assert lib.gpv_abi_version() == 1
    # every other exported gpv_* symbol is declared
    out = subprocess.run(['nm', '-D', '--defined-only', lib_path()], capture_output=True, text=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith('gpv_')})
    assert exported == names, set(exported) ^ set(names)


def test_production_library_has_no_tuning_code():
    """the timing-ablation instances of the weight-gradient kernel ("results wrong by construction"), the environment
    reads of the launch heuristics and the kernels that did not pay live in the -DGPV_TUNING build only (libgpv_hip_tuning.so)."""
    path = lib_path()
    assert path.endswith('libgpv_hip.so')
    syms = subprocess.run(['nm', '-C', path], capture_output=True, text=True).stdout
    assert '_abl' not in syms and 'ffn_fused' not in syms
    # the library never calls getenv (nm -D --undefined-only lists its imports)
    und = subprocess.run(['nm', '-D', '--undefined-only', path], capture_output=True, text=True).stdout
    assert not re.search(r'\bgetenv\b', und), 'libgpv_hip.so reads the environment'


def test_every_switch_is_documented():
    """INTEGRATION.md section 3 lists every GPV_* switch the tree reads (os.environ in the Python layer, tune_env in
    the native sources) -- regenerate the table with `python tools/list_knobs.py` when this fails."""
    names = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'list_knobs.py'), '--names'], capture_output=True, text=True).stdout.split()
    assert len(names) > 60
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    table = doc[doc.index('<!-- knobs:begin -->'):doc.index('<!-- knobs:end -->')]
    listed = set(re.findall(r'^\| `(GPV_[A-Z0-9_]+)` \|', table, flags=re.M))
    assert sorted(listed) == sorted(names), (sorted(set(names) - listed), sorted(listed - set(names)))


def test_ctypes_structs_match_the_c_layout(tmp_path):
    import gpv1_amd.hip as hip
    pairs = (('gpv_gemm_args', hip.GemmArgs), ('gpv_conv_args', hip.ConvArgs), ('gpv_attn_args', hip.AttnArgs), ('gpv_tt_problem', hip.TTProblem), ('gpv_fold_problem', hip.FoldProblem), ('gpv_tc_problem', hip.TCProblem), ('gpv_image_desc', hip.ImageDesc),
             ('gpv_conv_wgrad_problem', hip.ConvWgradProblem), ('gpv_jpeg_info', hip.JpegInfo), ('gpv_jpeg_desc', hip.JpegDesc))
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for cname, cls in pairs:
        prog.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for f, _ in cls._fields_:
            prog.append(f'  printf(" %zu", offsetof({cname}, {f}));')
        prog.append('  printf("\\n");')
    prog += ['  return 0;', '}']
    c = tmp_path / 'probe.c'
    c.write_text('\n'.join(prog))
    exe = tmp_path / 'probe'
    subprocess.run(['gcc', '-std=c99', '-o', str(exe), str(c)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for (cname, cls), line in zip(pairs, lines):
        vals = line.split()
        assert vals[0] == cname
        assert int(vals[1]) == ctypes.sizeof(cls), (cname, vals[1], ctypes.sizeof(cls))
        offs = [int(v) for v in vals[2:]]
        assert offs == [getattr(cls, f).offset for f, _ in cls._fields_], cname


def test_product_refuses_to_run_without_the_library(monkeypatch, tmp_path):
    """no CPU / eager fallback: a missing library or a CPU tensor is an error, not a silent detour"""
    import importlib
    import torch
    import gpv1_amd.hip as hip
    monkeypatch.setattr(hip, '_LIB', None)
    monkeypatch.setattr(hip, '_LIB_PATH', str(tmp_path / 'nope.so'))
    try:
        hip.lib()
        raise AssertionError('expected RuntimeError')
    except RuntimeError as e:
        assert 'no CPU/eager fallback' in str(e)
    monkeypatch.undo()
    try:
        hip._p(torch.zeros(4))
        raise AssertionError('expected RuntimeError')
    except RuntimeError as e:
        assert 'GPU' in str(e)


def test_the_mirror_refuses_short_epilogue_vectors():
    """The C ABI takes plain pointers and cannot see a short rowscale / bias; the Python mirror has the tensors and refuses before any
    launch (hip.check_epilogue_extents; include/gpv_hip.h gpv_conv_args: rowscale is per output ROW -- per PIXEL in conv modes 0 / 1,
    per Cout in mode 2).  Round 6: a tool passed a [Cout] rowscale to a forward convolution and the kernel read 19200 floats from 64."""
    import pytest
    import torch
    import gpv1_amd.hip as hip
    B, OH, OW, Cout, Cin = 1, 120, 160, 64, 64
    hip.check_epilogue_extents('conv2d mode 0', B * OH * OW, Cout, torch.ones(B * OH * OW), torch.zeros(Cout))
    hip.check_epilogue_extents('conv2d mode 2', Cout, Cin, torch.ones(Cout), None)
    hip.check_epilogue_extents('gemm', 192, 768, None, None)
    with pytest.raises(ValueError, match='rowscale has 64 elements.*19200 needed'):
        hip.check_epilogue_extents('conv2d mode 0', B * OH * OW, Cout, torch.ones(Cout), torch.zeros(Cout))
    with pytest.raises(ValueError, match='bias has 32 elements.*64 needed'):
        hip.check_epilogue_extents('conv2d mode 0', B * OH * OW, Cout, None, torch.zeros(32))
    with pytest.raises(ValueError, match='rowscale'):
        hip.check_epilogue_extents('gemm', 192, 768, torch.ones(191), torch.zeros(768))
