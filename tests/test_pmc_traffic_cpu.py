"""tools/pmc_traffic.py's kernel classification (bench.py measures roofline.traffic live with it): the streaming 1x1 kernel serves both the
backbone's convolutions and -- its LIN instances -- the transformer's K = 256 linear GEMMs; only the former count as conv traffic.  Round 6
added a trailing template argument (BITS) and the old suffix test silently swapped the two sets: names as rocprofv3 prints them."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location('pmc_traffic', os.path.join(ROOT, 'tools', 'pmc_traffic.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_streaming_kernel_instances_are_classified_by_their_lin_argument():
    m = _mod()
    conv = ['void gpvk::(anonymous namespace)::c1s_kernel<256, 256, true, false, false, false, 1, true>(gpvk::GemmK, int)',
            'void gpvk::(anonymous namespace)::c1s_kernel<128, 256, true, true, false, false, 2, true>(gpvk::GemmK, int)',
            'void gpvk::(anonymous namespace)::c1s_kernel<512, 128, false, false, false, false, 1, false>(gpvk::GemmK, int)',
            'void gpvk::(anonymous namespace)::c1s_kernel<64, 64, false, false, true, false, 1, false>(gpvk::GemmK, int)',
            'void gpvk::glds_halo_kernel<160>(gpvk::GemmK)', 'gpvk::wg8h_group_kernel(gpvk::WgGroupK)', 'void gpvk::stem_pool_kernel<8>(gpvk::StemK)']
    lin = ['void gpvk::(anonymous namespace)::c1s_kernel<256, 256, false, false, false, true, 1, false>(gpvk::GemmK, int)',
           'void gpvk::(anonymous namespace)::c1s_kernel<256, 256, false, true, false, true, 1, false>(gpvk::GemmK, int)',
           'void gpvk::(anonymous namespace)::c1s_kernel<256, 256, true, true, false, true, 1, false>(gpvk::GemmK, int)']
    other = ['ln_fwd_kernel<__hip_bfloat16, 2, false>', 'void gpvk::attn_qkv_kernel<20, true>(AttnK, QkvK)', 'glds_tt_group_kernel(GroupK)']
    assert all(m.is_conv(n) for n in conv)
    assert not any(m.is_conv(n) for n in lin + other)
    assert m.is_conv_helper('s2_dgrad_fill_kernel(...)') and m.is_conv_helper('gpvk::wgrad_group_reduce_kernel(gpvk::WgRedK)')
