"""The launch planner of ide3d_modconv2d through its host-only query (include/ide3d_hip.h: ide3d_modconv_plan; no GPU needed): which kernel
family, tile, split-K and grid the benchmark's layers get.  The rules these tests pin are the ones DESIGN.md section 4.3 measured: under
exclusive residency (one workgroup per CU) a launch costs ceil(workgroups / 256) workgroup times, so the planner aims at whole rounds of the
8-wave forms; the transposed layers run on the h x w position grid (strip plan) when their epilogue is the plain one; the heads go to the
resident / small-map / one-tile kernels by shape.  GPU parity of every one of these forms: tests/test_gpu_conv_arith.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))

from torch_utils import hip_plugin  # noqa: E402

CU = 256
pytestmark = pytest.mark.skipif(any(k.startswith(('IDE3D_MODCONV_', 'IDE3D_SP_', 'IDE3D_HEAD_')) for k in os.environ),
                                reason='planner experiment switches are set in the environment')


def plan(**kw):
    kw.setdefault('n', 4)
    kw.setdefault('arith', 6)
    return hip_plugin.modconv_plan(**kw)


@pytest.mark.parametrize('cin,res,tile,waves,split_k,wgs', [
    (512, 4, (8, 16), 4, 16, 256), (512, 8, (8, 16), 4, 16, 256), (512, 16, (8, 16), 4, 8, 256), (512, 32, (16, 16), 8, 4, 256),
    (512, 64, (16, 16), 8, 1, 256), (256, 128, (16, 16), 8, 1, 512), (128, 256, (16, 16), 8, 1, 1024)])
def test_3x3_layers_of_the_backbone_run_whole_rounds_on_the_split_loop(cin, res, tile, waves, split_k, wgs):
    p = plan(cin=cin, cout=cin, h=res, w=res)
    assert p['kind'] == 'split' and p['parts'] == 3 and p['f16'] == 0 and p['rows'] == 128
    assert (p['tile_h'], p['tile_w']) == tile and p['waves'] == waves and p['split_k'] == split_k
    assert p['workgroups'] == wgs and wgs % CU == 0


def test_64_row_3x3_layer_uses_the_32x16_tile():
    p = plan(cin=64, cout=64, h=512, w=512)
    assert (p['kind'], p['tile_h'], p['tile_w'], p['rows'], p['waves'], p['workgroups']) == ('split', 32, 16, 64, 8, 2048)
    small = plan(cin=64, cout=64, h=64, w=64)                 # too few 32 x 16 tiles for two rounds: the 8 x 16 / 16 x 16 forms
    assert small['tile_h'] != 32


@pytest.mark.parametrize('cin,cout,res,tile,rows,wgs', [
    (512, 512, 32, (8, 16), 64, 256), (512, 256, 64, (8, 16), 128, 256), (256, 128, 128, (16, 16), 128, 512), (128, 64, 256, (16, 16), 64, 1024),
    (32, 128, 128, (8, 16), 128, 512)])
def test_transposed_layers_run_the_strip_plan_at_whole_rounds(cin, cout, res, tile, rows, wgs):
    p = plan(cin=cin, cout=cout, h=res, w=res, mode=2, epilogue='plain')
    assert p['kind'] == 'split' and p['transposed_all_class'] == 1 and p['strip'] == 1 and p['split_k'] == 1 and p['waves'] == 8
    assert (p['tile_h'], p['tile_w']) == tile and p['rows'] == rows and p['workgroups'] == wgs and wgs % CU == 0
    # an epilogue with noise / bias / activation belongs to the main kernel: no strip kernel could finish it
    q = plan(cin=cin, cout=cout, h=res, w=res, mode=2, epilogue='conv')
    assert q['strip'] == 0


def test_row_parity_pairs_where_the_even_rows_fill_a_round(monkeypatch):
    """Round 5: one output-row parity per workgroup (modconv_split_pair_kernel: 16 x 16 positions x 128 rows, twice the workgroups of the
    same position grid) where >= 256 even-row workgroups exist; IDE3D_MODCONV_PAIR=0 restores the all-class 8 x 16 form, =2 forces the
    form wherever it exists (the GPU tests compare the two bit for bit)."""
    args = dict(cin=256, cout=128, h=128, w=128, mode=2, epilogue='plain')
    pair = plan(**args)
    assert (pair['tile_h'], pair['tile_w'], pair['rows'], pair['waves'], pair['workgroups'], pair['strip'], pair['split_k']) == (16, 16, 128, 8, 512, 1, 1)
    assert plan(**dict(args, epilogue='conv'))['tile_h'] != 16                # the pairs need the strip plan, i.e. the plain epilogue
    assert plan(**dict(args, n=1))['tile_h'] == 8                             # 64 even-row workgroups: no round to fill
    assert plan(cin=512, cout=256, h=64, w=64, mode=2, epilogue='plain')['tile_h'] == 8       # 128 per parity
    monkeypatch.setenv('IDE3D_MODCONV_PAIR', '0')
    off = plan(**args)
    assert (off['tile_h'], off['workgroups']) == (8, 512)
    monkeypatch.setenv('IDE3D_MODCONV_PAIR', '2')
    forced = plan(cin=64, cout=128, h=20, w=24, n=2, mode=2, epilogue='plain')
    assert (forced['tile_h'], forced['rows'], forced['strip'], forced['split_k']) == (32, 64, 1, 1) and forced['workgroups'] == 2 * 2 * 1 * 2 * 2


def test_batch_1_transposed_64_row_layer_runs_one_round_of_16x16():
    """Round 5: 128 -> 64 in@256 at batch 1 = 256 eight-wave workgroups of 16 x 16 positions on the strip grid (was 1024 four-wave workgroups of
    4 x 16: 92 -> 67 us)."""
    p = plan(n=1, cin=128, cout=64, h=256, w=256, mode=2, epilogue='plain')
    assert (p['kind'], p['tile_h'], p['tile_w'], p['rows'], p['waves'], p['workgroups'], p['strip'], p['split_k']) == ('split', 16, 16, 64, 8, 256, 1, 1)


def test_low_resolution_transposed_layers_use_two_team_workgroups_with_split_k():
    for res, split_k, wgs in ((4, 8, 256), (8, 8, 384), (16, 3, 480)):
        p = plan(cin=512, cout=512, h=res, w=res, mode=2, epilogue='plain')
        assert (p['kind'], p['tile_h'], p['tile_w'], p['rows'], p['waves'], p['strip']) == ('split_teams', 4, 16, 64, 8, 0)
        assert (p['split_k'], p['workgroups']) == (split_k, wgs)


def test_arithmetic_selects_the_loop():
    for mode, ep in ((0, 'conv'), (2, 'plain')):
        p = plan(cin=128, cout=128, h=256, w=256, mode=mode, epilogue=ep, arith=1)
        assert p['kind'] == 'fp32' and p['parts'] == 0                        # exact fp32 products: the fp32 matrix loop
    assert plan(cin=128, cout=128, h=256, w=256, arith=3)['parts'] == 2       # bf16x3
    no_bound = plan(cin=128, cout=128, h=256, w=256, arith=16)
    assert (no_bound['parts'], no_bound['f16']) == (3, 0)                     # f16x3 without the producer's amax runs bf16x6
    bound = plan(cin=128, cout=128, h=256, w=256, arith=16, x_amax=True)
    assert (bound['parts'], bound['f16']) == (2, 1)
    narrow = plan(cin=16, cout=128, h=256, w=256)                             # fewer than 3 K chunks: prologue + epilogue dominate
    assert narrow['kind'] == 'fp32'


@pytest.mark.parametrize('cin,cout,res,kind,wgs', [
    (512, 192, 4, 'head_small', 192), (512, 192, 8, 'head_small', 192), (512, 192, 16, 'head_small', 192), (512, 192, 32, 'fp32', None),
    (256, 192, 128, 'head_split', 256), (128, 192, 256, 'head_resident', 256), (128, 22, 256, 'head_resident', 256), (64, 22, 512, 'head_resident', 256)])
def test_heads_are_routed_by_shape(cin, cout, res, kind, wgs):
    p = plan(cin=cin, cout=cout, h=res, w=res, k=1, per_image=True, epilogue='head')
    assert p['kind'] == kind
    if wgs is not None:
        assert p['workgroups'] == wgs
    if kind == 'head_resident':
        assert p['waves'] == 8 and p['rows'] == (32 if cout <= 32 else 192)


def test_heads_in_exact_fp32_and_at_batch_1():
    assert plan(cin=128, cout=192, h=256, w=256, k=1, per_image=True, epilogue='head', arith=1)['kind'] == 'fp32'
    assert plan(cin=512, cout=192, h=8, w=8, k=1, per_image=True, epilogue='head', arith=1)['kind'] == 'head_small'     # fp32 FMAs: every arithmetic
    # one image: 2048 tiles of 32 pixels for 2048 waves — fewer than two per wave, so the one-tile kernel
    assert plan(n=1, cin=128, cout=192, h=256, w=256, k=1, per_image=True, epilogue='head')['kind'] == 'head_split'
    assert plan(n=1, cin=64, cout=22, h=512, w=512, k=1, per_image=True, epilogue='head')['kind'] == 'head_resident'


def test_bad_shapes_are_errors():
    with pytest.raises(RuntimeError):
        plan(cin=0, cout=8, h=4, w=4)
    with pytest.raises(RuntimeError):
        plan(cin=8, cout=8, h=4, w=4, k=5)


def test_per_image_transposed_layers_never_take_the_strip_plan():
    """`tconv_strip_kernel` reads ONE weight tensor; a launch with per-image weights [n, cout, cin, 3, 3] must stay on the
    (h + 1) x (w + 1) grid in every arithmetic (the GPU counterpart: tests/test_gpu_ops.py::test_modconv2d_transposed_per_image_weights)."""
    for arith in (0, 1, 6, 16):
        for shape in ((4, 128, 128, 128, 128), (4, 512, 256, 64, 64), (1, 64, 64, 32, 32)):
            p = hip_plugin.modconv_plan(*shape, mode=2, per_image=True, arith=arith, epilogue='plain')
            assert p['strip'] == 0 and p['kind'] == 'fp32', (arith, shape, p)
    assert hip_plugin.modconv_plan(4, 128, 128, 128, 128, mode=2, per_image=False, epilogue='plain')['strip'] == 1


def test_lowres_group_extent_by_batch_size():
    """`ide3d_lowres_layers_supported` (host only): the low-resolution block group covers layers while all images' halo'd input maps of a layer,
    as bf16 pieces of one 32-channel K slice, fit one CU's LDS beside the 55 KB weight slice (csrc/lowres.hip).  Layers of the 512-wide backbone
    from 4^2: conv1@4, up@8, conv1@8, up@16, conv1@16, up@32, conv1@32."""
    from torch_utils import hip_plugin
    ups = [1, 2, 1, 2, 1, 2, 1]
    fit = lambda n, arith=6: hip_plugin.LowresPlugin.layers_supported(n, 512, 4, ups, arith)
    assert fit(1) == 6          # through the up-sampling layer to 32^2 (its input is 16^2: 324 slots); conv1@32 reads 34^2 = 1156 slots
    assert fit(2) == 4 and fit(3) == 4 and fit(4) == 4          # through up@16 (input 8^2: n x 100 slots); conv1@16 reads n x 324
    assert fit(5) == 4 and fit(8) == 2          # 8 images: conv1@4 and up@8 (8 x 36 slots); conv1@8 would read 800
    assert fit(1, arith=3) >= 6          # bf16x3: two pieces per value, more room
    assert fit(4, arith=1) == 0 and fit(4, arith=16) == 0          # exact fp32 products / f16x3 have no such form: per-layer path
    assert hip_plugin.LowresPlugin.layers_supported(4, 500, 4, ups, 6) == 0          # C must be a multiple of 32
