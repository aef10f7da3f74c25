"""bench.py's launcher and N > 1 protocol without a GPU: `--dry-run-cpu` runs the same step / barrier / MAX-over-ranks / gather /
rank-0-only JSON code on CPU tensors (tiny generator, gloo).  Checks what the driver relies on: a plain
`python bench.py --gpus 2` launches itself as 2 ranks and prints exactly ONE JSON line, from rank 0."""

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--steps', '1', '--warmup', '1', '--blocks', '3', '--dry-run-cpu']


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f'expected exactly one stdout line, got {len(lines)}: {stdout[:2000]}'
    return json.loads(lines[0])


def test_launch_ranks_function_two_gloo_ranks():
    sys.path.insert(0, ROOT)
    import bench
    rc, out = bench.launch_ranks(2, ['--gpus', '2'] + ARGS, capture=True)
    assert rc == 0
    rec = _one_json_line(out)
    assert rec['n_gpus'] == 2 and rec['rccl_ranks'] == 2 and rec['steps'] == 1 and rec['warmup'] == 1
    assert rec['config']['global_batch'] == 8 and rec['scaling'] == 'weak'
    assert len(rec['frames_per_s_by_rank']) == 2 and all(v > 0 for v in rec['frames_per_s_by_rank'])
    assert rec['gather_ms'] > 0 and rec['timing']['blocks'] == 3
    assert rec['timing']['ms_per_step_min'] <= rec['ms_per_step'] <= rec['timing']['ms_per_step_max']
    assert 'DRY RUN' in rec['metric']
    # the double-buffered asynchronous gather: contents and order checked by rank 0 against its own render of every rank's inputs
    assert rec['gather_check']['ok'] is True and rec['gather_check']['steps'] == 3 and rec['gather_check']['ranks'] == 2
    assert rec['gather_check']['checksum_mismatches'] == 0 and rec['gather_check']['checksummed_steps'] == rec['steps'] * rec['timing']['blocks']      # every timed step
    ov = rec['gather_overlap']
    assert ov['ms_per_step_overlapped'] > 0 and ov['ms_per_step_blocking_gather'] > 0 and ov['gather_ms_alone'] > 0


def test_blocking_gather_flag_two_gloo_ranks():
    sys.path.insert(0, ROOT)
    import bench
    rc, out = bench.launch_ranks(2, ['--gpus', '2', '--blocking-gather'] + ARGS, capture=True)
    assert rc == 0
    rec = _one_json_line(out)
    assert rec['n_gpus'] == 2 and rec['gather_check']['ok'] is True


def test_plain_invocation_self_launches():
    """`python bench.py --gpus 2` with no torch.distributed environment (what the driver may run)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + ARGS, env=env, stdout=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0
    rec = _one_json_line(res.stdout)
    assert rec['n_gpus'] == 2 and rec['value'] > 0


def test_single_process_dry_run():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + ARGS, stdout=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0
    rec = _one_json_line(res.stdout)
    assert rec['n_gpus'] == 1 and 'rccl_ranks' not in rec and rec['timing']['blocks'] == 3


def _strict(line):
    """json.loads that rejects NaN / Infinity / -Infinity (what a strict driver-side parser does)."""
    def bad(tok):
        raise ValueError(f'non-standard JSON constant {tok}')
    return json.loads(line, parse_constant=bad)


def test_dry_run_line_is_compact_and_strict_json():
    """round 3's 20 KB line was not parsed by the driver.  The line is < 4096 bytes, ASCII, one line, strict JSON."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + ARGS, stdout=subprocess.PIPE, timeout=600)
    assert res.returncode == 0
    raw = res.stdout
    assert raw.endswith(b'\n') and raw.count(b'\n') == 1
    assert len(raw) < 4096, len(raw)
    raw.decode('ascii')
    rec = _strict(raw.decode())
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in rec, k
    assert rec['config']['conv_arithmetic'] == 'fp32' and rec['config']['conv_arithmetic_is_library_default'] is True
    assert 'workload' in rec['config'] and 'model' not in rec['config']


def test_compact_line_of_a_full_size_gpu_record():
    """The N = 1 GPU line carries every optional object; build one with worst-case field widths and check the limit, the strictness
    (NaN -> null) and the drop order when it would not fit."""
    sys.path.insert(0, ROOT)
    import bench
    big = 123456.789012
    out = {
        'metric': '512x512 RGB+seg frames/s @96 depth samples (whole job)', 'value': big, 'unit': 'frames/s', 'value_fp32_exact': big,
        'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': big, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': bench.ARITH_DTYPE_SHORT['bf16x6'], 'data': 'synthetic',
        'config': {'workload': 'w' * 180, 'global_batch': 4, 'parallelism': 'dp1', 'conv_arithmetic': 'bf16x6', 'conv_arithmetic_is_library_default': False},
        'timing': {'blocks': 5, 'reported': 'median block', 'ms_per_step_min': big, 'ms_per_step_max': big},
        'conv_tflops': big, 'hip_graph': True, 'parity_ok': True,
        'parity': {'max_rel_err': {'img': 2.34e-06, 'seg': 2.84e-06}, 'tol_rel': 2e-05, 'vs': 'tests/golden/bench_parity.npz (CPU oracle)'},
        'dropin_b1': {'frames_per_s': big, 'ms_per_image': big, 'images': 24, 'what': 'x' * 90},
        'by_conv_arithmetic': {k: {'frames_per_s': big, 'parity_ok': True} for k in ('fp32', 'bf16x6', 'f16x3', 'bf16x3')},
        'roofline': {'kernel': 'triplane_sample_tile_pc_kernel', 'bound': 'hbm', 'achieved': big, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.5591,
                     'traffic': 258900000, 'traffic_measured_in_this_run': False, 'bytes_per_launch': 320864256, 'avg_launch_us': 71.83, 'timed_launches': 100},
        'roofline_worst': [{'kernel': 'k' * 60, 'bound': 'split:f16x3', 'frac': 0.123, 'us': big} for _ in range(5)], 'roofline_rows': 55,
        'cpu_baseline': {'value': 1.001, 'unit': 'frames/s', 'cores': 16, 'kind': 'port', 'sample': 's' * 160},
        'parity_live': {'vs': 'v' * 60, 'max_rel_err': {'img': float('nan'), 'seg': 3.5e-06}, 'tol_rel': 2e-05, 'ok': True},
    }
    line = bench.compact_line(out)
    assert len(line) < 4096 and '\n' not in line
    rec = _strict(line)
    assert rec['parity_live']['max_rel_err']['img'] is None          # NaN never reaches the line
    assert 'dropped_for_size' not in rec and rec['roofline']['frac'] == 0.5591 and rec['cpu_baseline']['kind'] == 'port'
    # a record that cannot fit loses optional objects only, never the contract keys / roofline / cpu_baseline
    out['roofline_worst'] = [{'kernel': 'k' * 60, 'frac': 0.1} for _ in range(80)]
    rec = _strict(bench.compact_line(out))
    assert 'roofline_worst' in rec['dropped_for_size'] and 'roofline' in rec and 'cpu_baseline' in rec and rec['value'] == big


def test_live_pmc_passes_degrade_to_a_reason():
    """`bench.live_pmc` (the rocprofv3 counter passes bench.py runs itself for `roofline.traffic` / `roofline_step.mfma_busy_frac`) never raises and
    never hangs the bench: without a GPU, without the profiler or without rows for the kernel it returns (None, reason) and the committed figure is
    quoted with `traffic_measured_in_this_run: false`."""
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.time()
    vals, why = bench.live_pmc([('FETCH_SIZE',)], ['-c', 'print(1)'], 'no_such_kernel', timeout_s=60)
    assert vals is None and isinstance(why, str) and why
    traffic, why = bench.live_gather_traffic()
    assert traffic is None and isinstance(why, str) and why
    busy, why = bench.live_mfma_busy('modconv_split_kernel<0, 1, 16, 3, 2, 8, 0>', 'bf16x6')
    assert busy is None and isinstance(why, str) and why
    assert time.time() - t0 < 170
