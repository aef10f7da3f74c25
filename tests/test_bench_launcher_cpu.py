"""bench.py's launcher and N > 1 protocol without a GPU: `--dry-run-cpu` runs the same step / barrier / MAX-over-ranks / gather /
rank-0-only JSON code on CPU tensors (tiny generator, gloo).  Checks what the driver relies on: a plain
`python bench.py --gpus 2` launches itself as 2 ranks and prints exactly ONE JSON line, from rank 0."""

import json
import os
import subprocess
import sys

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
