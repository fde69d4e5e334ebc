"""GPU: bench.py's command line at the edges the driver may use -- `--warmup 0`, a handful of steps.  (Round 6: `--warmup 0` died with an
UnboundLocalError -- the reference result of the reproducibility checks was taken from the last warm-up step -- and would have fallen
back from the captured-graph measurement to the eager one.)  The line must be the headline metric measured through graph replays."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_with_zero_warmup_and_two_steps():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '0', '--no-cpu-baseline', '--train-probe', '0'],
                       cwd=ROOT, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = lines[0]
    assert j['steps'] == 2 and j['warmup'] == 0 and j['n_gpus'] == 1 and j['unit'] == 'images/s' and j['higher_is_better'] is True
    assert j['value'] > 50 and j['ms_per_step'] > 0 and j['dtype'] == 'f32' and j['vs_baseline'] is None
    assert isinstance(j.get('graph_replay_ms'), float), j.get('graph_replay_ms')      # not 'failed: ...': the replays were measured
    assert j['replays_identical_to_eager'] is True
    rf = j['roofline']
    assert rf['bound'] == 'mfma' and 0.05 < rf['frac'] < 1.0 and rf['peak'] == 2500.0 and rf['launches'] >= 2


def test_fp16_model_line_from_the_child_process():
    """`other_configs['fp16 model']` is measured by `bench.py --half-config 1` in a child process under a timeout (a half-precision model is
    one library flag away from a device stall, docs/notebook/round6.md 8): the child prints one compact line."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--half-config', '1', '--pipeline', '2'], cwd=ROOT, env=env, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert 'float16' in j['workload'] and j['images_in_flight'] == 2 and j['value'] > 50 and j['detections'] > 100
    assert j['library_deterministic_mode'] is False
