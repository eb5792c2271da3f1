"""The driver's contract for bench.py: one JSON line on stdout with the metric of BASELINE.json, the timing fields, and the roofline /
cpu_baseline objects of the hot-path tier.  (A short run: 3 timed steps, no extra records.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
	r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-records'],
					   capture_output=True, text=True, timeout=900, cwd=ROOT)
	assert r.returncode == 0, r.stderr[-2000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, lines
	d = json.loads(lines[0])
	base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
	# BASELINE.json: "deformed vertices x rendered views / sec (fwd+bwd); Chamfer vs ref" -- the throughput clause is the bench metric
	assert d['metric'] == base['metric'].split(';')[0].replace('\u00d7', 'x').strip() and d['unit'] == 'vertices*views/s'
	assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1
	assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic' and d['dtype'].startswith('f32')
	assert d['vs_baseline'] is None   # BASELINE.md holds no published number for this metric
	assert d['value'] > 0 and d['ms_per_step'] > 0
	assert abs(d['value'] - 16 * 6890 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
	assert isinstance(d['config'].get('workload'), str) and 'model' not in d['config']
	rf = d['roofline']
	assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s')
	assert rf['peak'] > 0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0.3 < rf['frac'] < 1.0
	assert len(rf['kernels']) >= 4 and all(k['isolated_us'] > 0 and 0.05 < k['frac'] < 1.0 for k in rf['kernels'])
	assert rf['traffic'] is None or rf['traffic'] > 0
	cb = d['cpu_baseline']
	assert cb['value'] > 0 and cb['unit'] == d['unit'] and cb['cores'] >= 1 and cb['kind'] in ('reference', 'port') and isinstance(cb['sample'], str)


def test_two_rank_bench_path_runs_end_to_end_on_one_gpu():
	"""First execution of everything `bench.py --gpus N` does for N > 1 before the driver's own: spawn_ranks (fresh child
	processes with the launcher's environment; the parent never touches the GPU), init_from_env, the parameter broadcast, the gradient
	arena bucket + one all-reduce per step, the priming phase's cross-rank "go on" all-reduce, barrier-bracketed timing with the MAX over
	ranks, rank 0 alone printing the line.  FIND_BENCH_SHARE_GPU=1 puts both ranks on device 0 with gloo as the transport (this box has
	one GPU; RCCL itself runs in tests/test_gpu_distributed.py) -- the code path is the one the 8-GPU run takes."""
	env = dict(os.environ, FIND_BENCH_SHARE_GPU='1')
	for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
		env.pop(k, None)
	r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-records'],
					   capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr[-3000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
	assert len(lines) == 1, r.stdout[-2000:]
	d = json.loads(lines[0])
	assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
	assert d['config']['parallelism'] == 'dp2' and d['config']['feet_per_gpu'] == 16
	# whole-job aggregate: both ranks' feet over the slowest rank's time
	assert abs(d['value'] - 2 * 16 * 6890 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
	assert 'cpu_baseline' not in d and 'records' not in d   # rank 0 at N = 1 only
	assert d['roofline']['frac'] > 0.3
