#!/usr/bin/env python3
"""Headline benchmark of the FIND hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (spawns its own N ranks, one per GPU, when N > 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric (BASELINE.json): deformed vertices x rendered views / second, forward+backward.

Headline workload = the reference's own training configuration, north_star's target: one network-stage step of
cfgs/train_3d.yaml (src/train/train.py:161-246, trainer.py:97-123) -- sample the batch's latent rows, ModelWithLoss.forward with
chamf + smooth + texture losses (5000 / 1000 surface samples, losses.py:27,61), backward, optim_network.step().  Nothing is rendered
in that configuration, so views := 1.  Per GPU: 16 feet x 6890-vertex template, 10 002-vertex GT scans.  Weak
scaling: every rank owns 16 distinct feet; the gradients of the replicated parameters are averaged with ONE RCCL all-reduce per
step (find_amd/distributed.py).  Rank 0 prints ONE JSON line.

Objects on that line:
  roofline     -- the step's dominant kernel (256x256 Linear+ReLU fp32-MFMA GEMM over all 110 240 head rows) timed live with HIP
                  events on the launch stream; achieved = 2*rows*256*256 flop / average duration.
  cpu_baseline -- the oracle's composition of the same step (oracle/mlp_ref.py + oracle/geom_ref.py) on a bounded sample, rank 0, N=1.
  records      -- (N=1 only) the same measurement for the other named configurations, so the driver's one run sees them all:
                  train3d_b1 (the reference's literal batch size 1, opts.py:40, label-addressed latents, model.py:137-149),
                  train3d_b1_graph (the same step replayed as one HIP graph), train3d_b1_reg_stage (stage 1: chamf only, SGD on reg),
                  train3d_b1_latent_stage[_graph] (stage 3: Trainer.val_epoch's step, Adam(latent_params)) and ..._frozen[_graph] (the same
                  with the network's weights frozen: latents-only backward),
                  c2 (BASELINE configs[1]), c3 (configs[2]), c4_rank_share (one rank's 16 feet x 4 views @512^2 of configs[3]),
                  c5_fp32 / c5_fp16 (configs[4]).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEET = 16
N_VERTS = 6890
N_GT_VERTS = 10002
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16 / 16x16x32: 512 FMA per cycle and SIMD at 2.4 GHz)
SUSTAINED_FP32_MFMA_TFLOPS = 146.9  # measured: tools/mfma_peak.hip, registers only, one wave per SIMD (profiles/r03_mfma_peak.txt)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable by a streaming kernel)
METRIC = 'deformed vertices x rendered views / sec (fwd+bwd)'
UNIT = 'vertices*views/s'
# MAC counts per vertex evaluation: reference-equivalent fwd+bwd, and what this build executes when the
# trunk is shared by the 16 feet of a batch (trunk fwd+bwd once per template vertex instead of once per foot-vertex).
MAC_FWDBWD_REF = 2465536
MAC_TRUNK_FWD = 515 * 256 + 4 * 256 * 256

# HBM-side traffic of one launch of the dominant kernel, measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes (tools/profile_round3.sh pmc; profiles/r03_gemm4_pmc_summary.txt): 2 x 65.0 MB fetched + 112.9 MB written (fp32 gemm4; 241.8 - 243.9 MB over the passes of rounds 2 and 3);
# fp16-mode gemm5 (profiles/r01_traffic_pmc_summary.txt): 2 x 57.10 MB + 112.9 MB.
GEMM_TRAFFIC_BYTES = 242.9e6
GEMM5_TRAFFIC_BYTES = 227.1e6


def pmc_traffic(stem, kernel):
	"""(HBM-side bytes per launch of `kernel`, file name) from the NEWEST committed profiles/rNN_<stem>.txt: the FETCH_SIZE / WRITE_SIZE averages
	(KiB per launch, separate rocprofv3 --pmc passes) of the kernel's section, bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- the gfx950 fetch
	correction of MI355X_MICROARCH.md.  (None, None) when no such file or section exists: the figure on the line is then null, never a constant
	from an older round (the kernel's store path had changed under a hard-coded number)."""
	import glob
	import re
	for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{stem}.txt')), reverse=True):
		vals, section = {}, None
		for ln in open(path):
			m = re.match(r'==\s*(\S+?):?\s*$', ln.strip()) or re.match(r'==\s*(\S+)\s', ln.strip())
			if ln.startswith('=='):
				section = m.group(1).rstrip(':') if m else None
				continue
			m = re.match(r'\s*(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+avg=\s*([0-9.]+)', ln)
			if m and section is not None and kernel in section and m.group(1) not in vals:
				vals[m.group(1)] = float(m.group(2))
		if len(vals) == 2:
			return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, os.path.basename(path)
	return None, None


OWN_STREAM = os.environ.get('FIND_BENCH_STREAM', '0') != '0'


def dtype_label():
	"""Arithmetic type of the path as configured, as a token ('f32' also when the large layers' fp32 products are formed as bf16x3: DTYPE_NOTE)."""
	from find_amd import functional as FF
	return 'f16' if FF.get_mlp_precision() == 'fp16' else 'f32'


def dtype_note():
	from find_amd import functional as FF
	return {'fp16': 'opt-in mode: f16 MFMA operands, f32 accumulation, f32 tensors; not the parity path',
			'bf16x3': 'tensors, sums, results fp32; 256->256 layer products as bf16x3 (exact 3-way bf16 split, 6 products, fp32 accumulate: fp32-faithful, tests/test_gpu_mlp_bf16x3.py)',
			'fp32': 'fp32 MFMA kernels'}[FF.get_mlp_precision()]


LINE_LIMIT = 4096   # the driver's parser lost round 4's 20.5-kB line: the stdout line stays below this, everything else goes to RECORDS_FILE + stderr
RECORDS_FILE = os.path.join(ROOT, 'bench_records.json')


def short(s, n=200):
	return s if len(s) <= n else s[:n - 3] + '...'


def emit_record(name, rec):
	"""One record = one short JSON line on stderr (long strings cut: the full text is in RECORDS_FILE)."""
	def cut(v):
		if isinstance(v, str):
			return short(v, 160)
		if isinstance(v, dict):
			return {k: cut(x) for k, x in v.items() if k not in ('note', 'traffic_note', 'what')}
		if isinstance(v, list):
			return [cut(x) for x in v]
		if isinstance(v, float):
			return float(f'{v:.6g}')
		return v
	out = json.dumps({'record': name, **cut(rec)})
	if len(out) > 1024:   # drop the nested objects before the timing fields
		slim = {k: v for k, v in cut(rec).items() if not isinstance(v, (dict, list))}
		out = json.dumps({'record': name, **slim})
	print(out, file=sys.stderr, flush=True)


def emit(obj):
	"""Print the result as the LAST line of stdout: RCCL writes its version banner through C stdio, which is block-buffered on a pipe
	and would otherwise come out after this line when the process group is destroyed."""
	import ctypes
	try:
		ctypes.CDLL(None).fflush(None)
	except Exception:
		pass
	print(json.dumps(obj), flush=True)


def host_threads():
	try:
		return len(os.sched_getaffinity(0))
	except AttributeError:
		return os.cpu_count() or 1


def set_cpu_threads(n):
	"""torch's intra-op pool and the OpenMP pool of oracle/raster_ref.c."""
	import ctypes
	torch.set_num_threads(n)
	try:
		ctypes.CDLL('libgomp.so.1').omp_set_num_threads(n)
	except OSError:
		pass


def best_of_cpu(one, repeats=5, counts=(16, 32, 64, 128, None), budget_s=16.0):
	"""Time `one()` (returns seconds) on the host the way SURVEY 8d asks: every thread count of a probe {16, 32, 64, 128, all} once after a
	warm-up, then the best count `repeats` times; returns (best seconds, cores used, description).  torch-CPU does not scale to every
	hardware thread of a big host on GEMMs this small, and a fixed count would flatter the GPU/CPU ratio.  Bounded: the probe stops
	widening once `budget_s` seconds have gone."""
	avail = host_threads()
	cand = sorted({min(avail, c if c is not None else avail) for c in counts})
	before = torch.get_num_threads()
	probe, t_start = {}, time.perf_counter()
	for c in cand:
		set_cpu_threads(c)
		one()
		probe[c] = one()
		# (wider is not tried once it has become clearly slower -- 256 threads took 20 s per step where 16 took 0.6 -- or the budget is half gone)
		if probe[c] > 1.5 * min(probe.values()) or time.perf_counter() - t_start > budget_s / 2:
			break
	cores = min(probe, key=probe.get)
	set_cpu_threads(cores)
	times = [probe[cores]]
	while len(times) < repeats and time.perf_counter() - t_start < budget_s:
		times.append(one())
	set_cpu_threads(before)   # the host-bound batch-1 records that follow are timed with the process as it was
	return min(times), cores, (f'best of {len(times)} with {cores} threads (probe over {sorted(probe)} of {cand} threads, stopped where wider got slower: '
							   + ', '.join(f'{c}: {probe[c] * 1e3:.0f} ms' for c in sorted(probe)) + f'; host exposes {avail} hardware threads)')


# ------------------------------------------------------------------------------------------------ self-launch
def spawn_ranks(argv, n):
	"""`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU, the environment
	torch.distributed.run would give them) and return the worst exit code.  The parent never touches the GPU
	(torch.cuda.device_count() does not initialise HIP) and never exec()s."""
	share = os.environ.get('FIND_BENCH_SHARE_GPU') == '1'   # diagnostic (tests/test_gpu_bench.py): every rank on device 0, collectives over gloo
	have = torch.cuda.device_count()
	if have < n and not share:
		raise SystemExit(f'bench.py: --gpus {n} but this node exposes {have} GPU(s)')
	with socket.socket() as s:
		s.bind(('127.0.0.1', 0))
		port = s.getsockname()[1]
	procs = []
	for r in range(n):
		env = dict(os.environ, RANK=str(r), LOCAL_RANK='0' if share else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
				   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
		if share:
			env['FIND_DIST_BACKEND'] = 'gloo'
		# ranks > 0 print nothing to stdout that matters; route it to stderr so that rank 0's JSON line stays the last stdout line
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=None if r == 0 else sys.stderr))
	rc = 0
	try:
		while procs:
			for p in list(procs):
				r = p.poll()
				if r is None:
					continue
				procs.remove(p)
				if r != 0:
					rc = rc or r
					for q in procs:  # a rank died: the others would wait for it in a collective forever
						q.terminate()
			time.sleep(0.05)
	finally:
		for p in procs:
			p.kill()
	return rc


class Run:
	"""Rank bookkeeping + the timing contract: W warm-up steps, then K steps bracketed by barrier + synchronize, MAX over ranks."""

	def __init__(self, gpus):
		from find_amd import distributed as fdist
		if not torch.cuda.is_available():
			raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
		self.rank, self.world, self.local = fdist.init_from_env()
		if self.world != gpus:
			raise SystemExit(f'bench.py: --gpus {gpus} but WORLD_SIZE={self.world}')
		torch.cuda.set_device(self.local)
		self.dev = torch.device('cuda', self.local)

	def barrier(self):
		if self.world > 1:
			torch.distributed.barrier()

	PRIME_S = 0.6   # see timed()

	def timed(self, step, steps, warmup, prime=True):
		"""_timed() with the backward passes on this thread, as find_amd.trainer.Trainer runs its loops (train_utils.backward_on_this_thread:
		autograd's worker thread costs the host 0.5 ms per step and made the eager loop host-bound; FIND_AUTOGRAD_THREADS=1 for torch's default)."""
		from find_amd.train_utils import backward_on_this_thread
		import contextlib
		own = None
		if OWN_STREAM:
			# the loop on a stream of the bench's own, as find_amd.trainer.Trainer runs its loops (not HIP's legacy default stream, with which
			# every blocking stream of the process -- the context's CU-masked side streams are of that kind -- synchronises implicitly)
			own = self.__dict__.setdefault('_own_stream', torch.cuda.Stream(device=self.dev))
			torch.cuda.synchronize()
		with backward_on_this_thread(), (torch.cuda.stream(own) if own is not None else contextlib.nullcontext()):
			return self._timed(step, steps, warmup, prime)

	def _timed(self, step, steps, warmup, prime=True):
		"""W untimed warm-up steps, then exactly K timed steps between barrier + synchronize on both sides (the driver's contract).  Before
		the warm-up the workload is PRIMED -- untimed steps for at least PRIME_S seconds, until the step time has settled -- so that first-use
		costs of a fresh process (allocator growth, queue probing, module loads) are behind it whatever W is.  What is timed is unchanged:
		K steps of the steady state.  `--no-prime` switches it off.  (what looked like a cold-start effect when this was added was
		a reference cycle in the product, found by the repeats and fixed.)"""
		if prime and Run.PRIME_S > 0:
			# batches of eight steps until at least PRIME_S seconds have gone by AND two consecutive batches agree within 5 % (at most
			# 8 x PRIME_S: the first process on a fresh box also pays first-use costs -- allocator growth, queue probing, module loads --
			# for a few dozen steps).  A step of the data-parallel run holds a collective: the ranks agree on every "go on".
			prev, total = None, 0.0
			while True:
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				for _ in range(8):
					step()
				torch.cuda.synchronize()
				dt = time.perf_counter() - t0
				total += dt
				done = (total >= Run.PRIME_S and prev is not None and abs(dt - prev) <= 0.05 * prev) or total >= 8 * Run.PRIME_S
				if self.world > 1:
					t = torch.tensor([int(done)], device=self.dev, dtype=torch.int64)
					torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
					done = bool(t.item())
				if done:
					break
				prev = dt
		for _ in range(warmup):
			step()
		self.barrier()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(steps):
			step()
		torch.cuda.synchronize()
		self.barrier()
		torch.cuda.synchronize()
		elapsed = time.perf_counter() - t0
		if self.world > 1:
			t = torch.tensor([elapsed], device=self.dev, dtype=torch.float64)
			torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
			elapsed = float(t.item())
		return elapsed / steps * 1e3

	def finish(self):
		if self.world > 1:
			import ctypes
			ctypes.CDLL(None).fflush(None)
			sys.stdout.flush()
			torch.distributed.barrier()
			torch.distributed.destroy_process_group()


def note(msg):
	"""Progress to stderr (which record is running: a fault in one of them is then attributable)."""
	print(f'[bench] {msg}', file=sys.stderr, flush=True)


def train_backward(loss):
	"""loss.backward() as find_amd.trainer.Trainer issues it (train_utils.backward: the seed gradient from a cache instead of a fill launch)."""
	from find_amd.train_utils import backward
	backward(loss)


def line(value, ms, run, steps, warmup, config, **extra):
	config = dict(config, workload=short(config['workload'], 240))
	out = {'metric': METRIC, 'value': value, 'unit': UNIT, 'n_gpus': run.world, 'steps': steps, 'warmup': warmup, 'ms_per_step': ms,
		   'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_label(), 'data': 'synthetic', 'config': config,
		   'prime_s': Run.PRIME_S}   # untimed steady-state priming before the W warm-up steps (Run.timed)
	out.update(extra)
	return out


# ------------------------------------------------------------------------------------------------ train_3d.yaml
def make_mwl(dev, n_items, opts, n_verts=None, labels=None, size=None, val_size=2):
	from find_amd import synthetic
	from find_amd.model_with_loss import ModelWithLoss
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=n_items, val_size=val_size,
						shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None, latent_labels=labels)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():  # the reference zero-initialises this layer (model.py:516-518): most of the head's backward would be exact zeros
		mwl.model.mlp_disp[-1].weight.copy_(torch.randn(mwl.model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		mwl.model.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
	mwl = mwl.to(dev)
	if size is not None:
		from find_amd.renderer import FootRenderer
		mwl.rdr = FootRenderer(image_size=size, device=dev)
	v, f = synthetic.template(n_verts or N_VERTS)
	mwl.model.set_template(v.to(dev), f.to(dev))
	return mwl


def fill_latents(m, n_feet, seed, dev):
	"""Seeded latent rows; tables addressed by label can be shorter than the item count (shared shape / tex rows)."""
	from find_amd import synthetic
	lat = synthetic.latents(n_feet, seed=seed, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			t = getattr(m, k).data
			t.copy_(lat[k][:t.shape[0]])


def train3d_setup(run, n_items, batch_size, stage='net', labels=False, seed=0, dp=True, capturable=False, frozen=False):
	"""One step of the train_3d.yaml experiment as src/train/trainer.py runs it.  stage 'net' (train.py:211-215, trainer.py:97-123): losses
	chamf + smooth + texture, optim_network = Adam(main_params); stage 'reg' (train.py:197-209): chamf only, optim_reg = SGD(reg_params,
	momentum 0.9); stage 'latent' (train.py:217-224 -> Trainer.val_epoch, trainer.py:150-163): the same three losses with is_train=False on
	the validation scans, their `*_val` rows, optim_latent = Adam(latent_params) -- `frozen` additionally takes requires_grad off the
	network's weights (nothing steps them in this stage; the reference leaves them trainable and computes their gradients for nothing).
	batch_size < n_items walks the items round-robin like the DataLoader (batch_size_train / _val default to 1, opts.py:40-41)."""
	from find_amd import distributed as fdist
	from find_amd import optim, synthetic
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	dev = run.dev
	net = stage != 'reg'
	val = stage == 'latent'
	opts = Opts(chamf_loss=True, smooth_loss=net, texture_loss=net, use_pose_code=True, use_latent_labels=labels)
	feet, names, lab = synthetic.scan_labels(n_items, n_val=n_items)
	if val:
		feet, names = [f'{9000 + i // 2:04d}' for i in range(n_items)], lab['pose_val']
	mwl = make_mwl(dev, n_items, opts, labels=lab if labels else None, val_size=n_items if val else 2)
	m = mwl.model
	fill_latents(m, n_items, seed, dev)
	if val:
		lat = synthetic.latents(n_items, seed=seed + 100, device=dev)
		with torch.no_grad():
			for k in ('shapevec', 'texvec', 'posevec', 'reg'):
				t = getattr(m, k + '_val').data
				t.copy_(lat[k][:t.shape[0]])
	gv, gf, gc = synthetic.gt_feet(n_items, N_GT_VERTS, seed=seed, device=dev)
	gc = gc.clamp(0.05, 0.95)
	batches = []
	for lo in range(0, n_items, batch_size):
		hi = lo + batch_size
		b = dict(mesh=Meshes(gv[lo:hi].contiguous(), gf, TexturesVertex(gc[lo:hi].contiguous())), idx=torch.arange(lo, hi, device=dev), name=names[lo:hi])
		if labels:  # what default_collate makes of the dataset items' label strings: lists of str
			b.update(shape=feet[lo:hi], tex=feet[lo:hi], pose=names[lo:hi], reg=names[lo:hi])
		batches.append(b)
	# learning rates: the reference's defaults (src/train/opts.py:54-57: lr_net 5e-5, lr_reg 1e-5, lr_latent 1e-4)
	if stage == 'net':
		opt = optim.Adam(m.main_params, lr=5e-5, capturable=capturable)
	elif stage == 'reg':
		opt = optim.SGD(m.reg_params, lr=1e-5, momentum=0.9)
	else:
		opt = optim.Adam(m.latent_params, lr=1e-4, capturable=capturable)
	if frozen:
		for seq in (m.base, m.mlp_disp, m.mlp_col):
			for p in seq.parameters():
				p.requires_grad_(False)
	bucket = None
	if run.world > 1 and dp:
		fdist.broadcast_parameters([p for p in m.parameters() if p.is_floating_point()])
		# the MLP's weight gradients are complete when the main pass's MLP backward returns: their part of the bucket goes out there, under
		# the loss-side tail of the backward (registration, latent scatter); the latent tables' part follows behind the backward
		mlp_w = [p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()]
		bucket = fdist.GradBucket([p for p in m.parameters() if p.requires_grad], early=mlp_w)
		bucket.arm_early(m.base[0].weight)
	flags = dict(chamf=True, smooth=net, texture=net)
	if stage == 'reg':
		flags = dict(chamf=True, smooth=False, gt_z_cutoff=opts.gt_z_cutoff)   # train.py:201, verbatim (gt_z_cutoff: None by default, opts.py:145)
	if val:
		flags['is_train'] = False
	vectors = m.latent_vectors_val if val else m.latent_vectors_train
	state = dict(i=0)

	def step():
		opt.zero_grad(set_to_none=True)
		b = dict(batches[state['i'] % len(batches)])
		state['i'] += 1
		b.update(sample_latent_vectors(b, vectors))
		loss, _ = mwl(b, 0, opts, **flags)
		train_backward(loss)
		if bucket is not None:
			bucket.allreduce_(async_op=True)   # the latent tables' part (the weights' part left inside the backward: GradBucket.arm_early);
			bucket.wait()                      # the optimiser needs both: the wait follows at once (stream-side dependency, the host does not block)
		opt.step()
		return loss

	def collective(iters=20):
		"""What the N > 1 line says about the exchange step (every rank calls this): backend, the world size the process group reports, the
		devices the ranks sit on, bucket bytes, and the collective's own time on rank 0 -- HIP events around allreduce_ + wait on the
		gradients the last step left behind (nothing else in flight: the ring's latency, not its overlap)."""
		import torch.distributed as dist
		if bucket is None:
			return None
		props = torch.cuda.get_device_properties(dev)
		mine = dict(rank=run.rank, device=f'cuda:{dev.index}', name=props.name, uuid=str(getattr(props, 'uuid', '')), pci_bus=getattr(props, 'pci_bus_id', None))
		ranks = [None] * run.world
		dist.all_gather_object(ranks, mine)
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		dist.barrier()
		torch.cuda.synchronize()
		e0.record()
		for _ in range(iters):
			bucket.allreduce_(async_op=True)
			bucket.wait()
		e1.record()
		e1.synchronize()
		return dict(backend=dist.get_backend(), world_size=dist.get_world_size(), bucket_bytes=bucket.numel * 4, early_prefix_bytes=bucket.n_early * 4,
					steps_with_early_prefix=bucket.early_issued, op='AVG in place' if dist.get_backend() == 'nccl' else 'SUM + divide',
					allreduce_us_per_step=e0.elapsed_time(e1) / iters * 1e3, ranks_devices=[f"{r['rank']}:{r['device']}" + (f"@{r['pci_bus']}" if r['pci_bus'] is not None else '') for r in ranks],
					distinct_devices=len({(r['device'], r['pci_bus'], r['uuid']) for r in ranks}))

	return dict(mwl=mwl, step=step, gt=(gv, gf, gc), opts=opts, opt=opt, batches=batches, flags=flags, collective=collective)


def train3d_executed_flops(n_feet, n_verts=None, n_tex=1000, stage='net', frozen=False, lazy_col=None):
	"""Flops this build executes for one train_3d.yaml step (2 x multiply-accumulates of every Linear layer, forward and backward; the
	loss kernels are not GEMM work and are left out).  Main pass: template rows shared by the feet of a batch (trunk and the heads' first
	layers once per TEMPLATE vertex, the later head layers per foot-vertex); its colour head runs forward only (nothing of a 3-D-loss step
	reads the colours of the predicted mesh, so autograd never enters it -- as in the reference) or, with lazy_col (the product's default
	since round 3: find_amd.model_with_loss.LAZY_COLOURS), not at all.  Texture pass (net / latent stages): n_tex samples per foot at
	per-foot positions, colour head only.  frozen: the latents-only backward (no trunk, no weight gradients)."""
	if lazy_col is None:
		import find_amd.model_with_loss as MWL
		lazy_col = MWL.LAZY_COLOURS
	V = n_verts or N_VERTS
	L = 256 * 256
	first, out = 515 * 256, 3 * 256
	nV = n_feet * V
	mac = V * (first + 4 * L) + V * 2 * L + nV * (4 * L + 2 * out)                       # main forward
	if lazy_col:
		mac -= V * L + nV * (2 * L + out)                                                 # ... without the template pass's colour head
	if frozen:
		mac += nV * (out + 2 * L)                                                         # disp head: output layer dX, two hidden dX
	else:
		mac += nV * 2 * out + nV * 4 * L + V * L + V * L + V * 8 * L + V * first          # disp head dX + dW, first-layer dW, trunk-out dX, trunk dX + dW, Fourier dW
	if stage != 'reg':
		r = n_feet * n_tex
		mac += r * (first + 4 * L + L + 2 * L + out)                                      # texture forward: trunk, colour head
		if frozen:
			mac += r * (out + 2 * L)
		else:
			mac += r * (2 * out + 4 * L + L + L + 8 * L + first)
	return 2.0 * mac


def train3d_workload(n_feet, stage, labels, frozen=False):
	what = {'net': 'chamf(5000)+smooth+texture(1000) losses, backward, Adam(main_params) step',
			'reg': 'chamf(5000) only, backward, SGD(reg_params, momentum 0.9) step',
			'latent': 'Trainer.val_epoch: is_train=False, chamf+smooth+texture on a val scan, backward, Adam(latent_params) step, weights '
					  + ('frozen' if frozen else 'trainable as in the reference')}[stage]
	name = {'net': 'network', 'reg': 'registration', 'latent': 'latent'}[stage]
	return (f'train_3d.yaml {name}-stage step: batch {n_feet} x {N_VERTS}-vertex template, {N_GT_VERTS}-vertex GT scans, {what}; '
			f'latents by {"label" if labels else "index"}; views:=1')


def train3d_cpu(mwl, gt, stage='net', sample_feet=1):
	"""Oracle composition of the same step (tests/test_gpu_pipeline.py::test_train_3d_loss_set_matches_oracle) on `sample_feet` feet,
	without the optimiser update."""
	from oracle import compose_ref, geom_ref, mlp_ref
	gv, gf, gc = gt
	m = mwl.model
	nf = sample_feet
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	lat = {k: getattr(m, k).data.detach().cpu()[:nf].clone().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec', 'reg')}
	gvc, gfc, gcc = gv.cpu()[:nf], gf.cpu(), gc.cpu()[:nf]
	g = torch.Generator().manual_seed(0)

	def draws(verts, faces, n):
		areas = geom_ref.face_areas(verts, faces)
		return torch.multinomial(areas, n, replacement=True, generator=g), torch.rand(verts.shape[0], n, 2, generator=g)

	def one():
		# (oracle.compose_ref.train3d_losses: the reference's ModelWithLoss.forward composition, pinned to the reference's own by
		# tests/test_oracle_pins.py; the draws as PyTorch3D's sampler makes them: multinomial over face areas + uniform (u, v))
		t0 = time.perf_counter()
		dr = dict(gt=draws(gvc, gfc, 5000), pred=lambda v: draws(v, tf, 5000))
		if stage == 'net':
			dr['tex'] = draws(gvc, gfc, 1000)
		total, _ = compose_ref.train3d_losses(sd, B, tv, tf, lat, gvc, gfc, gcc, dr, chamf=True, smooth=stage == 'net', texture=stage == 'net')
		total.backward()
		return time.perf_counter() - t0

	best, cores, how = best_of_cpu(one)
	return dict(value=nf * N_VERTS / best, unit=UNIT, cores=cores, kind='port',
				sample=f'{nf} foot of the same step (batch {nf}) without the optimiser update, oracle (torch-CPU / numpy), {how}')


def time_dominant_kernel(device, iters=100, warm=150, n_verts=None):
	"""Average duration of the dominant kernel (Linear 256->256 + ReLU over all head rows), HIP events on the launch stream."""
	import ctypes
	from find_amd import _lib
	L = _lib.lib()
	n_verts = n_verts or N_VERTS
	rows = N_FEET * n_verts
	g = torch.Generator().manual_seed(0)
	x = torch.randn(rows, 256, generator=g).to(device)
	w = (torch.randn(256, 256, generator=g) / 16).to(device)
	b = torch.randn(256, generator=g).to(device)
	y = torch.empty_like(x)
	stream = torch.cuda.current_stream(device)

	def launch():
		_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), N_FEET, n_verts, _lib.ptr(y),
										  ctypes.c_void_p(stream.cuda_stream)), 'find_linear_relu_fwd')

	# the GPU idled while the inputs were generated on the host: launch long enough for the clock to come back up before timing
	for _ in range(warm):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record(stream)
	for _ in range(iters):
		launch()
	e1.record(stream)
	e1.synchronize()
	ms = e0.elapsed_time(e1) / iters
	flops = 2.0 * rows * 256 * 256
	del x, y
	return ms, flops


def time_wgrad_kernel(device, iters=60, warm=60):
	"""Average duration of one 256 x 256 weight gradient over all head rows (find_linear_wgrad: the dW kernel + its slab reduce)."""
	import ctypes
	from find_amd import _lib
	L = _lib.lib()
	rows = N_FEET * N_VERTS
	g = torch.Generator().manual_seed(1)
	dz = (torch.randn(rows, 256, generator=g) * 0.1).to(device)
	x = torch.relu(torch.randn(rows, 256, generator=g)).to(device)
	dw, db = torch.empty(256, 256, device=device), torch.empty(256, device=device)
	nb = L.find_linear_wgrad_scratch_bytes(N_FEET)
	scratch = torch.empty(nb // 4, device=device)
	stream = torch.cuda.current_stream(device)

	def launch():
		_lib.check(L.find_linear_wgrad(_lib.ctx(), _lib.ptr(dz), _lib.ptr(x), N_FEET, N_VERTS, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch), nb,
									   ctypes.c_void_p(stream.cuda_stream)), 'find_linear_wgrad')
	for _ in range(warm):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record(stream)
	for _ in range(iters):
		launch()
	e1.record(stream)
	e1.synchronize()
	return e0.elapsed_time(e1) / iters


def time_small_pass(device, n_pts, shared, iters=40, warm=20):
	"""Forward and forward + backward time of one SMALL MLP call -- the texture pass (16 feet x 1000 per-foot points, colour head only: the
	64-row fused chain) or the template pass's shared trunk (batch of 16 on the 6890-vertex template is the large path; here batch 1) --
	HIP events around the Python call, so the few small launches around the fused chain are inside.  Returns (fwd ms, fwd+bwd ms)."""
	from find_amd import synthetic
	model = synthetic.make_model(N_VERTS, train_size=N_FEET, val_size=2, device=device)
	fill_latents(model, N_FEET, 0, device)
	g = torch.Generator().manual_seed(5)
	n = 1 if shared else N_FEET
	lat = {k: getattr(model, k).data[:n].detach().clone().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec')}
	pos = model.template_verts.data if shared else (torch.rand(N_FEET, n_pts, 3, generator=g) * 0.2 - 0.1).to(device)
	kw = {} if shared else dict(want=('col',))
	stream = torch.cuda.current_stream(device)

	def fwd():
		with torch.no_grad():
			return model(pos, shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'], **kw)

	def fwdbwd():
		for p in model.parameters():
			p.grad = None
		res = model(pos, shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'], **kw)
		(res['col'] ** 2).sum().backward() if not shared else ((res['col'] ** 2).sum() + (res['disp'] ** 2).sum()).backward()

	out = []
	from find_amd.train_utils import backward_on_this_thread
	with backward_on_this_thread():   # (as the step runs it: with autograd's worker thread this loop measures the host)
		for fn in (fwd, fwdbwd):
			for _ in range(warm):
				fn()
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record(stream)
			for _ in range(iters):
				fn()
			e1.record(stream)
			e1.synchronize()
			out.append(e0.elapsed_time(e1) / iters)
	return out[0], out[1]


# Time per step of the kernels that take the most of it INSIDE the headline step (rocprofv3 --kernel-trace of `bench.py --headline-only`, steps cut at
# the optimiser kernel; several streams run side by side there, so these sum to more than the step): read from the newest committed
# profiles/rNN_headline_step_stats.csv, as roofline.traffic is read from the newest PMC summary -- no figure typed into this file.
def in_step_us(*needles):
	"""us per step of all launches whose kernel name contains one of `needles`, and the file it came from; (None, None) without a file."""
	import csv
	import glob
	files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_headline_step_stats.csv')))
	if not files:
		return None, None
	tot, hit = 0.0, False
	with open(files[-1]) as f:
		for row in csv.DictReader(ln for ln in f if not ln.startswith('#')):
			if any(n in row['kernel'] for n in needles):
				tot += float(row['us_per_step'])
				hit = True
	return (tot if hit else None), os.path.basename(files[-1])


def dominant_roofline(device, fp16=False, n_verts=None):
	from find_amd import functional as FF
	n_verts = n_verts or N_VERTS
	kms, kflops = time_dominant_kernel(device, n_verts=n_verts, iters=100 if n_verts == N_VERTS else 30, warm=150 if n_verts == N_VERTS else 30)
	rows = N_FEET * n_verts
	nbytes = 2.0 * rows * 1024 + 256 * 1024   # algorithmic: rows x 1 KB read + rows x 1 KB written + the 256-KB weight matrix
	gbs = nbytes / (kms * 1e-3) / 1e9
	if fp16:
		# gemm5 is bound by its streams
		return {'bound': 'hbm', 'kernel': f'find::mlp::gemm5_kernel<1> (Linear 256->256 + bias + ReLU over {rows} rows, fp16 MFMA operands, fp32 tensors in HBM)',
				'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS, 'avg_kernel_ms': kms,
				'bytes_per_launch': nbytes, 'traffic': GEMM5_TRAFFIC_BYTES if n_verts == 6890 else (pmc_traffic('gemm5_c5_pmc_summary', 'gemm5_kernel')[0] if n_verts == 50002 else None),
				'traffic_note': 'HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): '
								+ ('profiles/r01_traffic_pmc_summary.txt' if n_verts == 6890 else f"profiles/{pmc_traffic('gemm5_c5_pmc_summary', 'gemm5_kernel')[1]}"),
				'mfma_tflops': kflops / (kms * 1e-3) / 1e12}
	ach = kflops / (kms * 1e-3) / 1e12
	if FF.get_mlp_precision() == 'bf16x3':
		# six bf16 products per fp32 multiply-accumulate: the flops the matrix pipe EXECUTES are 6 x the layer's
		ex = 6.0 * ach
		return {'bound': 'mfma', 'kernel': f'find::mlp::gemm7_kernel<1, 0, false, false> (Linear 256->256 + bias + ReLU over {rows} rows; bf16x3: v_mfma_f32_16x16x32_bf16, fp32 accumulation)',
				'achieved': ex, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ex / PEAK_BF16_MFMA_TFLOPS,
				'avg_kernel_ms': kms, 'flops_per_launch': 6.0 * kflops, 'flops_per_launch_fp32_equivalent': kflops,
				'fp32_equivalent_tflops': ach, 'x_fp32_mfma_peak': ach / PEAK_FP32_MFMA_TFLOPS,
				'hbm_gbs_algorithmic': gbs, 'frac_of_hbm_peak': gbs / PEAK_HBM_GBS, 'traffic': pmc_traffic('gemm7_pmc_summary', 'gemm7_kernel')[0] if n_verts == 6890 else None,
				'traffic_note': 'HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), read from '
								f"profiles/{pmc_traffic('gemm7_pmc_summary', 'gemm7_kernel')[1]}; algorithmic 226.0e6",
				'note': 'achieved / peak count the bf16 products the matrix pipe executes (6 per fp32 multiply-accumulate); fp32_equivalent_tflops is the layer\'s own '
						'2*rows*256*256 flop count over the same time -- above the fp32 MFMA peak (x_fp32_mfma_peak), which the round-3 kernel (gemm4, records.fp32_mfma) sat at 0.80 of'}
	return {'bound': 'mfma', 'kernel': f'find::mlp::gemm4_kernel<1, 4, 8> (Linear 256->256 + bias + ReLU over {rows} rows, fp32 MFMA)',
			'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TFLOPS,
			'avg_kernel_ms': kms, 'flops_per_launch': kflops, 'traffic': GEMM_TRAFFIC_BYTES if N_VERTS == 6890 else None,
			'traffic_note': 'HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), '
							'profiles/r03_gemm4_pmc_summary.txt; algorithmic 226.0e6',
			# `peak` is the guide's 2.4 GHz figure; a loop of nothing but back-to-back v_mfma_f32_32x32x2_f32 (tools/mfma_peak.hip, 64.0 cycles per
			# instruction and SIMD) sustains 146.9 TFLOP/s on this part: the clock under fp32 MFMA load settles at 2.24 GHz (profiles/r03_mfma_peak.txt)
			'sustained_mfma_tflops_measured': SUSTAINED_FP32_MFMA_TFLOPS, 'frac_of_sustained': ach / SUSTAINED_FP32_MFMA_TFLOPS}


def roofline_kernels(device, lin_ms):
	"""The kernels with the most time per step, each timed alone with HIP events (`isolated_us`) beside its time per step inside the headline
	step (`in_step_us`, from the committed kernel trace: the step's streams run kernels side by side, a kernel is slower there than alone)."""
	from find_amd import functional as FF
	x3 = FF.get_mlp_precision() == 'bf16x3'
	rows = N_FEET * N_VERTS
	L = 2.0 * rows * 65536
	mul = 6.0 if x3 else 1.0
	peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
	out = []

	def entry(name, what, flops32, us, executed_mul, in_step_key):
		ex = flops32 * executed_mul
		e = {'kernel': name, 'what': what, 'flops_fp32_equivalent': flops32, 'isolated_us': us, 'fp32_equivalent_tflops': flops32 / (us * 1e-6) / 1e12,
			 'executed_tflops': ex / (us * 1e-6) / 1e12, 'pipe': 'bf16 MFMA (bf16x3: 6 products per multiply-accumulate)' if executed_mul == 6.0 else 'fp32 MFMA',
			 'frac': ex / (us * 1e-6) / 1e12 / (PEAK_BF16_MFMA_TFLOPS if executed_mul == 6.0 else PEAK_FP32_MFMA_TFLOPS),
			 'in_step_us_per_step': in_step_us(*in_step_key)[0], 'in_step_source': in_step_us(*in_step_key)[1]}
		out.append(e)

	entry('gemm7_kernel<1>' if x3 else 'gemm4_kernel<1,4,8>', f'Linear 256->256 + bias + ReLU over {rows} rows (forward; 2 launches per step)', L, lin_ms * 1e3, mul, ('gemm7_kernel<1,',) if x3 else ('gemm4_kernel<1',))
	wg_ms = time_wgrad_kernel(device)
	entry(('dw6_kernel' if x3 else 'dw4_kernel') + ' + reduce_w_kernel', f'dW = dZ^T X over {rows} rows + its slab reduce (2 launches per step)', L, wg_ms * 1e3, mul, ('dw6_kernel', 'dw6v_kernel') if x3 else ('dw4_kernel',))
	# the small calls run whole layer chains per launch (bf16x3: fused6_kernel on the bf16 pipe; fp32: fused_chain_kernel): time the call (the chain
	# + the handful of small launches around it: repack, latent bias, weight split, head output)
	r = N_FEET * 1000
	Lm = 65536.0
	tex_fwd = 2.0 * r * (515 * 256 + 4 * Lm + Lm + 2 * Lm + 3 * 256)
	tex_bwd = 2.0 * r * (2 * 3 * 256 + 4 * Lm + Lm + Lm + 8 * Lm + 515 * 256)
	f_ms, fb_ms = time_small_pass(device, 1000, shared=False)
	chain = 'fused6_kernel<2>' if x3 else 'fused_chain_kernel<2>'
	entry(chain + ' (forward call)', f'texture pass forward: {N_FEET} x 1000 per-foot points, Fourier layer + trunk + colour head in one launch (+ repack, latent bias, '
		  'weight split, 3-wide output around it)', tex_fwd, f_ms * 1e3, mul, (chain,))
	# the backward call: dX chain in one launch, the seven weight gradients as one grouped launch (bf16x3: dw6_group) and the Fourier layer's beside
	# them (dwpe6); since round 6 all of it on the bf16 pipe under bf16x3 -- priced as executed products against that pipe's peak
	grp = ('dw6_group_kernel', 'dwpe6_kernel') if x3 else ('dw4_group_kernel', 'dwpe_kernel')
	entry(chain + ' + ' + ' + '.join(grp) + ' (backward call)', 'texture pass backward: dX chain in one launch, seven weight gradients as one grouped launch and the '
		  "Fourier layer's beside them (in_step: the grouped + Fourier weight-gradient kernels of BOTH passes of a step)", tex_bwd, (fb_ms - f_ms) * 1e3, mul, grp)
	return out


# ------------------------------------------------------------------------------------------------ C2 / C5 (BASELINE configs[1], [4])
def executed_flops_per_step(n_feet, n_verts):
	"""fwd+bwd flops this build executes for one batch with a shared template (per-foot latent columns folded into a bias)."""
	trunk_fwd = MAC_TRUNK_FWD
	trunk_bwd = 2 * MAC_TRUNK_FWD - 515 * 256  # dW + dX, no dX through layer 0
	heads_in = 2 * 256 * 256                      # first layer of both heads: the latent columns are a bias, and the trunk rows are shared,
	heads_rest = 4 * 256 * 256 + 2 * 3 * 256      # so H W^T (forward), dW and dH (backward, after the sum over feet) run once per TEMPLATE vertex
	return 2.0 * (n_verts * (trunk_fwd + trunk_bwd + 3 * heads_in) + n_feet * n_verts * 3 * heads_rest)


def build_step(device, seed, n_verts=None):
	from find_amd import synthetic
	n_verts = n_verts or N_VERTS
	model = synthetic.make_model(n_verts, train_size=N_FEET, val_size=2, device=device)
	fill_latents(model, N_FEET, seed, device)
	idx = torch.arange(N_FEET, device=device)
	params = [p for p in model.parameters() if p.requires_grad]

	def step():
		for p in params:
			p.grad = None
		batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx],
					 reg_train=model.reg[idx])
		res = model.get_meshes_from_batch(batch, is_train=True)
		loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
		train_backward(loss)
		return loss

	return model, params, step


def c2_cpu(sample_feet=2):
	"""oracle/mlp_ref.py = the reference's op sequence (no trunk sharing, latents concatenated per vertex) on host cores."""
	from find_amd import synthetic
	from oracle import mlp_ref
	model = synthetic.make_model(N_VERTS, train_size=N_FEET, val_size=2, device='cpu')
	sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in model.state_dict().items()}
	B = model.encoder[0]._B
	tv = model.template_verts.data

	def one_step(n_feet):
		lat = synthetic.latents(n_feet, seed=0, device='cpu')
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		t0 = time.perf_counter()
		res = mlp_ref.get_meshes_verts(sd, B, tv, lv['shapevec'], lv['reg'], lv['texvec'], lv['posevec'])
		loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
		train_backward(loss)
		return time.perf_counter() - t0

	best, cores, how = best_of_cpu(lambda: one_step(sample_feet))
	return dict(value=sample_feet * N_VERTS / best, unit=UNIT, cores=cores, kind='port',
				sample=f'{sample_feet} of {N_FEET} feet x {N_VERTS} verts, fwd+bwd, torch-CPU, {how}')


def c2_record(run, steps, warmup, n_verts=None, fp16=False, with_cpu=False, dp_overhead=False):
	"""BASELINE configs[1] (n_verts 6890) / configs[4] (50 002, with the opt-in fp16 matrix pipe when fp16): 16 feet x template through
	Fourier PE + trunk + heads + registration, loss = sum(verts^2) + sum(col^2), full backward."""
	from find_amd import distributed as fdist
	from find_amd import functional as FF
	n_verts = n_verts or N_VERTS
	prev = FF.set_mlp_precision('fp16') if fp16 else None
	try:
		model, params, step = build_step(run.dev, seed=run.rank, n_verts=n_verts)
		bucket = None
		if dp_overhead and run.world == 1:
			with socket.socket() as _s:   # (a free port, not a fixed one: TIME_WAIT of a run a minute ago)
				_s.bind(('127.0.0.1', 0))
				_port = _s.getsockname()[1]
			torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_port}', rank=0, world_size=1)
			bucket = fdist.GradBucket(params)
		if run.world > 1:
			fdist.broadcast_parameters([p for p in model.parameters() if p.is_floating_point()])
			bucket = fdist.GradBucket(params)

		def full_step():
			step()
			if bucket is not None:
				bucket.allreduce_()

		ms = run.timed(full_step, steps, warmup)
		fl_exec = executed_flops_per_step(N_FEET, n_verts)
		fl_ref = 2.0 * MAC_FWDBWD_REF * N_FEET * n_verts
		name = (('C5 geometry, ' + ('fp16 MLP' if fp16 else 'fp32')) if n_verts == 50002 else ('C2, fp16 MLP (opt-in mode)' if fp16 else 'C2'))
		cfg = {'workload': f'{name}: {N_FEET} feet x {n_verts}-vertex template per GPU, PE+trunk+heads+registration fwd+bwd, views:=1',
			   'feet_per_gpu': N_FEET, 'template_verts': n_verts,
			   'parallelism': f'dp{run.world}' + (' through the one-rank bucket + RCCL path (diagnostic)' if dp_overhead and run.world == 1 else ''),
			   'flops_executed_per_step': fl_exec, 'flops_reference_equiv_per_step': fl_ref,
			   'step_tflops_executed': fl_exec / (ms * 1e-3) / 1e12, 'step_frac_of_fp32_mfma_peak_executed': fl_exec / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
			   'step_tflops_reference_equiv': fl_ref / (ms * 1e-3) / 1e12}
		out = line(run.world * N_FEET * n_verts / (ms * 1e-3), ms, run, steps, warmup, cfg)
		if bucket is not None:
			bucket.close()
		if with_cpu and run.world == 1 and n_verts == N_VERTS:
			out['cpu_baseline'] = c2_cpu()
		if fp16 and n_verts == 50002 and run.rank == 0:
			# BASELINE configs[4] is "fp16 MLP with MFMA tiles, roofline run": the mode's Linear kernel is bound by its streams (HBM)
			del model, params, step
			torch.cuda.empty_cache()
			out['roofline'] = dominant_roofline(run.dev, fp16=True, n_verts=n_verts)
		return out
	finally:
		if prev is not None:
			FF.set_mlp_precision(prev)


# ------------------------------------------------------------------------------------------------ C3 / C4 (BASELINE configs[2], [3])
def c3_record(run, steps, warmup, with_cpu, n_feet=16, n_views=4, size=256, c4=False, mesh='uniform'):
	"""BASELINE.json configs[2] (and, with c4=True, the per-rank share of configs[3]: 16 of the 128 feet, 4 views @512^2, silhouette + pixel +
	Chamfer losses).  configs[2]: a batch of 16 feet x 4 views @256^2 with the silhouette render loss, end to end -- MLP query,
	registration, GT and predicted renders (the GT is re-rendered every step, as the reference does), silhouette loss, backward
	through rasteriser and MLP, optimiser step.
	mesh: the triangulation of template and GT scans (find_amd.synthetic.MESH_KIND).  'uniform' (the record `c3` / `c4_rank_share`): Fibonacci-
	sphere hulls, valence 5 - 7 everywhere -- what FIND's template and decimated scans look like to a rasteriser.  'latlong' (the records
	`*_latlong_stress`): latitude-longitude grids whose two poles put thousands of sliver faces into single tiles -- the binning's worst case,
	and what rounds 1 - 5 quoted as the configuration's number."""
	import numpy as np
	from find_amd import distributed as fdist
	from find_amd import optim, synthetic
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	dev = run.dev
	if c4:
		size = 512
	opts = Opts(sil_loss=True, pix_loss=c4, chamf_loss=c4, num_views=n_views)
	prev_kind, synthetic.MESH_KIND = synthetic.MESH_KIND, mesh
	try:
		mwl = make_mwl(dev, n_feet, opts, size=size)
		m = mwl.model
		fill_latents(m, n_feet, run.rank, dev)
		gv, gf, gc = synthetic.gt_feet(n_feet, N_GT_VERTS, seed=run.rank, device=dev)
	finally:
		synthetic.MESH_KIND = prev_kind
	batch = dict(mesh=Meshes(gv, gf, TexturesVertex(gc.clamp(0.05, 0.95))), idx=torch.arange(n_feet, device=dev), name=[f'{i:04d}' for i in range(n_feet)])
	np.random.seed(7)
	R, T = mwl.rdr.sample_views(nviews=n_views, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	opt = optim.Adam(m.main_params, lr=5e-5)
	bucket = None
	if run.world > 1:
		fdist.broadcast_parameters([p for p in m.parameters() if p.is_floating_point()])
		bucket = fdist.GradBucket([p for p in m.parameters() if p.requires_grad])

	def step():
		opt.zero_grad(set_to_none=True)
		b = dict(batch)
		b.update(sample_latent_vectors(b, m.latent_vectors_train))
		loss, _ = mwl(b, 0, opts, sil=True, pix=c4, chamf=c4, render_foot=True, views=(R, T))
		train_backward(loss)
		if bucket is not None:
			bucket.allreduce_(async_op=True)   # issued behind the backward's last kernel; nothing else of the step is independent of the gradients,
			bucket.wait()                      # so the wait follows at once (the host does not block: stream-side dependency)
		opt.step()
		return loss

	# (the median of three K-step timings, as the headline: one allocator stall inside a 20-step window once made a 4.8-ms step read 12.4)
	def mallocs():
		return torch.cuda.memory_stats(dev).get('num_device_alloc', 0) if dev.type == 'cuda' else 0
	m0 = mallocs()
	ms_all = [run.timed(step, steps, warmup, prime=(i == 0)) for i in range(3)]
	ms = sorted(ms_all)[1]
	n_malloc = mallocs() - m0   # (a timed window that has the caching allocator go to the driver reads milliseconds too long: reported)
	cfg = {'workload': f'{"C4 rank share" if c4 else "C3"}: {n_feet} feet x {n_views} views @{size}^2 per GPU, {N_VERTS}-vertex template, {N_GT_VERTS}-vertex GT '
					   f'scans re-rendered every step, {"sil+pix+chamf losses" if c4 else "silhouette loss"}, backward through rasteriser + MLP, Adam step; '
					   + ('uniform triangulations (Fibonacci-sphere hulls, F = 2V - 4, vertices in Morton order)' if mesh == 'uniform' else 'latitude-longitude grids (pole slivers: binning stress case)'),
		   'mesh': mesh, 'feet_per_gpu': n_feet, 'views': n_views, 'parallelism': f'dp{run.world}'}
	out = line(run.world * n_feet * N_VERTS * n_views / (ms * 1e-3), ms, run, steps, warmup, cfg, ms_per_step_repeats=[round(x, 4) for x in ms_all], device_mallocs_in_timed_steps=n_malloc)
	if bucket is not None:
		bucket.close()
	if with_cpu and run.world == 1 and not c4:
		out['cpu_baseline'] = c3_cpu(mwl, gv, gf, R, T, size)
	return out


def c3_cpu(mwl, gv, gf, R, T, size):
	"""Oracle composition of the C3 step on one foot x one view: MLP (torch-CPU), C rasteriser for both meshes, torch restatement of the
	soft silhouette for the gradient, backward to the MLP weights."""
	from oracle import mlp_ref, render_ref
	m = mwl.model
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	lat = {k: getattr(m, k).data.detach().cpu()[:1].clone().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec', 'reg')}
	Rc, Tc = R[:1].cpu(), T[:1].cpu()
	rp = render_ref.default_params(size)

	def one():
		t0 = time.perf_counter()
		res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
		gt = render_ref.render(gv[:1].cpu().numpy(), gf.cpu().numpy(), None, Rc.numpy(), Tc.numpy(), image_size=size, want_image=False)['mask']
		vproj = render_ref.project(rp, res['verts'].detach().numpy(), Rc.numpy(), Tc.numpy())
		p2f, _, _, _ = render_ref.rasterize(vproj, tf.numpy(), 1, size, size, 100, rp.sil_blur_radius)
		mask = render_ref.torch_mask(rp, res['verts'], tf, Rc, Tc, torch.from_numpy(p2f).long(), 1)
		((mask - torch.from_numpy(gt)) ** 2).mean().backward()
		return time.perf_counter() - t0

	best, cores, how = best_of_cpu(one)
	return dict(value=N_VERTS / best, unit=UNIT, cores=cores, kind='port',
				sample=f'1 foot x 1 view of the same step without the optimiser update, oracle (torch-CPU MLP + oracle/raster_ref.c OpenMP + '
					   f'torch autograd through the K=100 fragments), {how}')


def subpaths(with_cpu):
	"""Sub-path lines (tools/bench_paths.py workloads).  The CPU leg times the oracle on one foot / one image of the same inputs."""
	sys.path.insert(0, os.path.join(ROOT, 'tools'))
	import bench_paths

	def cpu(kind, **kw):
		from oracle import geom_ref, render_ref
		t0 = time.perf_counter()
		if kind == 'render':
			render_ref.render(kw['verts'], kw['faces'], kw['colors'], kw['R'], kw['T'], image_size=kw['size'], want_image=kw['want_image'])
		elif kind == 'chamfer':
			geom_ref.chamfer_distance(kw['x'], kw['y'])
		else:
			geom_ref.mesh_smoothness(kw['verts'], kw['faces'])
		return time.perf_counter() - t0

	if not torch.cuda.is_available():
		raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
	for r in bench_paths.run_all(cpu if with_cpu else None):
		emit(r)


# ------------------------------------------------------------------------------------------------ records of the N=1 line
def brief(rec, *keys):
	out = {k: rec[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype') if k in rec}
	out['workload'] = rec['config']['workload']
	for k in keys:
		if k in rec['config']:
			out[k] = rec['config'][k]
	for k in ('cpu_baseline', 'roofline', 'note', 'ms_per_step_repeats', 'device_mallocs_in_timed_steps'):
		if k in rec:
			out[k] = rec[k]
	if 'cpu_baseline' in out:
		out['x_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
	return out


def train3d_b1_records(run, with_cpu, steps=300, warmup=30, graph=True, only=None):
	"""The reference's literal batch size (batch_size_train = batch_size_val = 1, opts.py:40-41) with label-addressed latents: 16 scans of 8
	feet visited round-robin, one scan per step -- the three stages of train.py (registration, network, latent refinement), the network and
	latent stages also as one HIP-graph replay per step, the latent stage also with the network frozen."""
	recs = {}
	variants = [('net', False, 'train3d_b1'), ('reg', False, 'train3d_b1_reg_stage'), ('latent', False, 'train3d_b1_latent_stage'),
				('latent', True, 'train3d_b1_latent_stage_frozen')]
	cpu_net = None
	for stage, frozen, key in variants:
		if only and key not in only:
			continue
		note(f'record {key}')
		su = train3d_setup(run, 16, 1, stage=stage, labels=True, dp=False, frozen=frozen)
		ms = run.timed(su['step'], steps, warmup)
		# host time to enqueue one step with an empty queue: is the step GPU-bound?
		ts = []
		from find_amd.train_utils import backward_on_this_thread
		with backward_on_this_thread():
			for _ in range(20):
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				su['step']()
				ts.append(time.perf_counter() - t0)
		torch.cuda.synchronize()
		fl = train3d_executed_flops(1, stage=stage, frozen=frozen)
		rec = line(N_VERTS / (ms * 1e-3), ms, run, steps, warmup, {'workload': train3d_workload(1, stage, True, frozen), 'flops_executed_per_step': fl,
																   'step_tflops_executed': fl / (ms * 1e-3) / 1e12},
				   host_enqueue_ms_per_step=sorted(ts)[len(ts) // 2] * 1e3)
		if with_cpu and stage != 'latent':
			rec['cpu_baseline'] = train3d_cpu(su['mwl'], su['gt'], stage=stage, sample_feet=1)
			if stage == 'net':
				cpu_net = rec['cpu_baseline']
		elif with_cpu and cpu_net is not None and not frozen:
			# the oracle's composition of a latent-stage step is the network-stage step on the validation rows (same op sequence, the
			# reference computes the weights' gradients there too): the same timing serves
			rec['cpu_baseline'] = dict(cpu_net, sample=cpu_net['sample'] + ' [the network-stage timing: a latent-stage step is the same op sequence on the val rows]')
		recs[key] = brief(rec, 'step_tflops_executed')
		recs[key]['host_enqueue_ms_per_step'] = rec['host_enqueue_ms_per_step']
		recs[key]['note'] = ('eager loop: bound by the host where host_enqueue_ms_per_step exceeds the replay time; find_amd.trainer.Trainer runs this step as '
							 f'ONE HIP-graph replay: records.{key}_graph')
		if graph:
			try:
				note(f'record {key}_graph')
				recs[key + '_graph'] = train3d_b1_graph(run, steps, warmup, stage=stage, frozen=frozen)
				if 'cpu_baseline' in rec:
					recs[key + '_graph']['x_cpu_baseline'] = recs[key + '_graph']['value'] / rec['cpu_baseline']['value']
			except Exception as e:  # a capture failure must not lose the eager numbers
				recs[key + '_graph'] = {'error': f'{type(e).__name__}: {e}'[:300]}
		del su
	return recs


def eager_colour_head_record(run, steps, warmup):
	"""The headline step with the template pass's colour head evaluated in the forward pass, as the reference does (its output is dropped:
	nothing on a 3-D-loss step reads it) -- what the product ran until round 3, and the reference's own operation count."""
	import find_amd.model_with_loss as MWL
	prev = MWL.LAZY_COLOURS
	MWL.LAZY_COLOURS = False
	try:
		su = train3d_setup(run, N_FEET, N_FEET, stage='net', labels=False, seed=run.rank)
		ms = run.timed(su['step'], steps, warmup)
		fl = train3d_executed_flops(N_FEET, lazy_col=False)
	finally:
		MWL.LAZY_COLOURS = prev
	rec = line(run.world * N_FEET * N_VERTS / (ms * 1e-3), ms, run, steps, warmup,
			   {'workload': train3d_workload(N_FEET, 'net', False) + "; template pass's colour head eager (FIND_LAZY_COLOURS=0)",
				'flops_executed_per_step': fl, 'step_tflops_executed': fl / (ms * 1e-3) / 1e12,
				'step_frac_of_fp32_mfma_peak_executed': fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS})
	return brief(rec, 'flops_executed_per_step', 'step_tflops_executed', 'step_frac_of_fp32_mfma_peak_executed')


def fp32_mfma_record(run, steps, warmup):
	"""The headline step with the 256 -> 256 layers on the fp32 MFMA kernels (gemm4 / dw4: round 3's arithmetic) instead of bf16x3, and the isolated
	Linear kernel of that mode: what the default is compared with."""
	from find_amd import functional as FF
	prev = FF.set_mlp_precision('fp32')
	try:
		su = train3d_setup(run, N_FEET, N_FEET, stage='net', labels=False, seed=run.rank)
		ms = run.timed(su['step'], steps, warmup)
		fl = train3d_executed_flops(N_FEET)
		rec = line(run.world * N_FEET * N_VERTS / (ms * 1e-3), ms, run, steps, warmup,
				   {'workload': train3d_workload(N_FEET, 'net', False) + "; fp32 MFMA kernels (FIND_MLP_PRECISION=fp32)",
					'flops_executed_per_step': fl, 'step_tflops_executed': fl / (ms * 1e-3) / 1e12,
					'step_frac_of_fp32_mfma_peak_executed': fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS})
		del su
		rec['roofline'] = dominant_roofline(run.dev)
	finally:
		FF.set_mlp_precision(prev)
	return brief(rec, 'flops_executed_per_step', 'step_tflops_executed', 'step_frac_of_fp32_mfma_peak_executed')


def train3d_b1_graph(run, steps, warmup, stage='net', frozen=False):
	"""The batch-1 step as ONE HIP graph (find_amd/graph.py): sampling, forward, backward and the fused Adam step captured once,
	replayed per step with the scan copied into the graph's static buffers -- the host enqueues one graph instead of a few hundred kernels.
	This is what find_amd.trainer.Trainer runs on every epoch that writes no PNG."""
	from find_amd.graph import GraphedStep
	su = train3d_setup(run, 16, 1, stage=stage, labels=True, dp=False, capturable=True, frozen=frozen)
	gs = GraphedStep(su['mwl'], su['opts'], [su['opt']], **su['flags'])
	state = dict(i=0)

	def step():
		gs(su['batches'][state['i'] % len(su['batches'])])
		state['i'] += 1

	ms = run.timed(step, steps, warmup)
	rec = line(N_VERTS / (ms * 1e-3), ms, run, steps, warmup, {'workload': train3d_workload(1, stage, True, frozen) + '; one HIP graph replay per step'})
	return brief(rec)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=30)
	ap.add_argument('--warmup', type=int, default=5)
	ap.add_argument('--no-cpu-baseline', action='store_true')
	ap.add_argument('--repeats', type=int, default=5, help='repetitions of the headline K-step timing; the median is reported, all are listed (5: a host that hiccups during two of three repetitions moved a 1.53-ms step to 1.71)')
	ap.add_argument('--no-prime', action='store_true', help='no untimed priming phase before the warm-up steps (Run.timed): measures a cold GPU')
	ap.add_argument('--headline-only', action='store_true', help='only the timed headline loop (no records, no CPU leg, no isolated kernel loop): the command to put under rocprofv3')
	ap.add_argument('--no-graph', action='store_true', help='skip the HIP-graph variant of the batch-1 record')
	ap.add_argument('--no-records', action='store_true', help='skip the nested records of the other configurations')
	ap.add_argument('--train3d', action='store_true', help='(default) the headline line: the reference training configuration')
	ap.add_argument('--train3d-b1', action='store_true', help='instead of the headline line: only the batch-1 train_3d records')
	ap.add_argument('--b1-only', default='', help='with --train3d-b1: comma-separated record names to run (diagnosis)')
	ap.add_argument('--c2', action='store_true', help='instead of the headline line: BASELINE configs[1] (16 feet x 6890-vertex template, MLP fwd+bwd); round 1\'s headline')
	ap.add_argument('--c3', action='store_true', help='instead of the headline line: BASELINE configs[2] end to end (16 feet x 4 views @256^2, silhouette render loss)')
	ap.add_argument('--c5', action='store_true', help='BASELINE configs[4] geometry: the C2 workload on the 50 002-vertex dense template (fp32 unless --fp16)')
	ap.add_argument('--fp16', action='store_true', help="opt-in reduced precision (find_amd.functional.set_mlp_precision('fp16')): the 256->256 layers on the fp16 matrix pipe with fp32 accumulation -- BASELINE configs[4] with --c5; NOT the parity path, never the default line")
	ap.add_argument('--mesh', default='uniform', choices=('uniform', 'latlong'), help='--c3 / --c4: triangulation of template and scans (latlong: the pole-sliver stress case)')
	ap.add_argument('--c4', action='store_true', help='per-rank share of BASELINE configs[3]: 16 feet x 4 views @512^2, silhouette + pixel + Chamfer losses')
	ap.add_argument('--dp-overhead', action='store_true', help='diagnostic: run the C2 step on ONE GPU through the data-parallel code path (one-rank RCCL group, gradient bucket + all-reduce)')
	ap.add_argument('--subpaths', action='store_true', help='instead of the headline line: one JSON line per render / Chamfer / smoothness sub-path (SURVEY 8d), CPU oracle timed beside each')
	args = ap.parse_args()

	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
		raise SystemExit(spawn_ranks(sys.argv[1:], args.gpus))

	if args.no_prime:
		Run.PRIME_S = 0.0
	with_cpu = not args.no_cpu_baseline and not args.headline_only
	if args.subpaths:
		return subpaths(with_cpu)
	run = Run(args.gpus)
	from find_amd import functional as FF
	FF.set_mlp_precision(FF.get_mlp_precision())   # (also hands the process default -- bf16x3 -- to the context's knob, which the isolated-kernel entry points read)
	if args.fp16 and not (args.c2 or args.c5):
		FF.set_mlp_precision('fp16')

	if args.c2 or args.c5 or args.dp_overhead:
		out = c2_record(run, args.steps, args.warmup, n_verts=50002 if args.c5 else None, fp16=args.fp16, with_cpu=with_cpu, dp_overhead=args.dp_overhead)
		if run.rank == 0:
			if not args.headline_only:
				from find_amd import functional as FF
				prev = FF.set_mlp_precision('fp16') if args.fp16 else None
				out['roofline'] = dominant_roofline(run.dev, fp16=args.fp16)
				if prev is not None:
					FF.set_mlp_precision(prev)
			emit(out)
		return run.finish()
	if args.c3 or args.c4:
		out = c3_record(run, args.steps, args.warmup, with_cpu, c4=args.c4, mesh=args.mesh)
		if run.rank == 0:
			emit(out)
		return run.finish()
	if args.train3d_b1:
		recs = train3d_b1_records(run, with_cpu, graph=not args.no_graph, only=set(args.b1_only.split(',')) if args.b1_only else None)
		if run.rank == 0:
			emit(recs)
		return run.finish()

	# ---- headline: train_3d.yaml network-stage step, 16 feet per GPU
	note('headline')
	su = train3d_setup(run, N_FEET, N_FEET, stage='net', labels=False, seed=run.rank)
	# The headline's K-step timing is taken `--repeats` times (default 5; each: W warm-up steps, exactly K steps between barrier +
	# synchronize, the maximum over ranks) and the MEDIAN is reported, every repetition listed beside it (`ms_per_step_repeats`): a step whose
	# host work is most of its GPU time is sensitive to whatever else the host does, and a run that drifts shows in the list.
	ms_all = [run.timed(su['step'], args.steps, args.warmup, prime=(i == 0)) for i in range(max(1, args.repeats))]
	ms = sorted(ms_all)[(len(ms_all) - 1) // 2]
	coll = su['collective']() if run.world > 1 else None
	full = {}   # everything that does not fit the line: RECORDS_FILE
	if run.rank == 0:
		fl = train3d_executed_flops(N_FEET)
		cfg = {'workload': short(f'train_3d.yaml network-stage step: {N_FEET} feet x {N_VERTS}-vertex template per GPU, {N_GT_VERTS}-vertex GT scans, '
								 'chamf(5000)+smooth+texture(1000) losses, backward, Adam(main_params) step; nothing rendered, views:=1'),
			   'feet_per_gpu': N_FEET, 'template_verts': N_VERTS, 'parallelism': f'dp{run.world}',
			   'flops_executed_per_step': fl, 'step_tflops_executed': fl / (ms * 1e-3) / 1e12,
			   'step_frac_of_fp32_mfma_peak_executed': fl / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
		out = line(run.world * N_FEET * N_VERTS / (ms * 1e-3), ms, run, args.steps, args.warmup, cfg, dtype_note=dtype_note(),
				   ms_per_step_repeats=[round(x, 4) for x in ms_all])
		full['headline_notes'] = {
			'workload': train3d_workload(N_FEET, 'net', False),
			'flops_note': 'Linear-layer flops of the step as executed (bench.py: train3d_executed_flops): shared-template main pass without its colour head '
						  '(no loss of this configuration reads the colours of the predicted mesh: the head is evaluated when res["col"] / meshes.textures is first '
						  'read, model.get_meshes(lazy_colours=True); the reference computes it and drops it -- records.train3d_b16_eager_colour_head times that) '
						  '+ the 16 x 1000-sample texture pass',
			'reference_config': 'cfgs/train_3d.yaml:17-27 (chamf_loss, smooth_loss, texture_loss, use_pose_code, use_latent_labels); '
								'src/train/opts.py:40 batch_size_train=1 -> records.train3d_b1; 16 feet per GPU is the data-parallel shard of SURVEY 8e'}
		if coll is not None:
			out['collective'] = coll
		if not args.headline_only:
			rf = dominant_roofline(run.dev, fp16=args.fp16)
			full['roofline'] = dict(rf)
			if not args.fp16:
				full['roofline_kernels'] = roofline_kernels(run.dev, rf['avg_kernel_ms'])
			keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_kernel_ms', 'flops_per_launch', 'bytes_per_launch', 'traffic',
					'fp32_equivalent_tflops', 'x_fp32_mfma_peak')
			out['roofline'] = {k: (short(rf[k], 120) if isinstance(rf[k], str) else rf[k]) for k in keep if k in rf}
			if with_cpu and run.world == 1:
				cb = train3d_cpu(su['mwl'], su['gt'], stage='net', sample_feet=1)
				full['cpu_baseline'] = dict(cb)
				out['cpu_baseline'] = dict(cb, sample=short(cb['sample'], 200))
				out['x_cpu_baseline'] = out['value'] / cb['value']
	del su
	if run.world == 1 and not args.headline_only and not args.no_records:
		recs = {}

		def add(name, rec):
			recs[name] = rec
			emit_record(name, rec)

		try:
			note('record train3d_b16_eager_colour_head')
			add('train3d_b16_eager_colour_head', eager_colour_head_record(run, args.steps, args.warmup))
			note('record fp32_mfma')
			add('fp32_mfma', fp32_mfma_record(run, args.steps, args.warmup))
			for k, v in train3d_b1_records(run, with_cpu, graph=not args.no_graph).items():
				add(k, v)
			note('record c2')
			add('c2', brief(c2_record(run, 30, 5, with_cpu=with_cpu), 'step_tflops_executed', 'step_frac_of_fp32_mfma_peak_executed', 'step_tflops_reference_equiv'))
			note('record c3')
			add('c3', brief(c3_record(run, 20, 3, with_cpu)))
			note('record c4_rank_share')
			add('c4_rank_share', brief(c3_record(run, 10, 3, False, c4=True)))
			note('records *_latlong_stress')
			add('c3_latlong_stress', brief(c3_record(run, 20, 3, False, mesh='latlong')))
			add('c4_rank_share_latlong_stress', brief(c3_record(run, 10, 3, False, c4=True, mesh='latlong')))
			note('record c5')
			add('c5_fp32', brief(c2_record(run, 10, 3, n_verts=50002), 'step_tflops_executed', 'step_frac_of_fp32_mfma_peak_executed'))
			add('c5_fp16', brief(c2_record(run, 10, 3, n_verts=50002, fp16=True), 'step_tflops_executed'))
		except Exception as e:   # a failing record must not lose the headline (its own tests cover each record's path)
			recs['error'] = f'{type(e).__name__}: {e}'[:300]
			note(f'records stopped: {recs["error"]}')
		full['records'] = recs
		# on the line: name -> [ms_per_step, steps] only
		out['records'] = {k: [round(v['ms_per_step'], 4), v['steps']] for k, v in recs.items() if isinstance(v, dict) and 'ms_per_step' in v}
		if 'error' in recs:
			out['records_error'] = recs['error']
	if run.rank == 0:
		if full and not args.headline_only:
			full['line'] = out
			for path in (RECORDS_FILE, os.path.join(ROOT, 'gpurun_out', 'bench_records.json')):
				try:
					if os.path.isdir(os.path.dirname(path)):
						with open(path, 'w') as f:
							json.dump(full, f, indent=1)
				except OSError as e:
					note(f'could not write {path}: {e}')
			out['records_file'] = os.path.basename(RECORDS_FILE)
		text = json.dumps(out)
		if len(text) >= LINE_LIMIT:   # never again a line the driver cannot parse: shed the optional keys, in this order
			for k in ('records', 'dtype_note', 'ms_per_step_repeats', 'x_cpu_baseline'):
				out.pop(k, None)
				if len(json.dumps(out)) < LINE_LIMIT:
					break
		emit(out)
	run.finish()


if __name__ == '__main__':
	main()
