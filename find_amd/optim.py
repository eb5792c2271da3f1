"""Fused optimisers for FIND's training step: drop-in replacements for the three optimisers the reference
builds in src/train/train.py:161-168 --

	optim_network = Adam(model.model.main_params, lr=args.lr_net)
	optim_reg     = SGD(model.model.reg_params, lr=args.lr_reg, momentum=0.9)
	optim_latent  = Adam(model.model.latent_params, lr=args.lr_latent)

Same constructor keywords, param_groups, state keys ('step', 'exp_avg', 'exp_avg_sq' / 'momentum_buffer') and state_dict layout
as torch.optim, so checkpoints and schedulers keep working; step() updates every tensor of a param group in ONE launch of
libfind_hip.so (find_adam_step / find_sgd_step).  Dense updates with torch's arithmetic; amsgrad / maximize are not provided.
There is no CPU fallback: parameters must live on the GPU."""
import ctypes

import torch

from . import _lib
from ._lib import check, current_stream


def _ptr_array(tensors):
	return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _i64_array(vals):
	arr = (ctypes.c_int64 * len(vals))()
	for i, v in enumerate(vals):
		arr[i] = int(v)
	return arr


def _check_tensors(name, tensors):
	for t in tensors:
		if not t.is_cuda:
			raise RuntimeError(f'find_amd.optim.{name}: tensors must be on a ROCm device; there is no CPU fallback')
		if t.dtype != torch.float32 or not t.is_contiguous():
			raise RuntimeError(f'find_amd.optim.{name}: contiguous fp32 tensors required')


class Adam(torch.optim.Optimizer):
	"""torch.optim.Adam(params, lr, betas, eps, weight_decay) with the update of all tensors fused into one kernel."""

	def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False, capturable=False):
		"""capturable=True (torch's keyword): state['step'] lives on the device and the bias corrections are formed there in fp32, so a
		step can be captured in a HIP graph and replayed (find_amd.graph.GraphedStep); the default keeps torch's host arithmetic."""
		if amsgrad or maximize:
			raise NotImplementedError('find_amd.optim.Adam: amsgrad / maximize are not provided (the reference uses neither)')
		if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
			raise ValueError('find_amd.optim.Adam: invalid hyper-parameter')
		super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, capturable=bool(capturable)))
		self._cache = {}

	def load_state_dict(self, state_dict):
		super().load_state_dict(state_dict)
		self._cache = {}

	def state_dict(self):
		"""torch's layout.  capturable: the tensors of a bucket share ONE device step counter inside the optimiser (_prepare); what is handed
		out holds a copy per parameter, as torch's does -- no aliasing in a checkpoint, and a later step() does not move a saved count."""
		sd = super().state_dict()
		sd['state'] = {k: ({**st, 'step': st['step'].clone()} if torch.is_tensor(st.get('step')) else dict(st)) for k, st in sd['state'].items()}
		return sd

	def _prepare(self, gi, ps):
		"""Per-group launch tables, rebuilt only when the set of parameters with a gradient (or the state) changes: the pointer
		arrays of parameters and moments are stable across steps, only the gradients' addresses move."""
		capturable = bool(self.param_groups[gi].get('capturable', False))
		key = (capturable,) + tuple(id(p) for p in ps)
		c = self._cache.get(gi)
		if c is not None and c['key'] == key:
			return c
		for p in ps:
			st = self.state[p]
			if len(st) == 0:
				st['step'] = torch.zeros((), dtype=torch.float32, device=p.device) if capturable else torch.tensor(0.0, dtype=torch.float32)
				st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
				st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
		# tensors that have taken the same number of steps go out together (normally: all of them)
		buckets = {}
		for p in ps:
			buckets.setdefault(int(self.state[p]['step'].item()), []).append(p)
		c = dict(key=key, buckets=[])
		for step, plist in buckets.items():
			m = [self.state[p]['exp_avg'] for p in plist]
			v = [self.state[p]['exp_avg_sq'] for p in plist]
			_check_tensors('Adam', plist + m + v)
			bk = dict(step=step, plist=plist, steps=[self.state[p]['step'] for p in plist], parr=_ptr_array(plist), marr=_ptr_array(m),
					  varr=_ptr_array(v), numel=_i64_array([p.numel() for p in plist]))
			if capturable:
				# one device counter per bucket; every tensor's state['step'] becomes that same tensor (torch keeps one per parameter
				# and bumps them with a foreach add: one launch here, and the state_dict layout is unchanged)
				cnt = torch.full((), float(step), dtype=torch.float32, device=plist[0].device)
				for p in plist:
					self.state[p]['step'] = cnt
				bk['steps'] = [cnt]
				bk['step_dev'] = cnt
			c['buckets'].append(bk)
		self._cache[gi] = c
		return c

	@torch.no_grad()
	def step(self, closure=None):
		loss = None
		if closure is not None:
			with torch.enable_grad():
				loss = closure()
		L = _lib.lib()
		for gi, group in enumerate(self.param_groups):
			ps = [p for p in group['params'] if p.grad is not None]
			if not ps:
				continue
			b1, b2 = group['betas']
			for bk in self._prepare(gi, ps)['buckets']:
				plist = bk['plist']
				grads = [p.grad for p in plist]
				for i, gr in enumerate(grads):
					if not gr.is_cuda or gr.dtype != torch.float32:
						raise RuntimeError('find_amd.optim.Adam: gradients must be fp32 tensors on a ROCm device; there is no CPU fallback')
					if not gr.is_contiguous():
						grads[i] = gr.contiguous()
				bk['step'] += 1
				torch._foreach_add_(bk['steps'], 1)
				if 'step_dev' in bk:
					check(L.find_adam_step_dev(len(plist), bk['parr'], _ptr_array(grads), bk['marr'], bk['varr'], bk['numel'], float(group['lr']), float(b1),
											   float(b2), float(group['eps']), float(group['weight_decay']), ctypes.c_void_p(bk['step_dev'].data_ptr()),
											   current_stream(plist[0].device)), 'find_adam_step_dev')
					continue
				check(L.find_adam_step(len(plist), bk['parr'], _ptr_array(grads), bk['marr'], bk['varr'], bk['numel'], float(group['lr']), float(b1),
									   float(b2), float(group['eps']), float(group['weight_decay']), bk['step'], current_stream(plist[0].device)), 'find_adam_step')
		return loss


class SGD(torch.optim.Optimizer):
	"""torch.optim.SGD(params, lr, momentum, dampening, weight_decay, nesterov) fused over all tensors of a group."""

	def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, maximize=False):
		if maximize:
			raise NotImplementedError('find_amd.optim.SGD: maximize is not provided')
		if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
			raise ValueError('find_amd.optim.SGD: invalid hyper-parameter')
		if nesterov and (momentum <= 0 or dampening != 0):
			raise ValueError('Nesterov momentum requires a momentum and zero dampening')
		super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov, maximize=False))

	@torch.no_grad()
	def step(self, closure=None):
		loss = None
		if closure is not None:
			with torch.enable_grad():
				loss = closure()
		L = _lib.lib()
		for group in self.param_groups:
			ps = [p for p in group['params'] if p.grad is not None]
			if not ps:
				continue
			mom = group['momentum']
			# tensors without a buffer yet take torch's "clone the gradient" first step
			fresh = [p for p in ps if mom != 0 and 'momentum_buffer' not in self.state[p]]
			seasoned = [p for p in ps if not (mom != 0 and 'momentum_buffer' not in self.state[p])]
			for first, plist in ((True, fresh), (False, seasoned)):
				if not plist:
					continue
				bufs = None
				if mom != 0:
					for p in plist:
						if 'momentum_buffer' not in self.state[p]:
							self.state[p]['momentum_buffer'] = torch.empty_like(p, memory_format=torch.preserve_format)
					bufs = [self.state[p]['momentum_buffer'] for p in plist]
				grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
				_check_tensors('SGD', plist + grads + (bufs or []))
				check(L.find_sgd_step(len(plist), _ptr_array(plist), _ptr_array(grads), _ptr_array(bufs) if bufs else None,
									  _i64_array([p.numel() for p in plist]), float(group['lr']), float(mom), float(group['dampening']),
									  float(group['weight_decay']), int(bool(group['nesterov'])), int(first), current_stream(plist[0].device)), 'find_sgd_step')
		return loss
