"""The loops around the step, driven the way the reference drives them on cfgs/train_3d.yaml:

* train_network's keyword dictionary for epoch 0 of stages 2 and 3 (src/train/train.py:52-70: save_renders and render_foot are ON at
  epoch 0 because a checkpoint is saved there and --no_rendering is off), fed through the inner bodies of Trainer.train_epoch
  (src/train/trainer.py:96-123) and Trainer.val_epoch (:150-163);
* stage 3, latent refinement (train.py:217-224: val_only -> only val_epoch runs, is_train=False, the `*_val` tables, Adam(latent_params)):
  losses and the gradients of the addressed val rows against the oracle's composition, with the network's weights trainable (as the
  reference leaves them) and frozen (latents-only backward);
* find_amd.trainer.Trainer: the same epochs as HIP-graph replays and eagerly, same parameters afterwards."""
import os

import numpy as np
import pytest
import torch

from oracle import geom_ref, mlp_ref

pytestmark = pytest.mark.gpu

from test_gpu_train3d import FixedDraws, ITEM, ROWS, _oracle_inputs, _setup   # noqa: E402

VAL_ROWS = dict(shapevec=0, texvec=0, posevec=1, reg=1)   # scan 9000-B of foot 9000 (synthetic.scan_labels)


def _train3d_opts(opts):
	"""cfgs/train_3d.yaml: COMMON_ARGS.copy_over_masking + the FIND experiment's switches (the loss flags are set by _setup)."""
	opts.set_option('copy_over_masking', True)
	opts.set_option('reg', True)
	opts.set_option('latent_epochs', 1000)
	return opts


def _val_batch(gv, gf, gc, i=0, scan='9000-B'):
	from find_amd.structures import Meshes, TexturesVertex
	return dict(mesh=Meshes(gv[i:i + 1].contiguous(), gf, TexturesVertex(gc[i:i + 1].contiguous())), idx=torch.tensor([1], device='cuda'),
				name=[scan], shape=['9000'], tex=['9000'], pose=[scan], reg=[scan])


def _fill_val_tables(m, seed=5):
	from find_amd import synthetic
	lat = synthetic.latents(2, seed=seed, device='cuda')
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			t = getattr(m, k + '_val').data
			t.copy_(lat[k][:t.shape[0]])


def _draws(n_verts, gt_verts, seed=9):
	g = torch.Generator().manual_seed(seed)
	F_gt, F_t = 2 * (gt_verts - 2), 2 * (n_verts - 2)
	return [(torch.randint(0, F_gt, (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			(torch.randint(0, F_t, (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			(torch.randint(0, F_gt, (1, 1000), generator=g).cuda(), torch.rand(1, 1000, 2, generator=g).cuda())]


def test_reference_call_sequence_epoch0_of_stages_2_and_3(tmp_path):
	"""No keyword the reference passes on train_3d.yaml is refused; the PNG strip is written and holds the rendered images; the losses
	are those of the same step without rendering."""
	from PIL import Image
	from find_amd import optim
	from find_amd.trainer import stage_model_kwargs
	from find_amd.train_utils import batch_to_device, sample_latent_vectors
	n_verts = gt_verts = 1002
	mwl, opts, batch_of, (gv, gf, gc), optim_network = _setup(n_verts, gt_verts)
	opts = _train3d_opts(opts)
	m = mwl.model
	_fill_val_tables(m)
	optim_latent = optim.Adam(m.latent_params, lr=opts.lr_latent)
	draws = _draws(n_verts, gt_verts)

	# ---- stage 2, epoch 0 (train.py:211-215 -> train_network(num_epochs=net_epochs, save_every=net_save_every))
	epoch = 0
	model_kwargs, save_model = stage_model_kwargs(opts, epoch, opts.net_epochs, opts.net_save_every, render_dir=str(tmp_path))
	assert save_model and model_kwargs['save_renders'] and model_kwargs['render_foot']   # a checkpoint epoch with rendering on
	assert model_kwargs['copy_mask_out'] is True and model_kwargs['mask_out_pred_faces'] is False and model_kwargs['gt_z_cutoff'] is None
	os.makedirs(model_kwargs['render_dir'], exist_ok=True)
	# Trainer.train_epoch's body (trainer.py:93-123)
	model_kwargs['is_train'] = True
	batch = batch_of(ITEM)
	batch.update(**sample_latent_vectors(batch, m.latent_vectors_train))
	batch = batch_to_device(batch, 'cuda')
	[o.zero_grad() for o in [optim_network]]
	np.random.seed(11)
	with FixedDraws(draws):
		loss, loss_dict = mwl(batch, epoch, opts=opts, **model_kwargs)
	assert not (loss == 0)
	loss.backward()
	[o.step() for o in [optim_network]]
	assert set(loss_dict) == {'loss_chamf', 'loss_smooth', 'loss_tex'}
	png = os.path.join(model_kwargs['render_dir'], f'{epoch:04d}_{ITEM:02d}.png')
	assert os.path.isfile(png)
	strip = np.asarray(Image.open(png))
	M, H = opts.num_views, 256
	assert strip.shape == (M * H, 2 * H, 3) and strip.dtype == np.uint8   # views top to bottom, GT | prediction

	# the same step without any rendering gives the same losses (fresh model: the optimiser has stepped the first one)
	mwl2, opts2, batch_of2, _, _ = _setup(n_verts, gt_verts)
	b2 = batch_of2(ITEM)
	b2.update(**sample_latent_vectors(b2, mwl2.model.latent_vectors_train))
	with FixedDraws(draws):
		loss2, ld2 = mwl2(b2, epoch, opts=opts2, **opts2.net_train_kwargs(), is_train=True)
	assert torch.equal(loss.detach(), loss2.detach())
	for k in loss_dict:
		assert torch.equal(loss_dict[k].detach(), ld2[k].detach()), k
	# ... and the strip is the images return_renders hands out for the same camera draws: 8 bits by truncation of 255 * value
	np.random.seed(11)
	with FixedDraws(draws):
		_, _, rdr = mwl2(b2, epoch, opts=opts2, **opts2.net_train_kwargs(), render_foot=True, return_renders=True, copy_mask_out=True)
	want = np.hstack([np.vstack(rdr[k]['image'].detach().reshape(-1, H, H, 3).cpu().numpy()) for k in ('gt', 'pred')])
	want = (want * 255).astype(np.uint8)
	assert (strip.astype(int) - want.astype(int)).__abs__().max() <= 1   # (the two models differ by nothing: same seeds, same draws)
	assert strip[:, :H].min() < 255 and strip[:, H:].min() < 255            # both columns show a foot, not a white page

	# ---- stage 3, epoch 0 (train.py:217-224: val_only=True -> train_epoch is skipped, val_epoch runs with render_dir .../val)
	model_kwargs, save_model = stage_model_kwargs(opts, 0, opts.latent_epochs, opts.latent_save_every, render_dir=str(tmp_path))
	model_kwargs['render_dir'] = os.path.join(str(tmp_path), 'val')
	os.makedirs(model_kwargs['render_dir'], exist_ok=True)
	assert model_kwargs['save_renders'] and model_kwargs['render_foot']
	# Trainer.val_epoch's body (trainer.py:147-163)
	model_kwargs['is_train'] = False
	batch = _val_batch(gv, gf, gc)
	batch.update(**sample_latent_vectors(batch, m.latent_vectors_val))
	batch = batch_to_device(batch, 'cuda')
	assert set(k for k in batch if k.endswith('_val')) == {'shapevec_val', 'posevec_val', 'texvec_val', 'reg_val'}
	before = {n: p.detach().clone() for n, p in m.named_parameters()}
	optim_latent.zero_grad()
	loss, loss_dict = mwl(batch, 0, opts=opts, **model_kwargs)
	loss.backward()
	optim_latent.step()
	assert os.path.isfile(os.path.join(model_kwargs['render_dir'], '0000_01.png'))
	after = dict(m.named_parameters())
	# Adam(latent_params) moves the addressed val rows of shape / tex / pose, nothing of the network, nothing of the train tables, no reg row
	for k, r in VAL_ROWS.items():
		n = f'{k}_val.data'
		if k == 'reg':
			assert torch.equal(after[n], before[n])
		else:
			assert not torch.equal(after[n][r], before[n][r]), n
	for n in ('base.2.weight', 'mlp_disp.0.weight', 'mlp_col.6.bias', 'shapevec.data', 'posevec.data', 'reg.data'):
		assert torch.equal(after[n], before[n]), n


def _val_oracle(mwl, gv, gf, gc, draws_cpu, item=0):
	m = mwl.model
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	lat = {k: getattr(m, k + '_val').data.detach().cpu()[r:r + 1].clone().requires_grad_(True) for k, r in VAL_ROWS.items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	(fi_gt, uv_gt), (fi_pr, uv_pr), (fi_tx, uv_tx) = draws_cpu
	res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
	gvc, gfc, gcc = gv[item:item + 1].cpu(), gf.cpu(), gc[item:item + 1].cpu()
	gt_s = geom_ref.sample_points(gvc, gfc, fi_gt, uv_gt)
	pr_s = geom_ref.sample_points(res['verts'], tf, fi_pr, uv_pr)
	tx_p, tx_c = geom_ref.sample_points(gvc, gfc, fi_tx, uv_tx, attr=gcc)
	col = mlp_ref.mlp_forward(sd, B, tx_p, lat['shapevec'], lat['texvec'], lat['posevec'])['col']
	mask = (tx_c < 1).any(dim=-1, keepdim=True).expand(-1, -1, 3)
	ref = {'loss_chamf': geom_ref.chamfer_distance(pr_s, gt_s) * 10000., 'loss_smooth': geom_ref.mesh_smoothness(res['verts'], tf) * 1000.,
		   'loss_tex': (torch.nn.functional.mse_loss(col, tx_c, reduction='none') * mask).mean()}
	return sd, lat, ref


def _check_val_rows(m, lat, tol):
	worst = 0.0
	for k, r in VAL_ROWS.items():
		g = getattr(m, k + '_val').data.grad
		assert g is not None, k
		g = g.cpu()
		want = lat[k].grad[0]
		s = max(1e-3, want.abs().max().item())
		err = (g[r] - want).abs().max().item() / s
		worst = max(worst, err)
		assert err < tol, (k, err)
		others = torch.cat([g[:r], g[r + 1:]])
		assert others.numel() == 0 or others.abs().max().item() == 0.0, k
	return worst


def test_latent_refinement_stage_step_matches_oracle():
	"""Stage 3 at full size: mwl(batch, epoch, opts, is_train=False, **net_train_kwargs) on a 6890-vertex template and a 10 002-vertex scan,
	losses and the gradients of the addressed shapevec_val / texvec_val / posevec_val / reg_val rows against the oracle at 1e-4; then the
	same step with the network frozen (requires_grad False: find_mlp_bwd's latents-only path) -- identical latent gradients to the bit
	pattern the full backward gives up to summation order, no weight gradient -- and Adam(latent_params)."""
	from find_amd import optim
	from find_amd.train_utils import sample_latent_vectors
	from test_gpu_pipeline import DrawRecorder
	mwl, opts, batch_of, (gv, gf, gc), _ = _setup()
	m = mwl.model
	_fill_val_tables(m)
	opt = optim.Adam(m.latent_params, lr=1e-4)
	batch = _val_batch(gv, gf, gc)
	batch.update(sample_latent_vectors(batch, m.latent_vectors_val))
	for k, r in VAL_ROWS.items():
		assert torch.equal(batch[f'{k}_val'], getattr(m, k + '_val').data[r:r + 1]), k
	opt.zero_grad(set_to_none=True)
	for p in m.parameters():
		p.grad = None
	with DrawRecorder() as rec:
		loss, losses = mwl(batch, 0, opts, is_train=False, **opts.net_train_kwargs())
	loss.backward()
	draws_cpu = rec.chamfer_and_texture()
	sd, lat, ref = _val_oracle(mwl, gv, gf, gc, draws_cpu)
	for k in ref:
		assert abs(losses[k].item() - ref[k].item()) < 1e-4 * max(1.0, abs(ref[k].item())), (k, losses[k].item(), ref[k].item())
	sum(ref.values()).backward()
	worst = _check_val_rows(m, lat, 1e-4)
	# the reference leaves the network trainable in this stage (nothing freezes it; only the optimiser ignores it): its weights get gradients
	for k in ['base.0.weight', 'mlp_disp.2.weight', 'mlp_col.0.weight']:
		got, want = dict(m.named_parameters())[k].grad.cpu(), sd[k].grad
		err = (got - want).abs().max().item() / max(1e-3, want.abs().max().item())
		worst = max(worst, err)
		assert err < 1e-4, (k, err)
	# the train tables are not part of this step
	assert m.shapevec.data.grad is None and m.reg.data.grad is None
	full = {k: getattr(m, k + '_val').data.grad.clone() for k in VAL_ROWS}
	print(f'latent-stage step: worst gradient error {worst:.2e} of the tensor maximum')

	# ---- frozen network: same draws, latents-only backward
	weights = [p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()]
	for p in weights:
		p.requires_grad_(False)
	for p in m.parameters():   # (reg_val is not among latent_params: opt.zero_grad() alone would leave its gradient to accumulate)
		p.grad = None
	dev_draws = [(a.cuda(), b.cuda()) for a, b in draws_cpu]
	batch.update(sample_latent_vectors(batch, m.latent_vectors_val))   # (a fresh lookup: the first backward freed the graph of the old rows)
	with FixedDraws(dev_draws):
		loss_f, losses_f = mwl(batch, 0, opts, is_train=False, **opts.net_train_kwargs())
	loss_f.backward()
	assert abs(loss_f.item() - loss.item()) < 1e-6 * max(1.0, abs(loss.item()))
	assert all(p.grad is None for p in weights)
	for k in VAL_ROWS:
		g = getattr(m, k + '_val').data.grad
		s = max(1e-3, full[k].abs().max().item())
		assert (g - full[k]).abs().max().item() / s < 2e-5, k   # (summation order of the per-foot column sums)
	_check_val_rows(m, lat, 1e-4)
	before = {k: getattr(m, k + '_val').data.detach().clone() for k in VAL_ROWS}
	opt.step()
	for k, r in VAL_ROWS.items():
		moved = not torch.equal(getattr(m, k + '_val').data[r], before[k][r])
		assert moved == (k != 'reg'), k


def test_frozen_network_backward_at_batch_16_matches_full_backward():
	"""The latents-only path on the shared-template layout (16 feet, one trunk evaluation) and on per-foot positions (the texture pass):
	latent gradients equal those of the full backward."""
	from find_amd import synthetic
	m = synthetic.make_model(1002, train_size=16, val_size=2, device='cuda')
	lat = synthetic.latents(16, seed=3, device='cuda')
	g = torch.Generator().manual_seed(4)
	pos16 = (torch.randn(16, 700, 3, generator=g) * 0.05).cuda()
	for pos in (m.template_verts.data, pos16):
		grads = []
		for frozen in (False, True):
			for p in m.parameters():
				p.grad = None
			for seq in (m.base, m.mlp_disp, m.mlp_col):
				for p in seq.parameters():
					p.requires_grad_(not frozen)
			lv = {k: v.clone().requires_grad_(True) for k, v in lat.items() if k != 'reg'}
			res = m(pos, shapevec=lv['shapevec'], texvec=lv['texvec'], posevec=lv['posevec'])
			w = torch.linspace(0.5, 1.5, res['disp'].numel(), device='cuda').view_as(res['disp'])
			((res['disp'] * w).sum() + (res['col'] ** 2 * w).sum()).backward()
			grads.append({k: v.grad.clone() for k, v in lv.items()})
			if frozen:
				assert all(p.grad is None for p in m.base.parameters())
		for k in grads[0]:
			s = grads[0][k].abs().max().item()
			assert s > 0
			assert (grads[0][k] - grads[1][k]).abs().max().item() / s < 2e-5, (k, tuple(pos.shape))


def _trainer_run(graph, tmp_path, epochs=(0, 1, 2)):
	from find_amd import optim
	from find_amd.trainer import Trainer, stage_model_kwargs
	n_verts = gt_verts = 1002
	mwl, opts, batch_of, (gv, gf, gc), _ = _setup(n_verts, gt_verts, capturable=True)
	opts = _train3d_opts(opts)
	opts.set_option('net_epochs', 3)
	opts.set_option('net_save_every', 2)      # epoch 0 and the last one save (and render); epoch 1 does not
	opts.set_option('num_views', 2)
	m = mwl.model
	_fill_val_tables(m)
	optim_network = optim.Adam(m.main_params, lr=5e-4, capturable=True)
	optim_val = optim.Adam(m.latent_params, lr=1e-3, capturable=True)
	loader = [batch_of(i) for i in (0, 3, 1, 2)]          # scans of varying label rows, one per step (batch_size_train = 1)
	val_loader = [_val_batch(gv, gf, gc, 0, '9000-A'), _val_batch(gv, gf, gc, 1, '9000-B')]
	tr = Trainer([optim_network], mwl, loader, val_loader, opts, latent_vectors_train=m.latent_vectors_train,
				 latent_vectors_val=m.latent_vectors_val, val_optim=optim_val, device='cuda', graph=graph)
	draws = _draws(n_verts, gt_verts)
	modes, msgs = [], []
	with FixedDraws(draws):
		for epoch in epochs:
			kw, save_model = stage_model_kwargs(opts, epoch, opts.net_epochs, opts.net_save_every, render_dir=str(tmp_path / str(graph)))
			np.random.seed(100 + epoch)
			msgs.append(tr.train_epoch(epoch, save_model=save_model, model_kwargs=dict(kw)))
			modes.append(tr.last_mode)
			kw['render_dir'] = str(tmp_path / str(graph) / 'val')
			msg, res = tr.val_epoch(epoch, model_kwargs=dict(kw))
			modes.append(tr.last_mode)
			assert set(res) == {'Loss', 'chamf', 'smooth', 'tex'}
	torch.cuda.synchronize()
	return tr, {n: p.detach().clone() for n, p in m.named_parameters()}, modes


def test_trainer_graph_replays_equal_eager_epochs(tmp_path):
	"""find_amd.trainer.Trainer over label-addressed one-scan batches: checkpoint epochs (save_renders) run eagerly, the others as HIP-graph
	replays; parameters and logged losses equal those of the all-eager loop."""
	tr_g, p_g, modes_g = _trainer_run('auto', tmp_path)
	tr_e, p_e, modes_e = _trainer_run(False, tmp_path)
	assert modes_e == ['eager'] * 6
	assert modes_g == ['eager', 'eager', 'graph', 'graph', 'eager', 'eager'], modes_g   # epochs 0 and 2 save a checkpoint and render
	assert os.path.isfile(tmp_path / 'auto' / 'train' / '0000_00.png') and os.path.isfile(tmp_path / 'auto' / 'val' / '0002_01.png')
	assert not os.path.exists(tmp_path / 'auto' / 'train' / '0001_00.png')
	lr, n_steps = 1e-3, 3 * 4 + 3 * 2
	for n in p_e:
		# Two runs of the SAME loop are not bit-identical (float atomics in the sampling backward), and Adam's first steps move an element by
		# lr * sign(gradient) whatever the gradient's size: an element whose gradient is rounding noise may end a step of lr apart.  So: all but
		# a thousandth of the elements agree to 1e-5, and none is further apart than a few such flips (tools/check_defer.py: two eager runs differ
		# by 2e-6 most of the time and by 7.5e-4 -- one element -- now and then).
		d = (p_g[n] - p_e[n]).abs()
		assert (d > 0.1 * lr).float().mean().item() < 1e-3, (n, (d > 0.1 * lr).float().mean().item())
		assert d.max().item() < 0.1 * lr * n_steps, (n, d.max().item())
	for epoch in (0, 1, 2):
		for part in ('train_loss', 'val_loss'):
			a, b = tr_g.log[epoch][part], tr_e.log[epoch][part]
			assert set(a) == set(b) and ('Loss' in a)
			for k in a:
				assert len(a[k]) == len(b[k]) == (4 if part == 'train_loss' else 2)
				np.testing.assert_allclose(a[k], b[k], rtol=2e-3, atol=1e-6)
	assert set(tr_g.log[0]['train_loss']) == {'Loss', 'Chamf', 'Smooth', 'Tex'}   # pretty_print_loss keys (trainer.py:14-16,126)


# ------------------------------------------------------------------------------------------------ stage 1: registration
def _reg_stage_kwargs(opts):
	"""train.py:201: model_kwargs = dict(chamf=True, smooth=False, gt_z_cutoff=args.gt_z_cutoff)."""
	return dict(chamf=True, smooth=False, gt_z_cutoff=opts.gt_z_cutoff)


@pytest.mark.parametrize('gt_z_cutoff', [None, 0.01])
def test_registration_stage_val_epoch_step_matches_oracle(gt_z_cutoff):
	"""Stage 1's val_epoch body (train.py:197-209 -> trainer.py:150-163) at full size: is_train=False, chamf only, gt_z_cutoff as the stage
	passes it (None by default, opts.py:145; 0.01 cuts the synthetic scan at a third of its height: the ragged-cloud Chamfer of
	losses.py:79-85), val_optim = SGD(reg_params, momentum 0.9).  Loss and the gradient of the addressed reg_val row against the oracle at
	1e-4; after the step that row is reg_val - lr * grad (first momentum step), nothing else has moved."""
	from find_amd import optim
	from find_amd.train_utils import sample_latent_vectors
	from test_gpu_pipeline import DrawRecorder
	mwl, opts, batch_of, (gv, gf, gc), _ = _setup(stage='reg', seed=1)
	opts.set_option('gt_z_cutoff', gt_z_cutoff)
	m = mwl.model
	_fill_val_tables(m)
	lr = 1e-3
	optim_reg = optim.SGD(m.reg_params, lr=lr, momentum=0.9)
	model_kwargs = _reg_stage_kwargs(opts)
	model_kwargs['is_train'] = False
	batch = _val_batch(gv, gf, gc)
	batch.update(**sample_latent_vectors(batch, m.latent_vectors_val))
	before = {n: p.detach().clone() for n, p in m.named_parameters()}
	optim_reg.zero_grad()
	with DrawRecorder() as rec:
		loss, loss_dict = mwl(batch, 0, opts=opts, **model_kwargs)
	assert set(loss_dict) == {'loss_chamf'}
	loss.backward()
	optim_reg.step()
	(fi_gt, uv_gt), (fi_pr, uv_pr) = rec.draws
	sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
	lat = {k: before[f'{k}_val.data'][r:r + 1].cpu().clone().requires_grad_(k == 'reg') for k, r in VAL_ROWS.items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
	gt_s = geom_ref.sample_points(gv[:1].cpu(), gf.cpu(), fi_gt, uv_gt)
	pr_s = geom_ref.sample_points(res['verts'], tf, fi_pr, uv_pr)
	if gt_z_cutoff is not None:
		keep = gt_s[0, :, 2] <= gt_z_cutoff
		assert 0.1 < keep.float().mean().item() < 0.9
		gt_s = gt_s[:, keep]
	rl = geom_ref.chamfer_distance(pr_s, gt_s) * 10000.
	assert abs(loss.item() - rl.item()) < 1e-4 * max(1.0, abs(rl.item())), (loss.item(), rl.item())
	rl.backward()
	g = lat['reg'].grad[0]
	r = VAL_ROWS['reg']
	got = m.reg_val.data.grad[r].cpu()
	assert (got - g).abs().max().item() < 1e-4 * g.abs().max().item(), (got, g)
	want = before['reg_val.data'][r].cpu() - lr * g
	step = (lr * g).abs().max().item()
	assert step > 0 and (m.reg_val.data[r].detach().cpu() - want).abs().max().item() < 1e-3 * step + 1e-7
	for n, p in m.named_parameters():
		if n == 'reg_val.data':
			keep_rows = [i for i in range(p.shape[0]) if i != r]
			assert torch.equal(p[keep_rows], before[n][keep_rows])
		else:
			assert torch.equal(p, before[n]), n   # the train rows, the other val tables and the network: not this stage's business


def _reg_trainer_run(graph, gt_z_cutoff, epochs=3):
	from find_amd import optim
	from find_amd.trainer import Trainer
	n_verts = gt_verts = 1002
	mwl, opts, batch_of, (gv, gf, gc), _ = _setup(n_verts, gt_verts, stage='reg', seed=2)
	opts.set_option('gt_z_cutoff', gt_z_cutoff)
	m = mwl.model
	_fill_val_tables(m)
	optim_reg = optim.SGD(m.reg_params, lr=1e-5, momentum=0.9)   # lr_reg's default (opts.py:55); the Chamfer term is weighted 1e4: 1e-3 diverges
	loader = [batch_of(i) for i in (0, 3, 1, 2)]
	val_loader = [_val_batch(gv, gf, gc, 0, '9000-A'), _val_batch(gv, gf, gc, 1, '9000-B')]
	# train.py:183-186: trainer_reg = Trainer([optim_reg], ..., val_optim=optim_reg)
	tr = Trainer([optim_reg], mwl, loader, val_loader, opts, latent_vectors_train=m.latent_vectors_train, latent_vectors_val=m.latent_vectors_val,
				 val_optim=optim_reg, device='cuda', graph=graph)
	draws = _draws(n_verts, gt_verts)
	modes = []
	with FixedDraws(draws):
		for epoch in range(epochs):
			model_kwargs = _reg_stage_kwargs(opts)
			tr.train_epoch(epoch, model_kwargs=dict(model_kwargs))
			modes.append(tr.last_mode)
			msg, res = tr.val_epoch(epoch, model_kwargs=dict(model_kwargs))
			modes.append(tr.last_mode)
			assert set(res) == {'Loss', 'chamf'}
	torch.cuda.synchronize()
	return tr, {n: p.detach().clone() for n, p in m.named_parameters()}, modes, {n: p.detach().clone() for n, p in _setup(n_verts, gt_verts, stage='reg', seed=2)[0].model.named_parameters()}


@pytest.mark.parametrize('gt_z_cutoff', [None, 0.01])
def test_registration_stage_trainer_graph_equals_eager(gt_z_cutoff):
	"""Stage 1 through find_amd.trainer.Trainer as train.py:197-209 drives it -- train_epoch and val_epoch per epoch, ONE SGD(reg_params)
	as both optimisers: every epoch is HIP-graph replays by default (no PNG is ever written in this stage) and leaves the registration
	rows where the eager loop leaves them; only `reg` / `reg_val` move."""
	tr_g, p_g, modes_g, start = _reg_trainer_run('auto', gt_z_cutoff)
	tr_e, p_e, modes_e, _ = _reg_trainer_run(False, gt_z_cutoff)
	assert modes_g == ['graph'] * 6 and modes_e == ['eager'] * 6
	for n in p_e:
		if n in ('reg.data', 'reg_val.data'):
			moved = (p_e[n] - start[n]).abs().max().item()
			assert moved > 0, n
			# SGD is linear in the gradient: the two loops differ by the float-atomic noise of the sampling backward, not by flips
			assert (p_g[n] - p_e[n]).abs().max().item() < 1e-3 * moved, (n, (p_g[n] - p_e[n]).abs().max().item(), moved)
		elif not n.endswith('_val.data'):   # (the other val tables were filled by _fill_val_tables after `start` was taken)
			assert torch.equal(p_g[n], start[n]) and torch.equal(p_e[n], start[n]), n
		else:
			assert torch.equal(p_g[n], p_e[n]), n
	for epoch in range(3):
		for part, cnt in (('train_loss', 4), ('val_loss', 2)):
			a, b = tr_g.log[epoch][part], tr_e.log[epoch][part]
			assert set(a) == set(b) == {'Loss', 'Chamf'}
			for k in a:
				assert len(a[k]) == len(b[k]) == cnt
				np.testing.assert_allclose(a[k], b[k], rtol=1e-4, atol=1e-7)
	# the losses go down: the stage registers
	assert np.mean(tr_g.log[2]['train_loss']['Loss']) < np.mean(tr_g.log[0]['train_loss']['Loss'])


def test_render_watchdog_sees_the_renders_of_a_captured_step():
	"""Graph replay is the Trainer's default, and a captured render cannot hand its counters to a pinned slot per call (no host work inside
	a replay): it adds them to one static device slot instead, read back at the epoch boundary (functional_render._check_captured;
	ADVICE r3: graph epochs used to go unchecked).  A camera inside the mesh -- faces straddle the z-clip plane -- must be reported for a
	replayed step exactly as for an eager one, and a clean step must stay silent."""
	import warnings
	from find_amd import functional_render as FR
	from find_amd import optim
	from find_amd.cameras import look_at_view_transform
	from find_amd.graph import GraphedStep
	from find_amd.renderer import FootRenderer
	prev = FR.FLAG_POLICY
	FR.FLAG_POLICY = 'warn'
	try:
		FR.check_render_flags(wait=True)
		for dist, bad in ((0.3, False), (0.02, True)):
			mwl, opts, batch_of, _, _ = _setup(1002, 1002, capturable=True)   # (a model per capture: the first one's static loss keeps its graph alive)
			mwl.rdr = FootRenderer(image_size=64, device='cuda')
			R, T = look_at_view_transform(dist=np.full(2, dist), elev=np.array([0.0, 30.0]), azim=np.array([0.0, 40.0]), up=((1, 0, 0),))
			opt = optim.Adam(mwl.model.main_params, lr=1e-5, capturable=True)
			gs = GraphedStep(mwl, opts, [opt], warmup=1, sil=True, render_foot=True, views=(R.cuda(), T.cuda()))
			with warnings.catch_warnings(record=True) as wlist:
				warnings.simplefilter('always')
				for i in (0, 1, 2):
					gs(batch_of(i))
				FR.check_render_flags(wait=True)
			hits = [w for w in wlist if 'straddle the z-clip plane' in str(w.message)]
			assert bool(hits) == bad, (dist, [str(w.message)[:80] for w in wlist])
			if bad:
				assert any('replayed from a HIP graph' in str(w.message) for w in hits)
			# the check above must not be the last one -- `_watch` registers a device only while a stream CAPTURES, so a device taken
			# off the list at the first epoch boundary was never read again.  More replays, a second boundary: reported again (and clean stays clean).
			with warnings.catch_warnings(record=True) as wlist:
				warnings.simplefilter('always')
				FR.check_render_flags(wait=True)
				assert not [w for w in wlist if 'straddle' in str(w.message)]   # (the counters were reset by the first check)
				for i in (1, 2):
					gs(batch_of(i))
				FR.check_render_flags(wait=True)
			hits = [w for w in wlist if 'replayed from a HIP graph' in str(w.message)]
			assert bool(hits) == bad, (dist, 'second epoch boundary')
	finally:
		FR.FLAG_POLICY = prev
		try:
			FR.check_render_flags(wait=True)
		except RuntimeError:
			pass


def test_ragged_scans_share_bucketed_graphs():
	"""Real Foot3D scans are ragged (src/data/dataset.py:263-297; collate at :55-67): twelve scans of twelve different sizes through the
	Trainer's default graph mode must not mean twelve captures.  GraphedStep pads the batch's mesh to bucket sizes (faces with -1, which every
	kernel skips) -- the epoch replays at most 3 graphs and leaves the parameters where the eager loop on the UNPADDED scans leaves them."""
	from find_amd import optim, synthetic
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.trainer import Trainer
	sizes = [(26 + i, 30 + (i * 7) % 11) for i in range(12)]   # (rings, segments) of the lat-long GT scans: 782 .. 1300 vertices, twelve distinct sizes
	assert len({r * s for r, s in sizes}) == 12
	g = torch.Generator().manual_seed(3)
	scans = []
	for r, sg in sizes:
		v, f = synthetic.ellipsoid_mesh(r, sg)
		v = v * (1 + 0.1 * torch.rand(1, 3, generator=g)) + 0.002 * torch.randn(v.shape, generator=g)
		c = torch.rand(v.shape, generator=g).clamp(0.05, 0.95)
		scans.append((v.cuda(), f.cuda(), c.cuda()))

	def run(graph):
		mwl, opts, batch_of, _, _ = _setup(1002, 1002, capturable=True, seed=4)
		m = mwl.model
		feet, names, labels = synthetic.scan_labels(4)
		opt = optim.Adam(m.main_params, lr=5e-4, capturable=True)
		loader = []
		for i, (v, f, c) in enumerate(scans):
			j = i % 4
			loader.append(dict(mesh=Meshes(v[None], f[None], TexturesVertex(c[None])), idx=torch.tensor([j], device='cuda'), name=[names[j]],
							   shape=[feet[j]], tex=[feet[j]], pose=[names[j]], reg=[names[j]]))
		tr = Trainer([opt], mwl, loader, [], opts, latent_vectors_train=m.latent_vectors_train, latent_vectors_val=m.latent_vectors_val,
					 val_optim=opt, device='cuda', graph=graph)
		torch.manual_seed(11)   # the samplers draw from torch's device generator: same draws in both loops (the sizes do not enter them)
		msg = tr.train_epoch(0, model_kwargs=dict(opts.net_train_kwargs()))
		torch.cuda.synchronize()
		return tr, {n: p.detach().clone() for n, p in m.named_parameters()}, msg

	tr_g, p_g, msg = run('auto')
	tr_e, p_e, _ = run(False)
	assert tr_g.last_mode == 'graph' and tr_e.last_mode == 'eager'
	assert 1 <= tr_g.last_captures <= 3, msg
	assert 'graph(s) captured' in msg
	a, b = tr_g.log[0]['train_loss'], tr_e.log[0]['train_loss']
	assert len(a['Loss']) == len(b['Loss']) == 12
	# GraphedStep's warm-up before each capture advances the device generator (documented: the dry step draws), so the two loops do not see the
	# same samples: the losses agree as two draws of 5000 / 1000 surface samples do, and the parameters as Adam steps of lr do
	np.testing.assert_allclose(a['Loss'], b['Loss'], rtol=0.15)
	lr, n_steps = 5e-4, 12
	for n in p_e:
		assert (p_g[n] - p_e[n]).abs().max().item() <= 2 * lr * n_steps, n
	# the cache is bounded: a GraphedStep that may keep ONE graph serves the same epoch by re-capturing, never holding more than one
	from find_amd.graph import GraphedStep, bucket_size
	mwl, opts, batch_of, _, _ = _setup(1002, 1002, capturable=True, seed=4)
	# (a stream of the caller's: every capture of this GraphedStep then runs where the earlier ones' gradient-accumulation nodes live)
	gs = GraphedStep(mwl, opts, [optim.Adam(mwl.model.main_params, lr=5e-4, capturable=True)], warmup=1, max_graphs=1, stream=torch.cuda.Stream(),
					 **opts.net_train_kwargs())
	feet, names, labels = synthetic.scan_labels(4)
	small, large = scans[0], scans[11]
	assert bucket_size(small[0].shape[0]) != bucket_size(large[0].shape[0])
	for v, f, c in (small, large, small):
		gs(dict(mesh=Meshes(v[None], f[None], TexturesVertex(c[None])), idx=torch.tensor([0], device='cuda'), name=[names[0]], shape=[feet[0]], tex=[feet[0]],
				pose=[names[0]], reg=[names[0]]))
	torch.cuda.synchronize()
	assert gs.n_captures == 3 and len(gs._graphs) == 1


def test_bucketed_replay_equals_the_unpadded_step_with_fixed_draws():
	"""The padding itself changes no number: with the samplers' draws fixed (face indices chosen inside the real face range), a scan replayed from
	a graph captured on a DIFFERENT, larger scan of the same bucket gives the loss of the eager step on the unpadded scan."""
	from find_amd import optim, synthetic
	from find_amd.graph import GraphedStep, bucket_size
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	g = torch.Generator().manual_seed(5)
	scans = []
	for r, sg in ((28, 33), (30, 34)):   # 926 and 1022 vertices: both in the 1024 bucket
		v, f = synthetic.ellipsoid_mesh(r, sg)
		v = v * (1 + 0.1 * torch.rand(1, 3, generator=g))
		scans.append((v.cuda(), f.cuda(), torch.rand(v.shape, generator=g).clamp(0.05, 0.95).cuda()))
	assert bucket_size(scans[0][0].shape[0]) == bucket_size(scans[1][0].shape[0]) and scans[0][0].shape != scans[1][0].shape
	F_small = scans[0][1].shape[0]
	draws = [(torch.randint(0, F_small, (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, 2 * (1002 - 2), (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, F_small, (1, 1000), generator=g).cuda(), torch.rand(1, 1000, 2, generator=g).cuda())]
	feet, names, labels = synthetic.scan_labels(4)

	def batch(i):
		v, f, c = scans[i]
		return dict(mesh=Meshes(v[None], f[None], TexturesVertex(c[None])), idx=torch.tensor([2], device='cuda'), name=[names[2]], shape=[feet[2]], tex=[feet[2]],
					pose=[names[2]], reg=[names[2]])

	mwl, opts, _, _, _ = _setup(1002, 1002, capturable=True, seed=6)
	with FixedDraws(draws):
		b = batch(0)
		b.update(sample_latent_vectors(b, mwl.model.latent_vectors_train))
		want, want_terms = mwl(b, 0, opts, **opts.net_train_kwargs())
		want = want.item()
	mwl2, opts2, _, _, _ = _setup(1002, 1002, capturable=True, seed=6)
	gs = GraphedStep(mwl2, opts2, [optim.Adam(mwl2.model.main_params, lr=0.0, capturable=True)], warmup=1, **opts2.net_train_kwargs())
	with FixedDraws(draws):
		gs(batch(1))                 # capture on the LARGER scan of the bucket (lr = 0: the parameters stay put)
		got, got_terms = gs(batch(0))   # replay on the smaller one: its tail must be overwritten with padding
	torch.cuda.synchronize()
	assert gs.n_captures == 1
	assert abs(got.item() - want) < 1e-5 * max(1.0, abs(want)), (got.item(), want)
	for k in want_terms:
		assert abs(got_terms[k].item() - want_terms[k].item()) < 1e-5 * max(1.0, abs(want_terms[k].item())), k


def test_a_scan_that_fills_its_bucket_exactly_leaves_no_stale_faces_behind():
	"""ADVICE r4 (graph.py `_load`): smaller scan, then a scan whose face count EQUALS the bucket size, then a mid-sized one.  The exact-size scan
	is copied whole; round 4 did not record that the static tensors were full afterwards, so the mid-sized scan skipped the tail fill and
	the graph kept sampling (and rendering) the previous scan's last faces.  The static GT mesh after every load must be the scan padded
	with -1 faces / zero features, and the replayed loss must equal the eager step's on the unpadded scan (fixed draws)."""
	from find_amd import optim, synthetic
	from find_amd.graph import GraphedStep, bucket_size
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	g = torch.Generator().manual_seed(8)
	v, f = synthetic.ellipsoid_mesh(24, 32)     # 770 vertices, 1536 faces = a bucket size exactly
	assert f.shape[0] == 1536 == bucket_size(1536)
	v = (v * (1 + 0.1 * torch.rand(1, 3, generator=g))).cuda()
	c = torch.rand(v.shape, generator=g).clamp(0.05, 0.95).cuda()
	f = f[torch.randperm(f.shape[0], generator=g)].cuda()   # (so that a prefix of the face list is a mesh with holes all over, not a capless one)
	counts = [1400, 1536, 1500, 1536, 1300]
	feet, names, labels = synthetic.scan_labels(4)

	def batch(n):
		return dict(mesh=Meshes(v[None], f[None, :n].contiguous(), TexturesVertex(c[None])), idx=torch.tensor([1], device='cuda'), name=[names[1]], shape=[feet[1]],
					tex=[feet[1]], pose=[names[1]], reg=[names[1]])

	draws = [(torch.randint(0, 1300, (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, 2 * (1002 - 2), (1, 5000), generator=g).cuda(), torch.rand(1, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, 1300, (1, 1000), generator=g).cuda(), torch.rand(1, 1000, 2, generator=g).cuda())]
	mwl, opts, _, _, _ = _setup(1002, 1002, capturable=True, seed=6)
	gs = GraphedStep(mwl, opts, [optim.Adam(mwl.model.main_params, lr=0.0, capturable=True)], warmup=1, **opts.net_train_kwargs())
	with FixedDraws(draws):
		for n in counts:
			got, _ = gs(batch(n))
			torch.cuda.synchronize()
			st = next(iter(gs._graphs.values()))
			sf = st.batch['mesh'].faces_padded()
			sf = sf if sf.dim() == 2 else sf[0]
			assert sf.shape[0] == 1536
			assert torch.equal(sf[:n].long(), f[:n].long()), n
			assert bool((sf[n:] == -1).all()), f'{int((sf[n:] != -1).any(dim=-1).sum())} stale face(s) behind a scan of {n} faces'
	assert gs.n_captures == 1
	# the replay of the last (smallest) scan against the eager step on the unpadded scan
	mwl2, opts2, _, _, _ = _setup(1002, 1002, capturable=True, seed=6)
	with FixedDraws(draws):
		b = batch(counts[-1])
		b.update(sample_latent_vectors(b, mwl2.model.latent_vectors_train))
		want, _ = mwl2(b, 0, opts2, **opts2.net_train_kwargs())
	assert abs(got.item() - want.item()) < 1e-5 * max(1.0, abs(want.item())), (got.item(), want.item())
