"""Debug aid: the two scenes of tests/test_gpu_render.py::test_lists_longer_than_the_binning_buffer_and_the_pole_of_a_scan_vs_oracle, mask error by pixel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import render_ref, camera_ref
from find_amd import synthetic, _lib, functional_render as FR
size = 64
rp = render_ref.default_params(size)
for n_verts, elev, azim in ((50002, 25.0, -40.0), (10002, 0.0, 0.0)):
	v, f = synthetic.template(n_verts)
	verts = v[None].clone()
	R, T = camera_ref.look_at_view_transform(dist=np.full(1, 0.3), elev=np.array([elev]), azim=np.array([azim]), up=((1, 0, 0),))
	R, T = torch.from_numpy(R), torch.from_numpy(T)
	vproj = render_ref.project(rp, verts.numpy(), R.numpy(), T.numpy())
	p2f, z, _, d = render_ref.rasterize(vproj, f.numpy(), 1, size, size, 400, rp.sil_blur_radius)
	cnt = (p2f[0] >= 0).sum(-1)
	ref = render_ref.render(verts.numpy(), f.numpy(), None, R.numpy(), T.numpy(), image_size=size, want_image=False)['mask'][0, 0]
	for bits in (0, 8, 16, 2):
		_lib.set_tuning('raster_ablate', bits)
		mask, _, _, _ = FR.render(verts.cuda(), None, f.cuda(), R.cuda(), T.cuda(), FR.make_params(size), want_image=False)
		_lib.set_tuning('raster_ablate', 0)
		err = np.abs(mask[0, 0].cpu().numpy() - ref)
		bad = np.argwhere(err > 1e-4)
		print(f'{n_verts} verts ablate {bits}: max err {err.max():.3e}, {len(bad)} pixels > 1e-4; candidates at those: {[int(cnt[y, x]) for y, x in bad[:12]]}; errs {[float(f"{err[y, x]:.2e}") for y, x in bad[:12]]}; px {[(int(y), int(x)) for y, x in bad[:12]]}')
