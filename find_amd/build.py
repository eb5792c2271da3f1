"""Build the two libraries (gfx950) in-tree with hipcc: libfind_hip.so, the product, and libfind_hip_diag.so, the same sources with -DFIND_DIAG
(fault reproducers, superseded A/B kernels, timers, wrong-result ablation bits: include/find_hip_diag.h) for tools/.
`python -m find_amd.build [-j N] [--force] [--no-diag]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libfind_hip.so')
LIB_DIAG = os.path.join(LIBDIR, 'libfind_hip_diag.so')
OBJDIR = os.path.join(LIBDIR, 'obj')
OBJDIR_DIAG = os.path.join(LIBDIR, 'obj_diag')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC] + os.environ.get('FIND_EXTRA_HIPCC_FLAGS', '').split()


def _sources():
	return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
	hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
	hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith('.h')]
	return max(os.path.getmtime(h) for h in hdrs)


LAST_COMPILED = []   # objects the last build() call compiled (what a caller may report: __graft_entry__.build)


def build(force=False, jobs=4, verbose=True, diag=True):
	"""Compile every csrc/*.hip for gfx950 and link libfind_hip.so (and, with diag, libfind_hip_diag.so).  Incremental on mtimes."""
	del LAST_COMPILED[:]
	lib = _build_one(LIB, OBJDIR, [], force, jobs, verbose)
	if diag:
		_build_one(LIB_DIAG, OBJDIR_DIAG, ['-DFIND_DIAG'], force, jobs, verbose)
	return lib


def _build_one(LIB, OBJDIR, extra, force, jobs, verbose):
	os.makedirs(OBJDIR, exist_ok=True)
	hdr_m = _deps_mtime()
	todo, objs = [], []
	for src in _sources():
		obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
		objs.append(obj)
		if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
			todo.append((src, obj))

	def cc(job):
		src, obj = job
		cmd = [HIPCC] + FLAGS + extra + ['-c', src, '-o', obj]
		if verbose:
			print('[find_amd.build]', ' '.join(cmd), flush=True)
		r = subprocess.run(cmd, capture_output=True, text=True)
		if r.returncode != 0:
			raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')

	if todo:
		with ThreadPoolExecutor(max_workers=jobs) as ex:
			list(ex.map(cc, todo))
		LAST_COMPILED.extend(os.path.relpath(obj, ROOT) for _, obj in todo)
	if todo or not os.path.exists(LIB):
		cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
		if verbose:
			print('[find_amd.build]', ' '.join(cmd), flush=True)
		r = subprocess.run(cmd, capture_output=True, text=True)
		if r.returncode != 0:
			raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
	return LIB


if __name__ == '__main__':
	j = 4
	if '-j' in sys.argv:
		j = int(sys.argv[sys.argv.index('-j') + 1])
	print(build(force='--force' in sys.argv, jobs=j, diag='--no-diag' not in sys.argv))
