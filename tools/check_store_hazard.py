#!/usr/bin/env python3
"""Lint the gfx950 code objects of a built library for the store-data hazard LLVM does not cover (tools/store_hazard_probe.hip):
a buffer store of more than 64 bits WITH an SGPR offset whose data registers the very next instruction overwrites with a VALU result loses
the first dword in ~0.1 % of the stores on an MI355X.  LLVM's hazard recognizer inserts the wait state only for the form without an SGPR
offset.  Usage: python tools/check_store_hazard.py [lib.so ...]; exit code 1 if a hazard is found.  (Also imported by tests/test_host_api.py.)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
WIDE = re.compile(r'^\s*(buffer_store_dwordx[34]|buffer_store_format_xyzw?|tbuffer_store_format_xyzw?)\s+([va])\[(\d+):(\d+)\],\s*([^,]+),\s*([^,]+),\s*(\S+)')


def _regs(tok):
	"""(file, numbers) of a vector register operand: 'v' = VGPRs, 'a' = accumulation registers (a store may read its data from either)."""
	m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
	if m:
		return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
	m = re.match(r'([va])(\d+)$', tok)
	return {(m.group(1), int(m.group(2)))} if m else set()


def _writes(line):
	"""Vector registers a VALU / matrix instruction writes (first operand; v_cmp* / v_readlane etc. write none)."""
	t = line.split('//')[0].strip()
	if not t.startswith('v_') or t.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
		return set()
	ops = re.split(r'[ ,]+', t)
	return _regs(ops[1]) if len(ops) > 1 else set()


def code_objects(lib, tmp):
	dst = os.path.join(tmp, os.path.basename(lib))
	shutil.copy(lib, dst)
	subprocess.run([OBJDUMP, '--offloading', dst], cwd=tmp, capture_output=True, check=True)
	return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if 'amdgcn' in f)


def scan(ins, out, stats):
	"""ins: [(kernel, instruction text)] in program order.  Appends (kernel, store, next instruction) for every wide store with an SGPR
	offset whose data registers the very next instruction of the same kernel writes."""
	for i, (k, l) in enumerate(ins):
		m = WIDE.match(l)
		if not m:
			continue
		stats['wide_stores'] += 1
		soff = m.group(7)
		if not soff.startswith('s') and soff not in ('m0',):
			continue   # immediate / "off" offset: the compiler's own hazard handling applies
		stats['sgpr_offset'] += 1
		data = {(m.group(2), r) for r in range(int(m.group(3)), int(m.group(4)) + 1)}
		if i + 1 < len(ins) and ins[i + 1][0] == k and (_writes(ins[i + 1][1]) & data):
			out.append((k, l.strip(), ins[i + 1][1].strip()))


def hazards(lib):
	"""[(kernel, store line, next line)] for every unprotected wide store in the library's gfx950 code objects."""
	out = []
	stats = dict(wide_stores=0, sgpr_offset=0)
	with tempfile.TemporaryDirectory() as tmp:
		for co in code_objects(lib, tmp):
			txt = subprocess.run([OBJDUMP, '-d', co], capture_output=True, text=True, check=True).stdout.splitlines()
			kernel = '?'
			ins = []
			for l in txt:
				m = re.match(r'^[0-9a-f]+ <(.+)>:', l)
				if m:
					kernel = m.group(1)
					continue
				if l.startswith('\t') or l.startswith('  '):
					ins.append((kernel, l.split('//')[0].rstrip()))
			scan(ins, out, stats)
	return out, stats


if __name__ == '__main__':
	libs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'find_amd', 'lib', 'libfind_hip.so')]
	rc = 0
	for lib in libs:
		hz, st = hazards(lib)
		print(f'{lib}: {st["wide_stores"]} stores of more than 64 bits, {st["sgpr_offset"]} with an SGPR offset, {len(hz)} followed at once by a VALU write of their data')
		for k, a, b in hz:
			print(f'   {k}\n      {a}\n      {b}')
		rc |= bool(hz)
	sys.exit(rc)
