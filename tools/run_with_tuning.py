"""python tools/run_with_tuning.py key=value [key=value ...] -- <pytest args>: set find_ctx knobs (_lib.set_tuning), then run pytest in-process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from find_amd import _lib

sep = sys.argv.index('--')
for kv in sys.argv[1:sep]:
	k, v = kv.split('=')
	_lib.set_tuning(k, int(v))
sys.exit(pytest.main(sys.argv[sep + 1:]))
