"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV: every launch between two marker kernels (the optimiser's) with its start
offset, duration, queue, grid / workgroup and LDS -- what runs beside what, and which kernels form the critical path.
python tools/step_timeline.py <kernel_trace.csv> <marker> [steps_from_end=6]"""
import csv
import re
import sys

path, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
k = len(marks) - back
seg = rows[marks[k] + 1:marks[k + 1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
end = 0
for r in seg:
	full = r['Kernel_Name']
	n = re.sub(r'\(.*', '', full).replace('find::', '').replace('void ', '')[:44]
	if 'at::native' in full:   # torch's own: the functor says which op it is
		f = re.findall(r'(\w*Functor\w*|\w+_kernel_cuda|\w*[Cc]opy\w*|distribution\w*|reduce_kernel|multi_tensor_apply\w*|CatArray\w*|index\w*|\w*scan\w*)', full)
		n = ('torch:' + '/'.join(dict.fromkeys(f)))[:44] if f else n
	s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
	gap = f'idle {(s - end) / 1e3:6.1f}' if s > end else ' ' * 11
	end = max(end, e)
	print(f"{s / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r['Queue_Id']:>2} {n:44s} {int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])):6d} x {int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']):>4} lds {r['LDS_Block_Size']:>6} regs {r['VGPR_Count']}+{r['Accum_VGPR_Count']}  {gap}")
