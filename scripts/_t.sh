python bench.py --workload rektnet --no-cpu-baseline --no-fp32 --steps 12 --warmup 4 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['workloads']['rektnet']
print(round(l['value'],1), round(l['ms_per_step'],3)); print(r['kernel_ms_per_step'])
for k in r['roofline_kernels'][:40]: print('   ', k['kernel'][:86], k['launches'], round(k['avg_us'],1), round(k['total_ms'],3), round(k['achieved'],1), k['unit'])
"
