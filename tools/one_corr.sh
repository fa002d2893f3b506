#!/bin/bash
# kernel trace + HIP API trace of ONE chunk per GPU in the same run: for every kernel, when was its launch call made
# relative to the end of the previous kernel on its queue (host-late or device-late?) -> gpurun_out/one_corr.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
rm -rf /tmp/onec
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/onec -o one -- python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/onec_bench.json 2> gpurun_out/onec.err || exit 1
python3 - $(find /tmp/onec -name "one_kernel_trace.csv") $(find /tmp/onec -name "one_hip_api_trace.csv") > gpurun_out/one_corr.txt <<'PY'
import csv,sys
ks=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Kernel_Name'] for x in ('at::','elementwise','vectorized','rocprim','hipcub','fillBuffer'))]
api={}
for r in csv.DictReader(open(sys.argv[2])):
    if r['Function']=='hipLaunchKernel': api[r['Correlation_Id']]=(int(r['Start_Timestamp']),int(r['End_Timestamp']))
ks.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(ks); ks=ks[n//2:n//2+260]
t0=int(ks[0]['Start_Timestamp'])
lastend={}
byc={r['Correlation_Id']:(r['Queue_Id'],r['Kernel_Name'].split('(')[0][-45:],r['Grid_Size_X'] if 'Grid_Size_X' in r else '') for r in csv.DictReader(open(sys.argv[1]))}
# host calls longer than 30 us in the window
t1=int(ks[-1]['End_Timestamp'])
seq=[(int(r['Start_Timestamp']),r['Function'],int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(sys.argv[2])) if int(r['Start_Timestamp'])>=t0-200000 and int(r['End_Timestamp'])<=t1 and r['Function'] in ('hipLaunchKernel','hipEventRecord','hipStreamWaitEvent')]
seq.sort()
pos=[i for i,x in enumerate(seq) if x[2]>30000]
print('ordinals of the long calls among launches/records/waits:',pos)
print('spacing:',[b-a for a,b in zip(pos,pos[1:])])
launch_pos=[i for i,x in enumerate([y for y in seq if y[1]=='hipLaunchKernel']) if x[2]>30000]
print('spacing counted in launches only:',[b-a for a,b in zip(launch_pos,launch_pos[1:])])
print('long host calls in the window:')
for r in csv.DictReader(open(sys.argv[2])):
    s_=int(r['Start_Timestamp']); e_=int(r['End_Timestamp'])
    if s_>=t0-200000 and e_<=t1 and e_-s_>30000: print('   %-24s at %9.1f for %7.1f us -> %s'%(r['Function'],(s_-t0)/1e3,(e_-s_)/1e3,byc.get(r['Correlation_Id'],'?')))
print('kernel start(us) dur | launch call made at (us, relative to the same clock) | gap on its queue | call-to-start')
for r in ks:
    q=r['Queue_Id']; s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    a=api.get(r['Correlation_Id'])
    nm=r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
    gap=(s-lastend[q])/1e3 if q in lastend else 0
    print('q%s %-28s %9.1f %7.1f | call %9.1f | gap %7.1f | call->start %8.1f'%(q,nm,(s-t0)/1e3,(e-s)/1e3,((a[0]-t0)/1e3 if a else float('nan')),gap,((s-a[1])/1e3 if a else float('nan'))))
    lastend[q]=e
PY
