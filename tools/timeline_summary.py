#!/usr/bin/env python3
"""What the two-stream denoiser pipeline's time is made of: a rocprofv3 --kernel-trace CSV of
    python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 --gate-ms 60
(the timed aipt_frames call queued behind a spin, so that the GPU meets it fully enqueued: under the profiler the host needs
~0.85 ms per frame to enqueue and the streams would run host-bound) -> per frame: the time with a given set of kernel classes in
flight (big = conv3x3_f16x3r, small = the LDS-tiled / f32 kernels, quad = the output layer's passes), and the summed durations.
    python tools/timeline_summary.py k_kernel_trace.csv [first_event count]     (optionally lists events of the timed call)"""
import csv, sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
def short(n):
    if 'spin_kernel' in n: return 'spin_kernel'               # torch.cuda._sleep: at::cuda::(anonymous namespace)::spin_kernel(long)
    n=n.replace('void aipt::','').replace('aipt::','')
    return n.split('(')[0].replace(' ','')
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),r['Queue_Id']) for r in rows]
ev.sort()
# the timed call: everything queued behind the gate's spin kernel (torch.cuda._sleep) up to its 20th denoised frame
spin=[i for i,e in enumerate(ev) if e[2]=='spin_kernel']
assert spin, 'no spin kernel in the trace: run bench.py with --gate-ms'
gate_end=ev[spin[-1]][1]
call=[e for e in ev if e[0]>=gate_end]
t0=call[0][0]
seg=[]; nq=0; last_trace_end=t0
for e in call:
    if e[2].startswith('trace_'):
        if nq==0: last_trace_end=max(last_trace_end,e[1])
        continue
    if 'fillBuffer' in e[2] and not seg: continue
    if nq<20:
        seg.append(e)
        if e[2]=='conv3x3_quad<3,3,false>': nq+=1
tA=seg[0][0]; tB=max(e[1] for e in seg)
print('trace ms/frame', (last_trace_end-t0)/1e6/20, 'denoise ms/frame', (tB-tA)/1e6/20, 'total', (tB-t0)/1e6/20)
def cls(n):
    if n.startswith('conv3x3_f16x3r'): return 'big'
    if n.startswith('conv3x3_quad'): return 'quad'
    if n.startswith('conv3x3'): return 'small'
    return 'other'
tot=defaultdict(float)
for e in seg: tot[cls(e[2])]+=(e[1]-e[0])/1e3
print({k:round(v/20,1) for k,v in tot.items()}, 'us per frame (sum of durations)')
pts=[]
for e in seg: pts+=[(e[0],1,cls(e[2])),(e[1],-1,cls(e[2]))]
pts.sort()
act=defaultdict(int); last_t=pts[0][0]; acc=defaultdict(float)
for t,d,c in pts:
    key=tuple(sorted((k,v) for k,v in act.items() if v))
    acc[key]+=t-last_t
    act[c]+=d; last_t=t
for k,v in sorted(acc.items(), key=lambda kv:-kv[1])[:12]: print(f"{v/1e3/20:8.1f} us/frame  active: {k}")
if len(sys.argv)>2:
    s=int(sys.argv[2])
    for e in seg[s:s+int(sys.argv[3])]: print(f"{(e[0]-tA)/1e3:9.1f} -> {(e[1]-tA)/1e3:9.1f} {(e[1]-e[0])/1e3:7.1f} q{e[3]} {e[2][:44]:44s}")
