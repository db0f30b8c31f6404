import sys, os, time, threading, json
sys.path.insert(0, "/root/repo/image-cases-studies_amd"); sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from lib import _native
M=N=4096; MK=15
mode = sys.argv[1] if len(sys.argv)>1 else "nonblind"
nf = int(sys.argv[2]) if len(sys.argv)>2 else 2
blind = mode=="blind"
image,u0,pt,pu = bench.synth_frame(M,N,MK,0)
jobs=[]
for i in range(nf):
    ctx=_native.Context(0)          # own stream per job
    job=_native.RLJob(M,N,MK,ctx); job.upload(image,u0,pu if blind else pt); jobs.append((ctx,job))
win=(8,247,8,247)
def run(job,n): 
    p=job.params(*win,1e9,n//5,1e-3,10000.0,blind,0,3,stop_test=2,profile=0); return job.run(p)
for ctx,job in jobs: run(job,5)
for ctx,job in jobs: ctx.synchronize()
steps=40
t0=time.perf_counter()
th=[threading.Thread(target=run,args=(job,steps)) for ctx,job in jobs]
[t.start() for t in th]; [t.join() for t in th]
for ctx,job in jobs: ctx.synchronize()
dt=time.perf_counter()-t0
print(mode, "frames/gpu", nf, "ms per inner iteration per frame-set %.4f"%(dt*1e3/steps), "MPix/s/iter %.1f"%(nf*M*N*steps/dt/1e6), "frac of 8TB/s %.4f"%((144 if not blind else 204)*nf*M*N*steps/dt/8e12))
