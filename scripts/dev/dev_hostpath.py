import sys, time; sys.path.insert(0,".")
import numpy as np, torch
from libwave_amd import capi, synth
ref,tgt,T=synth.pair(1000000,seed=42)
ctx=capi.Context(0)
def step(a,b):
    ctx.set_source(a); ctx.set_target(b)
    return ctx.icp_align(max_corr=3.0,force_iterations=50,nn_method=capi.WM_NN_GRID,carry_state=0)
for name,(a,b) in {"host numpy (pageable)":(ref,tgt),"device":(torch.from_numpy(ref).cuda(),torch.from_numpy(tgt).cuda())}.items():
    for _ in range(3): step(a,b)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): step(a,b)
    torch.cuda.synchronize(); print(name,"%.2f ms/registration"%((time.perf_counter()-t0)/10*1e3))
    t0=time.perf_counter()
    for _ in range(10): ctx.set_source(a)
    torch.cuda.synchronize(); print("  set_source alone %.2f ms"%((time.perf_counter()-t0)/10*1e3))
