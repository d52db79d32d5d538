import sys, time, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); [sys.path.insert(0,os.path.join(R,p)) for p in ('.','tests','gp-quadrature_amd')]
import torch
from test_gpu_cg_hermitian import _system
from efgp_hip import ToeplitzOp, cg_solve
for mtot in (23, 17, 13):
    v,T,ws,b=_system(mtot,3)
    op=ToeplitzOp(v.cuda())
    diag=(700.0*ws.abs().pow(2).real+0.25).cuda()
    for env in (None,"1"):
        if env: os.environ["EFGP_NO_CG48"]=env
        else: os.environ.pop("EFGP_NO_CG48",None)
        for rep in range(3):
            torch.cuda.synchronize(); t=time.perf_counter()
            x,it,_=cg_solve(op,ws.cuda(),0.25,0,b.cuda(),torch.zeros_like(b).cuda(),1e-300,max_iter=2000,early_stop=False,diag=diag,batched=False,hermitian=True)
            torch.cuda.synchronize(); dt=time.perf_counter()-t
        print(mtot, "64" if env else "48", it, f"{1e6*dt/it:.3f} us/iter")
