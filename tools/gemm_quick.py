import sys, json, os, torch
sys.path.insert(0, ".")
from specforge_amd import ops
dev="cuda"; torch.manual_seed(0)
def t(fn, w=2, n=6):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
tot_f=0; tot_ms=0; out=[]
for (M,N,K) in [(16384,28672,4096),(16384,4096,14336),(16384,32000,4096),(16384,6144,8192),(32000,4096,114688),(28672,4096,114688),(4096,14336,114688)]:
    a=torch.randn(M,K,device=dev).to(torch.bfloat16); b=torch.randn(N,K,device=dev).to(torch.bfloat16); c=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    ms=t(lambda: ops.gemm_nt(a,b,c)); tot_f+=2.0*M*N*K; tot_ms+=ms; out.append(round(2.0*M*N*K/ms/1e9))
    del a,b,c
print(os.environ.get("SF_GEMM_GM"), os.environ.get("SF_GEMM_FLAGS"), "avg", round(tot_f/tot_ms/1e9), out)
