"""Cost of the fused data-gradient epilogues (ReLU mask, residual add, BN-backward reductions) per layer shape:
time of the NT kernel alone, from the library's HIP-event profiler.  usage: python tools/epilogue_probe.py"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from instaorder_amd import _lib, engine
L=_lib.lib(); P=lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ST=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t_nt(fn, reps=5):
    fn(); torch.cuda.synchronize()
    engine.prof_begin()
    for _ in range(reps): fn()
    pr=engine.prof_end()
    tot=sum(v["total_ms"] for k,v in pr.items() if k.startswith("conv_nt"))
    return tot/reps
N=512
for (H,Cin,Cout,k) in [(16,1024,256,1),(16,256,1024,1),(16,256,256,3),(64,256,64,1),(64,64,256,1)]:
    pad=k//2
    dy=torch.randn(N,H,H,Cout,device="cuda"); wt=torch.randn(Cin,k*k,Cout,device="cuda")*0.05
    dx=torch.empty(N,H,H,Cin,device="cuda"); add=torch.randn(N,H,H,Cin,device="cuda"); mask=torch.randn(N,H,H,Cin,device="cuda")
    y=torch.randn(N,H,H,Cin,device="cuda")
    fl=2.0*N*H*H*Cin*Cout*k*k
    a=t_nt(lambda: L.io_conv2d_dgrad(P(dy),P(wt),P(dx),None,None,N,H,H,Cin,Cout,k,k,1,pad,ST()))
    b=t_nt(lambda: L.io_conv2d_dgrad(P(dy),P(wt),P(dx),None,P(mask),N,H,H,Cin,Cout,k,k,1,pad,ST()))
    c=t_nt(lambda: L.io_conv2d_dgrad(P(dy),P(wt),P(dx),P(add),P(mask),N,H,H,Cin,Cout,k,k,1,pad,ST()))
    G=2; M=N*H*H
    tiles=M//128; per=(tiles+tiles//64+G+2)*Cin
    ws=torch.empty(2*per+2*G*Cin,device="cuda")
    g_=torch.ones(Cin,device="cuda"); mean=torch.zeros(G*Cin,device="cuda"); rstd=torch.ones(G*Cin,device="cuda")
    dg=torch.empty(Cin,device="cuda"); db=torch.empty(Cin,device="cuda"); dyb=torch.empty_like(dx)
    d=t_nt(lambda: L.io_conv2d_dgrad_bnbwd(P(dy),P(wt),P(dx),N,H,H,Cin,Cout,k,k,pad,P(y),G,P(g_),P(mean),P(rstd),P(rstd),P(mean),P(dg),P(db),P(dyb),P(ws),ws.numel(),ST()))
    print("H%d %d->%d k%d | plain %.3f ms %.1f TF | +mask %.3f | +add+mask %.3f | +bnbwd(y,mask from y) %.3f"%(H,Cout,Cin,k,a,fl/a/1e9,b,c,d))
