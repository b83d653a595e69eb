import torch, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(True),torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
N=1<<30   # bytes*... elements of fp32 = 4 GiB
x=torch.empty(N//2, dtype=torch.float32, device='cuda')   # 2 GiB
y=torch.empty_like(x)
gb=x.numel()*4/1e9
print('fill (write only)    %.2f TB/s'%(gb/t(lambda: x.fill_(1.0))/1e-3/1e3))
print('copy (read+write)    %.2f TB/s total'%(2*gb/t(lambda: y.copy_(x))/1e-3/1e3))
print('sum  (read only)     %.2f TB/s'%(gb/t(lambda: x.sum())/1e-3/1e3))
xb=torch.empty(N//2, dtype=torch.bfloat16, device='cuda')
print('fill bf16            %.2f TB/s'%(xb.numel()*2/1e9/t(lambda: xb.fill_(1.0))/1e-3/1e3))
print('add_ in place (r+w)  %.2f TB/s total'%(2*gb/t(lambda: x.add_(1.0))/1e-3/1e3))
