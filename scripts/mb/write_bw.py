import torch, time
x = torch.empty(int(5.5e9)//4, dtype=torch.float32, device='cuda')
for _ in range(3): x.fill_(1.0)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): x.fill_(1.0)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("fill 5.5 GB: %.3f ms  %.2f TB/s" % (dt*1e3, 5.5e9/dt/1e12))
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print("copy 5.5 GB: %.3f ms  %.2f TB/s (r+w)" % (dt*1e3, 11e9/dt/1e12))
