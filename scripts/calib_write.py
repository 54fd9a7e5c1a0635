import torch
x = torch.empty(2**28, dtype=torch.float32, device="cuda")   # 1 GiB
for _ in range(3):
    x.fill_(1.0)
torch.cuda.synchronize()
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
