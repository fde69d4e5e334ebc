"""Known-traffic kernel for calibrating the FETCH_SIZE / WRITE_SIZE counters: copies a 256 MiB fp32 tensor 5 times
(expected per launch: 268 435 456 bytes read, 268 435 456 bytes written).  Run under rocprofv3 --pmc."""
import torch
x = torch.randn(64 * 1024 * 1024, device='cuda:0')
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
print('done')
