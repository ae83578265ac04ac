"""Streaming bandwidth references on this box (ATen fill / copy; tuning only): what a write-heavy pass can hope for."""
import torch
dev = torch.device('cuda')
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (59, 118, 236, 472, 944):
    a = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    b = torch.empty_like(a)
    f = t(lambda: a.fill_(1.0))
    c = t(lambda: b.copy_(a))
    print(f'{mb} MB: fill {f*1e3:.1f} us = {mb*1.048576/f/1e3:.2f} TB/s written; copy {c*1e3:.1f} us = {2*mb*1.048576/c/1e3:.2f} TB/s read+written')
