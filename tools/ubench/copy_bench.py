import torch
dev="cuda:0"
for rows,cols in ((81920,64),(81920,192),(819200,64)):
    x=torch.randn(rows,cols,device=dev); y=torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/50*1e3
    print(f"copy {rows}x{cols}: {us:.1f} us, {2*rows*cols*4/us/1e6:.2f} TB/s (read+write)")
