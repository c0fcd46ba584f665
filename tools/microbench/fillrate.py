import torch
for gb in (0.5, 2, 8):
    n = int(gb * (1 << 30) / 8)
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(2): x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): x.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    print("fill %.1f GB: %.2f TB/s" % (gb, n * 8 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12))
    del x
