import torch, time
n = 1 << 28  # 2 GiB of int64
for pinned in (False, True):
    h = torch.empty(n, dtype=torch.int64, pin_memory=pinned)
    h.fill_(1)
    d = torch.empty(n, dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pinned" if pinned else "pageable", "H2D GB/s", n * 8 / dt / 1e9)
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pinned" if pinned else "pageable", "D2H GB/s", n * 8 / dt / 1e9)
