import torch, time
nbytes = 1 << 30
src_t = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
dst_t = torch.empty_like(src_t)
dst_t.copy_(src_t)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    dst_t.copy_(src_t)
ev1.record()
torch.cuda.synchronize()
print("copy GB/s", 2.0 * nbytes * 5 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9)
