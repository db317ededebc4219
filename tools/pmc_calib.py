"""Known-byte-count launches for calibrating rocprofv3's WRITE_SIZE / FETCH_SIZE on this box:
three fills of a 4 GiB tensor (4 GiB written each) and three copies (4 GiB read + 4 GiB written)."""
import torch
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
y = torch.ones(1 << 30, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    x.fill_(2.0)
for _ in range(3):
    x.copy_(y)
torch.cuda.synchronize()
