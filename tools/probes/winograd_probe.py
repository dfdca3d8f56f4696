"""Feasibility probe: time the 16 batched GEMMs of a Winograd F(2x2,3x3) conv4 layer (4 x 38 x 67, 256 -> 256) against
MIOpen's direct NHWC convolution of the same layer."""
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
N, C, H, W = 4, 256, 38, 67
tiles = N * ((H + 1) // 2) * ((W + 1) // 2)
V = torch.randn(16, tiles, C, device=dev); U = torch.randn(16, C, C, device=dev); M = torch.empty(16, tiles, C, device=dev)
print("tiles", tiles, "bmm us", t(lambda: torch.bmm(V, U, out=M)))
V2 = torch.randn(tiles, 16 * C, device=dev)
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
print("miopen conv us", t(lambda: F.conv2d(x, w, None, 1, 1)))
for (c, hh, ww, d) in [(512, 38, 67, 2), (128, 75, 134, 1), (64, 150, 267, 1)]:
    x = torch.randn(N, c, hh, ww, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(c, c, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    tl = N * ((hh + 1) // 2) * ((ww + 1) // 2)
    V = torch.randn(16, tl, c, device=dev); U = torch.randn(16, c, c, device=dev); M = torch.empty(16, tl, c, device=dev)
    print(c, hh, ww, "dil", d, "miopen us", t(lambda: F.conv2d(x, w, None, 1, d, d)), "bmm us", t(lambda: torch.bmm(V, U, out=M)))
