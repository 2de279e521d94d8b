"""Dev: conv_raw3 on structured inputs (ones / one-hot) to see WHAT is wrong, not only where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
import torch, torch.nn.functional as F
from phasegen import ops
torch.set_printoptions(linewidth=220, precision=3, sci_mode=False)
tr, Cin, Cout, k, s, p, Lin, B = (False, 16, 160, 8, 1, 2, 65, 3)
def run(x, w, tag):
    want = F.conv1d(x, w, stride=s, padding=p)
    y = torch.full(want.shape, float("nan"), device="cuda")
    ops.conv_fwd(x.cuda(), w.cuda(), y, s, p, transposed=tr, schedule=1)
    y = y.cpu()
    print(tag, "max err", float((y - want).abs().max()))
    print(" got  b0 row0 :", y[0, 0, :12].tolist()); print(" want b0 row0 :", want[0, 0, :12].tolist())
    print(" got  b0 row40:", y[0, 40, :12].tolist()); print(" want b0 row40:", want[0, 40, :12].tolist())
    return y, want
x = torch.ones(B, Cin, Lin); w = torch.ones(Cout, Cin, k)
run(x, w, "ones x ones")
# weights that encode the tap index: y = sum over valid taps of tap index * Cin
w = torch.arange(k, dtype=torch.float32).view(1, 1, k).expand(Cout, Cin, k).contiguous()
run(x, w, "w = tap index")
# weights that encode the channel index
w = torch.arange(Cin, dtype=torch.float32).view(1, Cin, 1).expand(Cout, Cin, k).contiguous()
run(x, w, "w = channel index")
# weights that encode the row index
w = torch.arange(Cout, dtype=torch.float32).view(Cout, 1, 1).expand(Cout, Cin, k).contiguous()
run(x, w, "w = row index")
# x encodes the position, single channel active
x = torch.zeros(B, Cin, Lin); x[:, 0, :] = torch.arange(Lin, dtype=torch.float32)
w = torch.zeros(Cout, Cin, k); w[:, 0, 0] = 1.0
run(x, w, "x = position on channel 0, tap 0 only")
w = torch.zeros(Cout, Cin, k); w[:, 0, 3] = 1.0
run(x, w, "x = position on channel 0, tap 3 only")
x = torch.zeros(B, Cin, Lin); x[:, 5, :] = torch.arange(Lin, dtype=torch.float32)
w = torch.zeros(Cout, Cin, k); w[:, 5, 3] = 1.0
run(x, w, "x = position on channel 5, tap 3 only")
