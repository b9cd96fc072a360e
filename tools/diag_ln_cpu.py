"""CPU-only diagnostic: intermediates of the oracle LN chain for one row, in hex (machine comparison)."""
import torch, struct, platform
def hx(t): return struct.pack('>f', float(t)).hex()
torch.manual_seed(0)
s1 = torch.tensor(0.0034242335241287947)   # any non-PoT scale
C = 384
S1 = torch.tensor([-525.0]); S2 = torch.tensor([2065499.0])
a = s1 / C
var = C * S2 - S1 * S1
sq = torch.sqrt(var)
std = a * sq
rs = s1 / std
print(platform.processor(), torch.__version__, torch.backends.cpu.get_cpu_capability())
print('s1', hx(s1), 's1/C', hx(a), 'var', hx(var), 'sqrt', hx(sq), 'std', hx(std), 'rs', hx(rs))
# the same through expanded (vectorised) tensors
S1v = S1.expand(64).contiguous(); S2v = S2.expand(64).contiguous()
stdv = (s1 / C) * torch.sqrt(C * S2v - S1v * S1v)
print('vec std', hx(stdv[0]), hx(stdv[63]), 'vec rs', hx((s1 / stdv)[0]), hx((s1/stdv)[63]))
x = torch.rand(1000) + 0.5
y = torch.rand(1000) + 0.5
import hashlib
for name, v in (('div', x / y), ('sqrt', torch.sqrt(x)), ('mul', x * y), ('sdiv', s1 / x), ('divs', x / C)):
    print(name, hashlib.sha1(v.numpy().tobytes()).hexdigest()[:12])
v = (x.reshape(-1, 1) * y.reshape(1, -1))
print('bcast mul', hashlib.sha1(v.numpy().tobytes()).hexdigest()[:12])
