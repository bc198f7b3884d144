"""Precision gate of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct fp32 conv, CPU only.

Every variant is evaluated in fp32 exactly as a kernel would (transforms in fp32, the 36 / 16 element-wise products accumulated
over Cin in fp32) and compared with an fp64 direct convolution of the same fp32 operands.  Layers: the K-heavy 3x3 layers F(4x4)
would serve (Cin >= 128, maps <= 64^2), operands as in training (activations ~ LeakyReLU(N(0,1)) after PixelNorm scale, weights
N(0,1) x sqrt(2 / fan_in)), and the same layers with the heavy-tailed operands of a backward pass (gradients: N(0,1) x lognormal).
Prints rel-L2 and max-norm errors per layer and the ratio to the direct fp32 conv."""
import torch, math, sys
torch.manual_seed(0)
F = torch.nn.functional

def mats(m):
    if m == 2:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:       # Lavin & Gray F(4x4,3x3), interpolation points 0, +-1, +-2, inf
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
        G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    return [torch.tensor(a, dtype=torch.float64) for a in (BT, G, AT)]

def wino(x, w, m, dt=torch.float32):
    """x [N,C,H,W], w [K,C,3,3], pad 1; H, W multiples of m.  All arithmetic in dt; sums over C in dt (chunks of 8 like the kernels)."""
    BT, G, AT = [a.to(dt) for a in mats(m)]
    N, C, H, W = x.shape
    K = w.shape[0]
    a = m + 2
    xp = F.pad(x.to(dt), (1, 1, 1, 1))
    d = xp.unfold(2, a, m).unfold(3, a, m)                        # [N,C,th,tw,a,a]
    V = torch.einsum('ij,nctujk,lk->nctuil', BT, d, BT)           # B^T d B
    U = torch.einsum('ij,kcjl,ml->kcim', G, w.to(dt), G)          # G g G^T
    M = torch.zeros(N, K, V.shape[2], V.shape[3], a, a, dtype=dt)
    for c0 in range(0, C, 8):                                     # fp32 accumulation over channel chunks, as the K loop does
        M += torch.einsum('kcim,nctuim->nktuim', U[:, c0:c0 + 8], V[:, c0:c0 + 8])
    Y = torch.einsum('ij,nktujl,ml->nktuim', AT, M, AT)           # A^T M A
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)

def direct32(x, w):
    N, C, H, W = x.shape
    y = torch.zeros(N, w.shape[0], H, W)
    for c0 in range(0, C, 8):
        y += F.conv2d(x[:, c0:c0 + 8], w[:, c0:c0 + 8], padding=1)
    return y

def report(tag, x, w):
    ref = F.conv2d(x.double(), w.double(), padding=1)
    out = {}
    for name, y in (('direct', direct32(x, w)), ('F(2,3)', wino(x, w, 2)), ('F(4,3)', wino(x, w, 4))):
        e = (y.double() - ref)
        out[name] = (float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()))
    d = out['direct'][0]
    print('%-34s  ' % tag + '  '.join('%s rel-L2 %.2e max %.2e (%.1fx)' % (k, v[0], v[1], v[0] / d) for k, v in out.items()), flush=True)
    return out

if __name__ == '__main__':
    torch.set_num_threads(16)
    rows = []
    for (n, c, k, h) in ((3, 128, 256, 64), (3, 256, 128, 64), (3, 256, 512, 32), (3, 512, 512, 16), (3, 512, 512, 8)):
        w = torch.randn(k, c, 3, 3) * math.sqrt(2.0 / (9 * c))
        x = F.leaky_relu(torch.randn(n, c, h, h), 0.2)
        rows.append(report('fwd  n%d %d->%d @%d' % (n, c, k, h), x, w))
        g = torch.randn(n, c, h, h) * torch.exp(1.5 * torch.randn(n, c, 1, 1))          # heavy-tailed channel scales: adjoint-like operand
        rows.append(report('bwd-like n%d %d->%d @%d' % (n, c, k, h), g, w))
    r2 = sum(r['F(2,3)'][0] / r['direct'][0] for r in rows) / len(rows)
    r4 = sum(r['F(4,3)'][0] / r['direct'][0] for r in rows) / len(rows)
    print('mean error ratio to the direct fp32 conv: F(2,3) %.1fx, F(4,3) %.1fx' % (r2, r4))
