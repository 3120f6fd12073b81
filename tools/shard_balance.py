import sys, torch
sys.path.insert(0, '.')
from myrrix_recommender_amd import synth
dev = torch.device('cuda', 0)
p = synth.torch_problem(10_000_000, 1_000_000, 1_000_000_000, 64, dev)
for name, csr in (('users', p['r_csr']), ('items', p['c_csr'])):
    rp = csr[0]
    n = rp.shape[0] - 1
    for N in (2, 4, 8):
        per = (n + N - 1) // N
        tot = [int(rp[min(n, (r + 1) * per)] - rp[min(n, r * per)]) for r in range(N)]
        m = sum(tot) / N
        # cost model of the solve: entries + 70 per row (K3 ~ 7K cycles vs ~100 cycles per entry)
        print(name, N, 'max/mean nnz %.4f' % (max(tot) / m))
    lens = (rp[1:] - rp[:-1])
    print(name, 'max row', int(lens.max()), 'top-8 rows share of nnz %.4f' % (float(lens.topk(8).values.sum()) / float(rp[-1])))
