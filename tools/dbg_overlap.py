import os, sys, tempfile, torch
sys.path.insert(0, '/root/repo')
import torch.multiprocessing as mp
from tests.test_distributed_gpu import _worker, _free_port
if __name__ == '__main__':
    runs = []
    os.environ['GPV_TRAIN_GRAPHS'] = '0'
    for overlap in sys.argv[1:]:
        os.environ['GPV_OVERLAP'] = overlap
        out = tempfile.mkdtemp()
        mp.spawn(_worker, args=(2, _free_port(), out, 'fp32', True), nprocs=2, join=True)
        runs.append((overlap, torch.load(os.path.join(out, 'rank0.pt'))))
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            a, b = runs[i][1], runs[j][1]
            lo, hi = a['head']
            for name, (s, e) in {'backbone': (0, lo), 'detr head': (lo, hi), 'behind': (hi, a['G'].numel())}.items():
                ga, gb = a['G'][s:e].double(), b['G'][s:e].double()
                print(runs[i][0], runs[j][0], name, float((ga - gb).norm() / gb.norm()), 'losses', a['losses'][-1], b['losses'][-1])
