import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gpv1_amd.decode as dec
orig = torch.cuda.CUDAGraph.replay
cnt = [0]
def replay(self):
    orig(self); torch.cuda.synchronize(); cnt[0] += 1; print('replayed', cnt[0], flush=True)
torch.cuda.CUDAGraph.replay = replay
ostep = dec.GreedyKVDecoder._step
def step(self, t):
    cap = torch.cuda.is_current_stream_capturing()
    ostep(self, t)
    if not cap:
        torch.cuda.synchronize()
    print('step', t, 'B', self.B, 'capturing' if cap else 'eager', flush=True)
dec.GreedyKVDecoder._step = step
sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
bench.main()
