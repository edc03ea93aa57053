"""Greedy inference at B = 1 (BASELINE metric, second half) under rocprofv3 --kernel-trace: where the ~2000 graph nodes of one image go.
  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_greedy -o g -- python tools/prof_greedy.py run [B]
  analyse:  python tools/prof_greedy.py <trace dir>
Prints nodes / wall / GPU-busy of the last inference, the kernel families by count and the node sequence of its last decode step."""
import csv, glob, os, re, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if sys.argv[1] == 'run':
    import time, torch
    import bench
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device('cuda:0')
    import gpv1_amd.hip as hip
    from gpv1_amd.gpv import GPV
    hip.lib()
    torch.manual_seed(0)
    model = GPV(bench.make_cfg())
    for n, buf in model.named_buffers():
        if n.endswith('running_var'):
            buf.uniform_(0.5, 1.5)
    model.to(dev).eval()
    with torch.no_grad():
        images, mask, ids, attn, _ = bench.make_batch(7, B, dev)
        s = nested_tensor_from_tensor_list(images)
        for _ in range(3):
            model(s, (ids, attn), None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(s, (ids, attn), None, None)
        torch.cuda.synchronize()
        print('B=%d: %.3f ms per inference' % (B, (time.perf_counter() - t0) / 10 * 1e3))
    sys.exit(0)
d = sys.argv[1]
trace = list(csv.DictReader(open(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0])))
trace.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('gpvk::', '')
    n = re.sub(r'^void ', '', n)
    m = re.match(r'_ZN(?:4gpvk)?12_GLOBAL__N_1(\d+)([a-z0-9_]+)I(.*)', n)
    if m:
        n = m.group(2)[:int(m.group(1))] + '<' + m.group(3)[:28] + '>'
    return n[:72]
idx = [i for i, r in enumerate(trace) if 'image_to_nhwc4' in r['Kernel_Name'] or 'stem_pool' in r['Kernel_Name']]
starts = [i for j, i in enumerate(idx) if j == 0 or i - idx[j - 1] > 4]
seq = trace[starts[-2]:starts[-1]]
t0, t1 = int(seq[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in seq)
busy, end = 0, t0
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += max(0, e - max(s, end)); end = max(end, e)
print('last inference: %d nodes, wall %.3f ms, GPU busy %.3f ms, mean gap %.2f us' % (len(seq), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e3 / len(seq)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seq:
    k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('  x%-4d %8.1f us  avg %6.1f  %s' % (v[0], v[1], v[1] / v[0], k))
# decode steps: delimited by the embedding gather of the step's token
emb = [i for i, r in enumerate(seq) if 'embedding' in r['Kernel_Name'] or 'index_select' in r['Kernel_Name'] or 'indexSelect' in r['Kernel_Name']]
print('gather launches at', emb[:40])
if len(emb) >= 3:
    a, b = emb[-2], emb[-1]
    print('\nfirst decode gather at node %d (t = %.3f ms): everything before is backbone / DETR / BERT / co-attention' % (emb[-20] if len(emb) >= 20 else emb[0], (int(seq[emb[-20] if len(emb) >= 20 else emb[0]]['Start_Timestamp']) - t0) / 1e6))
    print('one decode step: %d nodes, %.1f us' % (b - a, (int(seq[b]['Start_Timestamp']) - int(seq[a]['Start_Timestamp'])) / 1e3))
    prev = int(seq[a]['Start_Timestamp'])
    for r in seq[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('   gap %5.1f  dur %5.1f  %s' % ((s - prev) / 1e3, (e - s) / 1e3, short(r['Kernel_Name'])))
        prev = e
if len(sys.argv) > 2 and sys.argv[2] == 'timeline':
    # every node of the last inference in start order: start (us from the first node), duration, gap to the previous END, queue, grid, name
    print('# timeline of the last inference')
    prev_end = t0
    for i, r in enumerate(seq):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%4d %9.1f us  dur %6.1f  gap %6.1f  q %s  grid %8s  %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'),
                                                                        r.get('Grid_Size', r.get('Grid_Size_X', '?')), short(r['Kernel_Name'])))
        prev_end = max(prev_end, e)
