"""stride-2 dgrad at full backbone sizes against torch (debug aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'; B = int(os.environ.get('B', 32))
SH = [('l2.0c2', 128, 128, 3, 2, 1, 120, 160), ('l2.0ds', 256, 512, 1, 2, 0, 120, 160),
      ('l3.0c2', 256, 256, 3, 2, 1, 60, 80), ('l3.0ds', 512, 1024, 1, 2, 0, 60, 80),
      ('l4.0c2', 512, 512, 3, 2, 1, 30, 40), ('l4.0ds', 1024, 2048, 1, 2, 0, 30, 40)]
for name, ci, co, k, s, p, H, W in SH:
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    torch.manual_seed(0)
    dy = torch.randn(B, OH, OW, co, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, ci, k, k, device=dev) / (co * k * k) ** 0.5).to(torch.bfloat16)
    wd = w.permute(1, 2, 3, 0).reshape(ci, k * k, co).contiguous()
    for use_res in (False, True):
        dx = torch.full((B, H, W, ci), float('nan'), device=dev, dtype=torch.bfloat16)
        res = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16) if use_res else None
        msk = (torch.rand(B, H, W, ci, device=dev) > 0.5).to(torch.bfloat16) if use_res else None
        hip.conv2d(1, dy, wd, dx, B, OH, OW, co, co, H, W, ci, k, k, s, s, p, p, res=res, relu_mask=msk)
        torch.cuda.synchronize()
        ref = torch.nn.grad.conv2d_input((B, ci, H, W), w.float(), dy.float().permute(0, 3, 1, 2), stride=s, padding=p).permute(0, 2, 3, 1)
        if use_res:
            ref = (ref + res.float()) * msk.float()
        err = ((dx.float() - ref).abs().max() / ref.abs().max()).item()
        print(name, 'res' if use_res else 'plain', 'rel err', err, flush=True)
