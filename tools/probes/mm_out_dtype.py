import torch
a = torch.randn(3648, 768, device='cuda').to(torch.bfloat16); b = torch.randn(3648, 3072, device='cuda').to(torch.bfloat16)
try:
    c = torch.mm(a.t(), b, out_dtype=torch.float32)
    print('mm out_dtype ok', c.dtype, (c - a.float().t() @ b.float()).abs().max().item())
except Exception as e:
    print('mm out_dtype failed:', str(e)[:200])
try:
    acc = torch.zeros(768, 3072, device='cuda')
    c = torch.addmm(acc, a.t(), b, out_dtype=torch.float32)
    print('addmm out_dtype ok', c.dtype)
except Exception as e:
    print('addmm out_dtype failed:', str(e)[:200])
try:
    acc = torch.ones(768, 3072, device='cuda')
    torch.addmm(acc, a.t(), b, out_dtype=torch.float32, out=acc)
    print('addmm out= ok', (acc - 1 - a.float().t() @ b.float()).abs().max().item())
except Exception as e:
    print('addmm out= failed:', str(e)[:200])
