import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gan_amd
import test_gpu_bn_rows as T
cands = [("conv_s1",128,128,32,64,torch.bfloat16),("conv_s1",64,64,32,64,torch.bfloat16),("conv",64,128,32,128,torch.bfloat16),("convT",128,64,32,32,torch.bfloat16),
         ("convT",256,128,32,32,torch.bfloat16),("conv",256,512,64,16,torch.bfloat16),("conv",512,512,64,8,torch.bfloat16),("conv",128,256,64,64,torch.bfloat16),
         ("convT",512,512,64,8,torch.bfloat16),("convT",512,256,64,16,torch.bfloat16),("conv_s1",64,72,32,64,torch.bfloat16),("conv",256,512,64,32,torch.bfloat16),("conv_s1",512,512,32,32,torch.bfloat16),("convT",512,256,64,32,torch.bfloat16),("conv",128,256,64,32,torch.bfloat16)]
for case in cands:
    kind,cin,cout,N,H,dt = case
    net = T._pair(gan_amd, kind, cin, cout, dt)
    x = (torch.randn(N,cin,H,H,device="cuda").to(dt).contiguous(memory_format=torch.channels_last)*2+0.3)
    y0,gx0,gp0,buf0,n0 = T._run(gan_amd, net, x, False)
    y1,gx1,gp1,buf1,n1 = T._run(gan_amd, net, x, True)
    conv = [n for n in n1 if "conv_" in n and "wgrad" not in n]
    print(case[:5], "rel", T.rel(y1,y0), "rows" if "gc::bn_partial_kernel<0>" not in n1 else "SWEEP", conv, flush=True)
