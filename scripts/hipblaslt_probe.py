import torch
M=65536
for N,K in [(2304,768),(768,768),(3072,768),(768,3072)]:
    A=torch.randn((M,K),device="cuda").bfloat16(); W=(torch.randn((N,K),device="cuda")*0.05).bfloat16(); b=torch.randn(N,device="cuda").bfloat16()
    for i in range(3): torch.nn.functional.linear(A,W,b)
    torch.cuda.synchronize()
