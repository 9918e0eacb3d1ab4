import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
dev="cuda"
for name, N, H, W, Cin, Cout in [("128@512",8,512,512,128,128),("256@256",8,256,256,256,256),("512@128",8,128,128,512,512),("320@64",36,64,64,320,320)]:
    x=(torch.randn(N,H,W,Cin,device=dev)*0.5).to(torch.bfloat16)
    w=torch.randn(Cout,Cin,3,3,device=dev)/(Cin*9)**0.5
    pw=ops.pack_conv_weight(w, torch.zeros(Cout,device=dev))
    m,r,_=ops.group_norm_stats(x,32,1e-6)
    g=torch.ones(Cin,device=dev); b=torch.zeros(Cin,device=dev)
    sc,sh=ops.group_norm_affine(m,r,g,b,Cin)
    for label,fn in [("plain",lambda: ops.conv2d(x,pw,pad=1)),("fused",lambda: ops.conv2d(x,pw,pad=1,in_norm=(sc,sh,ops.ACT_SILU))),("apply+conv",lambda: ops.conv2d(ops.group_norm_apply(x,m,r,g,b,32,ops.ACT_SILU),pw,pad=1))]:
        fn(); torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
        print(f"{name:10s} {label:12s} {dt*1e3:8.3f} ms  {2.0*N*H*W*Cin*9*Cout/dt/1e12:8.1f} TF", flush=True)
