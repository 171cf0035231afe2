import sys, zlib, numpy as np, torch
sys.path.insert(0,'.')
import pure_zlib_amd as P
ctx=P.Context(0); ctx.set_stream(0)
dev=torch.device('cuda',0)
for nb in [300<<20, (513<<20)+7, (1<<30)+3, (2<<30)+11, (4<<30)+5, (6<<30)]:
    buf=torch.empty(nb,dtype=torch.uint8,device=dev)
    g=torch.Generator(device=dev); g.manual_seed(1)
    ch=1<<28
    for lo in range(0,nb,ch):
        hi=min(nb,lo+ch); buf[lo:hi]=torch.randint(0,256,(hi-lo,),dtype=torch.uint8,device=dev,generator=g)
    res=torch.zeros(1,dtype=torch.int32,device=dev)
    ctx.adler32_device(buf.data_ptr(), nb, res.data_ptr())
    got=int(res.cpu().numpy().view(np.uint32)[0])
    exp=1
    for lo in range(0,nb,ch): exp=zlib.adler32(buf[lo:lo+ch].cpu().numpy().tobytes(),exp)
    print(nb, hex(got), hex(exp), got==exp, flush=True)
    del buf
