"""Does the library's C x C GEMM of the step (M = 41 984 rows: 1 312 tiles of 256 x 256 = 5.125 rounds on 256 CUs) pay for its last, 1/8-filled round?
Time against M around the step's shape.  (A batched form of the three projections -- torch.bmm of 3 x 41 984 x 2 048 x 2 048 -- faults inside the library
on this image: 'Memory access fault by GPU node', so that route is closed.)"""
import torch, json, sys
sys.path.insert(0, '/root/repo')
from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()
dev='cuda'
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
out=[]
W=torch.randn(2048,2048,device=dev,dtype=torch.bfloat16)*0.02
W3=torch.randn(3,2048,2048,device=dev,dtype=torch.bfloat16)*0.02
for M in (40960, 41984, 43008, 45056, 49152):
    x=torch.randn(M,2048,device=dev,dtype=torch.bfloat16)
    ms=t(lambda: torch.nn.functional.linear(x,W))
    print("M",M,flush=True)
    out.append({"M":M,"tiles":(M+255)//256*8,"rounds":(M+255)//256*8/256,"ms":round(ms,4),"PF":round(2*M*2048*2048/ms/1e12,3)})
for o in out: print(json.dumps(o))
