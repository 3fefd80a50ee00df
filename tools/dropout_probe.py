import sys; sys.path.insert(0,"/root/repo")
import torch
from lstm_ctc_amd import ops
for rows,P,ld in ((32000,320,640),(64000,1024,2048),(32000,512,1024)):
    x=torch.randn(rows,ld,device="cuda")
    v=x[:,P:] if ld>P else x
    for _ in range(3): ops.dropout_scale(v,0.9,1,2)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dropout_scale(v,0.9,1,2)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print("dropout %dx%d: %.1f us, %.2f TB/s"%(rows,P,us,rows*P*8/us/1e6))
