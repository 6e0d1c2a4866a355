import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, ctypes as C
import video_coding_amd as hvc
from video_coding_amd.synth import synth_pixels
import torch
W,H=1920,1080
ctx=hvc.Context(0)
jp=[]
for f in range(4):
    y=synth_pixels(1+f,1088,1920)[:H]; u=synth_pixels(20+f,544,960)[:540]; v=synth_pixels(30+f,544,960)[:540]
    jp.append(ctx.jpeg_encode(y,u,v,W,H,420,75))
for n in (1, 8, 64):
    jpegs=[jp[i%4] for i in range(n)]
    info=hvc.hvc.jpeg_read_header(jpegs[0])
    ptrs=(C.c_void_p*n)(*[C.cast(C.c_char_p(j),C.c_void_p) for j in jpegs]); sizes=(C.c_size_t*n)(*[len(j) for j in jpegs])
    out=torch.empty((n,info.coef_count),dtype=torch.int16,device="cuda"); used=C.c_int(0)
    call=lambda: hvc.hvc._chk(hvc.lib().hvc_jpeg_entropy_decode_gpu(ctx._h,ptrs,sizes,n,out.data_ptr(),info.coef_count,1,C.byref(info),C.byref(used)))
    for _ in range(3): call()
    t0=time.perf_counter(); k=10
    for _ in range(k): call()
    dt=(time.perf_counter()-t0)/k
    print("n=%d: %.3f ms per call, %.1f Mpixel/s, used_gpu=%d" % (n, dt*1e3, n*W*H/dt/1e6, used.value))
