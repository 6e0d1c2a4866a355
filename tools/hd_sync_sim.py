#!/usr/bin/env python3
"""Why the GPU Huffman reader needs about eight passes on the bench content: a CPU simulation of the
synchronisation it relies on.  One 1080p bench frame is parsed truly; then decoders are started at 1024-bit
boundaries with the reader's guessed state (k = 0, b = 0) and the distance to the true parse is measured,
once for the full state (bit position, zig-zag index k, block-in-MCU b) and once ignoring b.
Uses the oracle's encoder only to make the frame (a development tool, not product or test code).

    python tools/hd_sync_sim.py
"""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import orc
from video_coding_amd.synth import synth_pixels
import video_coding_amd as hvc
W,H=1920,1080
y=synth_pixels(1,1088,1920)[:H]; u=synth_pixels(20,544,960)[:540]; v=synth_pixels(30,544,960)[:540]
jpg=orc.encode_yuv(y,u,v,W,H,420,75)
info=hvc.hvc.jpeg_read_header(jpg)
ecs=jpg[info.ecs_offset:-2].replace(b'\xff\x00',b'\xff')
bits=np.unpackbits(np.frombuffer(ecs,dtype=np.uint8))
# tables: default Annex K via oracle? build from DHT in file
def parse_dht(j):
    i=2; tabs={}
    while i < len(j):
        if j[i]==0xff and j[i+1]==0xc4:
            ln=(j[i+2]<<8)|j[i+3]; tc=j[i+4]; L=list(j[i+5:i+21]); vals=list(j[i+21:i+2+ln]); tabs[(tc>>4,tc&15)]=(L,vals); i+=2+ln
        elif j[i]==0xff and j[i+1]==0xda: break
        elif j[i]==0xff and j[i+1] in (0xd8,): i+=2
        elif j[i]==0xff: ln=(j[i+2]<<8)|j[i+3]; i+=2+ln
        else: i+=1
    return tabs
tabs=parse_dht(jpg)
def build(L,vals):
    d={}; code=0; k=0
    for ln in range(1,17):
        for _ in range(L[ln-1]):
            d[(ln,code)]=vals[k]; k+=1; code+=1
        code<<=1
    return d
T={k:build(*v) for k,v in tabs.items()}
B=6; b2c=[0,0,0,0,1,1]
N=len(bits)
def step(p,k,b):
    t=T[(0 if k==0 else 1, b2c[b])]
    code=0
    for ln in range(1,17):
        if p+ln>N: return None
        code=(code<<1)|int(bits[p+ln-1])
        if (ln,code) in t:
            val=t[(ln,code)]; break
    else:
        return (p+1,k,b)
    p+=ln
    if k==0:
        p+=val; return (p,1,b)
    run,size=val>>4,val&15
    p+=size
    if size==0 and run==0: return (p,0,(b+1)%B)
    k+=run
    if k>=64: return (p,0,(b+1)%B)
    k+=1
    if k==64: return (p,0,(b+1)%B)
    return (p,k,b)
# true parse: set of states at symbol starts
true={}
st=(0,0,0); nsym=0
while st and st[0] < N-32:
    true[st[0]]=(st[1],st[2]); st=step(*st); nsym+=1
print("bits",N,"symbols",nsym)
import random
random.seed(1)
dist=[]
for j in random.sample(range(1,N//1024-8),150):
    st=(j*1024,0,0); start=st[0]
    while st and st[0] < N-64:
        if st[0] in true and true[st[0]]==(st[1],st[2]): break
        st=step(*st)
    dist.append(st[0]-start)
dist=np.array(dist)
print("sync distance bits: median %d  mean %d  p90 %d  max %d" % (np.median(dist), dist.mean(), np.percentile(dist,90), dist.max()))
# bit/k sync only (ignoring b)
dist2=[]
for j in random.sample(range(1,N//1024-8),150):
    st=(j*1024,0,0); start=st[0]
    while st and st[0] < N-64:
        if st[0] in true and true[st[0]][0]==st[1]: break
        st=step(*st)
    dist2.append(st[0]-start)
dist2=np.array(dist2)
print("ignoring b: median %d mean %d p90 %d max %d" % (np.median(dist2), dist2.mean(), np.percentile(dist2,90), dist2.max()))
