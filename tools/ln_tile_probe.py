# LayerNorm-folded GEMM on every persistent tile shape, 20 launches each: outputs must be bit-identical to the 256x256 tile
import sys, math, os
sys.path.insert(0,'tests'); sys.path.insert(0,'tensorflow-image-models_amd'); sys.path.insert(0,'.')
import numpy as np, torch
import hip_ops as Hh
from tfimm.engine import pack
r = np.random.default_rng(1)
M,K,N = 1000,768,2304
x = (r.standard_normal((M,K))*1.5+0.3).astype(np.float32)
gam = r.uniform(0.5,1.5,K).astype(np.float32); bet=(0.3*r.standard_normal(K)).astype(np.float32)
w = (r.standard_normal((K,N))/math.sqrt(K)).astype(np.float32); b=r.standard_normal(N).astype(np.float32)
wf=(w*gam[:,None]).astype(np.float32); bf=(bet@w+b).astype(np.float32)
wt,bvec=pack.pack_dense(wf,bf); c1=pack.pack_ln_c1(wt,N,K)
xd=Hh.dev_bf16(x); st=Hh.row_stats(xd,1e-6)
wtd=Hh.dev_bits(wt); bd=Hh.dev_f32(bvec); c1d=Hh.dev_bits(c1)
ref=Hh.gemm(xd,wtd,N,K,bias=bd,act="gelu",tile_hint=21,ln_stats=st,ln_c1=c1d); Hh.sync(); ref=ref.float().cpu().numpy()
for t in (21,22,23,24,25,26,27,29,0):
    nb=0
    for rep in range(20):
        got=Hh.gemm(xd,wtd,N,K,bias=bd,act="gelu",tile_hint=t,ln_stats=st,ln_c1=c1d)
        Hh.sync(); g=got.float().cpu().numpy()
        nb += int((np.abs(g-ref)>0).any())
    print("dbg",os.environ.get("TFIMM_GEMM_DBG"),"tile",t,"bad runs of 20:",nb)
