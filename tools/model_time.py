"""ms per hipGraph replay of one model at one batch (native input size), e.g. with / without an environment switch:
    [TFIMM_NO_FOLD_SHORTCUT=1] python tools/model_time.py resnet50d 256"""
import sys,time,os
sys.path.insert(0,"tensorflow-image-models_amd")
import torch,tfimm
from tfimm.utils.init import synthetic_weights
name=sys.argv[1]
m=tfimm.create_model(name); m.set_weights(synthetic_weights(m,2021))
prog=m.program(); plan=prog.make_plan(int(sys.argv[2]))
x=torch.randn(int(sys.argv[2]),*m.cfg.input_size,3,device="cuda").to(torch.bfloat16)
plan.run(x); torch.cuda.synchronize(); print("eager ok", flush=True)
cap=plan.capture(x)
for _ in range(5): cap.replay()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(30): cap.replay()
torch.cuda.synchronize(); print(name, os.environ.get("TFIMM_NO_FOLD_SHORTCUT"), len(prog.ops), "ops", round((time.perf_counter()-t)/30*1e3,4), "ms")
