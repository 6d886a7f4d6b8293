"""Are the squeeze sums (memset node + atomics) of a recorded plan reproducible from replay to replay?
    python tools/sums_probe.py [model] [batch] [replays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

import bench
import tfimm
from tfimm.utils.init import synthetic_weights

name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
model = tfimm.create_model(name)
model.set_weights(synthetic_weights(model, 2021))
x = bench.synthetic_batch(model.cfg, B, 7)
prog = model.program()
watch = [t for t in prog.tensors if (t.name or "").endswith(":sums") or (t.name or "") == "se_gate"]
for t in watch:
    t.keep = True
prog.outputs_keep = True
plan = prog.make_plan(B)
plan.run(x)
torch.cuda.synchronize()
ref = [plan.tensor_view(t).clone() for t in watch]
logits0 = plan.tensor_view(prog.outputs["logits"]).clone()
cap = plan.capture(x)
for r in range(n):
    cap.replay()
    torch.cuda.synchronize()
    bad = [(t.name, int((plan.tensor_view(t).view(torch.int32) != o.view(torch.int32)).sum().item())) for t, o in zip(watch, ref)]
    bad = [b for b in bad if b[1]]
    ld = float((plan.tensor_view(prog.outputs["logits"]).float() - logits0.float()).abs().max().item())
    print(f"replay {r}: logits max |diff| {ld:.3g}; differing watched tensors: {bad[:4]}{' ...' if len(bad) > 4 else ''}", flush=True)
