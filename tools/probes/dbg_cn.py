import sys, os
ROOT="/root/repo"
for p in (ROOT, ROOT+"/tensorflow-image-models_amd", ROOT+"/tests"): sys.path.insert(0,p)
import model_checks as mc, test_architectures
r = mc.compare_model("convnext_test_model", batch=2, features=True)
for k,v in r.items(): print(k, v)
