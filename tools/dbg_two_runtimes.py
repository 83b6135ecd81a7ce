import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
order = sys.argv[1]
import numpy as np
def mine():
    import mi355cd, mi355_synth as synth
    v, t = synth.soup(1000, 0.1, 1)
    with mi355cd.CollisionDetector(v, t) as cd:
        print("mi355cd ok", cd.self_collide()[1])
def tor():
    import torch
    torch.cuda.set_device(0)
    print("torch ok", torch.zeros(4, device="cuda").sum().item())
for o in order:
    try:
        (mine if o == "m" else tor)()
    except Exception as e:
        print("FAILED", o, repr(e)[:200])
maps = open("/proc/self/maps").read()
for l in sorted({x.split()[-1] for x in maps.splitlines() if "amdhip" in x or "hsa-runtime" in x or "rccl" in x}): print(l)
