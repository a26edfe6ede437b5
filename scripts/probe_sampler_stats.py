"""dev probe: iterations per epoch of the surface sampler on the bench field."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
for seed in (1, 2, 3):
    s, a, st = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=seed)
    torch.cuda.synchronize()
    print(seed, [(int(r[0]), int(r[1]), int(r[3])) for r in st.cpu()], "(iterations, still invalid, m_last) per epoch")
