"""Dev probe: eager vs hipGraph replay vs two-stream replay of the cold query."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
i = [0]
def eager():
    i[0] += 1
    pipe.query(tok, 593, seed=i[0], k=100)
print("eager ms/step", timeit(eager))
cq = pipe.capture_query(tok.shape, 593, seed=0, k=100)
cq.tokens.copy_(tok)
print("graph ms/step", timeit(lambda: cq.replay()))
a = cq.replay().clone(); torch.cuda.synchronize(); b = cq.replay().clone(); torch.cuda.synchronize()
print("fresh stream per replay:", not torch.equal(a, b), a[:3, 3].tolist(), b[:3, 3].tolist())
# eager result with the same effective seed equals the replay's
cnt = int(cq.counter.item())
ref = pipe.query(tok, 593, seed=cnt, k=100)[0]
print("graph == eager for the same seed:", torch.equal(ref, b))
cq2 = pipe.capture_query(tok.shape, 593, seed=1 << 32, k=100)
cq2.tokens.copy_(tok)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def two():
    with torch.cuda.stream(s1): cq.replay()
    with torch.cuda.stream(s2): cq2.replay()
print("2-stream graph ms/step", timeit(two, 100) / 2)
