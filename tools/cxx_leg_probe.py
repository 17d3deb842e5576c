import json, os, sys, time, torch
sys.path.insert(0, "/root/repo")
os.environ["NVBIO_SHIM_TRACE"] = "1"
import bench, nvbio_amd as nvb
from nvbio_amd import workloads as W, pipeline as P, aligner as AL, select as SEL
dev = torch.device("cuda:0")
ng, n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000, int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
g = torch.Generator(device=dev); g.manual_seed(3)
text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
fmi = W.build_fm_index(text).with_dimer()
gw = W._pack_chunked(text, 2, True)
sym, pos, _ = P.make_reads(text, n, 100, seed=4)
packed = P.pack_read_streams(sym)
names = SEL.pack_names(["r%d" % i for i in range(n)], dev)
prm = AL.Params(hits_stride=16, batch_size=n)
r = AL.best_approx(fmi, None, sym, gw, ng, prm, names=names, packed=packed)
t0 = time.perf_counter(); r = AL.best_approx(fmi, None, sym, gw, ng, prm, names=names, packed=packed); torch.cuda.synchronize()
print("python driver ms", (time.perf_counter() - t0) * 1e3)
print(json.dumps(bench.cxx_driver_leg(None, dev, fmi, sym, packed, gw, ng, names, prm, r["best"], r["mapq"])))
