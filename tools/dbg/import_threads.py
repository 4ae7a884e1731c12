"""Do hipModuleLoadData calls of different plans run side by side on different host threads?"""
import sys, time, threading, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package(); synth = pkg.synth
s = synth.resnet50_3x3(N=256)[3]
src = pkg.Plan(pkg.ConvDesc.from_shape(s)); src.weight_align(synth.pruned_weights(s, 1))
blob = src.export_aligned(); dblob = torch.from_numpy(blob).cuda()
print("blob", blob.size, "code", src.stat("code_bytes"))
def imp(k, out, dev):
    p = pkg.Plan(pkg.ConvDesc.from_shape(s))
    t = time.perf_counter(); p.import_aligned(dblob if dev else blob); out[k] = (time.perf_counter() - t) * 1e3
    plans.append(p)
plans = []
for dev in (False, True):
    for n in (1, 1, 3, 6):
        out = [0] * n
        t0 = time.perf_counter()
        th = [threading.Thread(target=imp, args=(k, out, dev)) for k in range(n)]
        [t.start() for t in th]; [t.join() for t in th]
        print("dev" if dev else "host", n, "threads: wall %.1f ms, per import %s" % ((time.perf_counter() - t0) * 1e3, ["%.1f" % v for v in out]))
