import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package(); synth = pkg.synth
dev = torch.device("cuda:0")
s = synth.shape("gk", 64, 1024, 7, 7, 512, 1, bias=True, sparsity=0.0)
w = synth.pruned_weights(s, 1)
for mode in ("eager", "graph_nowarm", "graph_warm"):
    plan = pkg.Plan(pkg.ConvDesc.from_shape(s), kernel=pkg.KERNEL_DENSE)
    plan.weight_align(w)
    x = torch.rand((s.N, s.C, s.H, s.W), device=dev)
    y = torch.zeros((s.N, s.M) + tuple(plan.out_hw), device=dev)
    if mode == "eager":
        for i in range(3):
            plan.forward(x, None, y); torch.cuda.synchronize()
            print(mode, i, "streamk", plan.stat("streamk"), "gave_up", plan.stat("streamk_gave_up"))
    else:
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        if mode == "graph_warm":
            with torch.cuda.stream(side):
                plan.forward(x, None, y)
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            print(mode, "after warm", plan.stat("streamk_gave_up"))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            plan.forward(x, None, y)
        print(mode, "after capture: streamk", plan.stat("streamk"))
        for i in range(3):
            g.replay(); torch.cuda.synchronize()
            print(mode, "replay", i, "gave_up", plan.stat("streamk_gave_up"))
    plan.close()
