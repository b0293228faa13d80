import torch, sys
a = torch.load("/tmp/diag_%s.pt" % sys.argv[1]); b = torch.load("/tmp/diag_%s.pt" % sys.argv[2])
fl = [(i, int((x != y).sum()), x.numel()) for i, (x, y) in enumerate(zip(a["masks"], b["masks"])) if (x != y).any()]
print("mask flips between runs:", fl, "total activations", sum(x.numel() for x in a["masks"]))
worst = sorted(((a["grads"][k].double() - b["grads"][k].double()).norm().item() / (b["grads"][k].double().norm().item() + 1e-30), k) for k in a["grads"])[-5:]
print("worst grad rel L2 between runs:", worst)
print("out max diff", (a["out"] - b["out"]).abs().max().item())
