import torch
for dev in ("cpu", "cuda"):
    d = torch.tensor([[1., 2.], [1., 0.], [0., 2.], [5., 5.]], dtype=torch.float64, requires_grad=True, device=dev)
    o = torch.segment_reduce(d, "max", lengths=torch.tensor([3, 0, 1], device=dev), axis=0, unsafe=True, initial=-3.4e38)
    o.backward(torch.tensor([[1., 1.], [7., 7.], [1., 1.]], dtype=torch.float64, device=dev))
    print(dev, d.grad.tolist())
    x = torch.tensor([[1., 2.], [3., 0.]], dtype=torch.float64, requires_grad=True, device=dev)
    idx = torch.tensor([0, 0, 1], device=dev)
    o = torch.segment_reduce(x[idx], "max", lengths=torch.tensor([3], device=dev), axis=0, unsafe=True, initial=-3.4e38)
    o.backward(torch.tensor([[1., 1.]], dtype=torch.float64, device=dev))
    print(dev, "dup source:", x.grad.tolist())
