"""World-size-2 gloo test of the bucketed gradient reducer (the N>1 path of bench.py) on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 8))
    reducer = GradientReducer(model, bucket_mb=0.2)          # several buckets
    assert len(reducer.buckets) >= 3
    results = []
    for step in range(2):
        reducer.zero_grad()
        g = torch.Generator().manual_seed(100 * step + rank)  # each rank owns different utterances
        x = torch.randn(16, 64, generator=g)
        model(x).pow(2).mean().backward()
        reducer.finish()
        results.append([p.grad.clone() for p in model.parameters()])
    # reference: average of the per-rank gradients computed serially
    ref_model = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 8))
    ref_model.load_state_dict(model.state_dict())
    for step in range(2):
        acc = [torch.zeros_like(p) for p in ref_model.parameters()]
        for r in range(world):
            ref_model.zero_grad()
            g = torch.Generator().manual_seed(100 * step + r)
            ref_model(torch.randn(16, 64, generator=g)).pow(2).mean().backward()
            for a, p in zip(acc, ref_model.parameters()):
                a += p.grad / world
        for got, want in zip(results[step], acc):
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)
    for p in model.parameters():                              # gradients still alias the communication buckets
        assert p.grad.data_ptr() >= reducer.flat[reducer.bucket_of[p]].data_ptr()
    with open(os.path.join(tmp, f'ok{rank}'), 'w') as f:
        f.write('ok')
    dist.destroy_process_group()


def test_bucketed_all_reduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()
