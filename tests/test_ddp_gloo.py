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


class _SinkLinearFn(torch.autograd.Function):
    """A CPU stand-in for the HIP backward kernels in gradient-sink mode: the weight / bias gradients are ADDED straight into the
    pre-allocated ``param.grad`` (a view into a communication bucket) and autograd is handed ``None`` for them."""

    @staticmethod
    def forward(ctx, x, w, b, rt):
        ctx.save_for_backward(x, w)
        ctx.params, ctx.sink = (w, b), rt.sink                 # captured at forward, as functional.py does
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dw, db = dy.t() @ x, dy.sum(0)
        if ctx.sink:
            pw, pb = ctx.params
            pw.grad.add_(dw)
            pb.grad.add_(db)
            dw = db = None
        return dy @ w, dw, db, None


class _SinkNet(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        from ubisoft_laforge_daft_exprt_amd.ops import Runtime
        torch.manual_seed(seed)
        self.layers = torch.nn.ModuleList([torch.nn.Linear(32, 200), torch.nn.Linear(200, 200), torch.nn.Linear(200, 4)])
        self.runtime = Runtime('f32')

    def forward(self, x):
        for i, l in enumerate(self.layers):
            x = _SinkLinearFn.apply(x, l.weight, l.bias, self.runtime)
            if i < 2:
                x = torch.relu(x)
        return x


def _sink_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ubisoft_laforge_daft_exprt_amd.ddp import GradientReducer
    model = _SinkNet(seed=10 + rank)                           # replicas START DIFFERENT: construction must broadcast rank 0's
    reducer = GradientReducer(model, bucket_mb=0.1, grad_sink=True)
    assert model.runtime.sink and len(reducer.buckets) >= 2
    ref = _SinkNet(seed=10)                                    # rank 0's initialisation, plain autograd
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a, b), 'parameters were not broadcast from rank 0'
    accum = 3
    for step in range(2):
        reducer.zero_grad()
        for k in range(accum):                                 # micro-batch accumulation: only the last backward communicates
            g = torch.Generator().manual_seed(1000 * step + 10 * k + rank)
            x = torch.randn(8, 32, generator=g)
            with reducer.accumulate(sync=(k == accum - 1)):
                (model(x).pow(2).mean() / accum).backward()    # hooks fire on UNDEFINED gradients here
            assert (len(reducer.works) > 0) == (k == accum - 1)
        reducer.finish()
        want = [torch.zeros_like(p) for p in ref.parameters()]
        for r in range(world):
            for k in range(accum):
                ref.zero_grad()
                g = torch.Generator().manual_seed(1000 * step + 10 * k + r)
                (ref(torch.randn(8, 32, generator=g)).pow(2).mean() / accum).backward()   # ref.runtime.sink is False: plain autograd
                for a, p in zip(want, ref.parameters()):
                    a += p.grad / world
        for got, w in zip(model.parameters(), want):
            assert torch.allclose(got.grad, w, rtol=1e-5, atol=1e-7)
    # a backward that skips a parameter is caught, not silently mis-reduced
    reducer.zero_grad()
    reducer.pending[0] += 1
    model(torch.randn(4, 32)).sum().backward()
    try:
        reducer.finish()
        raise AssertionError('unbalanced bookkeeping was not detected')
    except RuntimeError:
        pass
    with open(os.path.join(tmp, f'ok{rank}'), 'w') as f:
        f.write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sink_accumulation_and_broadcast_world2(tmp_path):
    port = _free_port()
    mp.spawn(_sink_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def test_feature_shards_are_equal_sized(tmp_path):
    """DistributedSampler(shuffle=False) semantics: pad by repetition, so no rank runs an extra step (and hangs the all-reduce)."""
    from tests import helpers
    from ubisoft_laforge_daft_exprt_amd.features import FeatureSet
    from ubisoft_laforge_daft_exprt_amd.hparams import HyperParams
    list_file, rows = helpers.write_synthetic_features(str(tmp_path / 'f'), n=7)
    hp = HyperParams(stats=helpers.FEATURE_STATS)
    for world in (2, 3, 4):
        shards = [FeatureSet(list_file, hp, batch_size=1, rank=r, world=world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1, [len(s) for s in shards]
        names = [[f for _, f, _ in s.rows] for s in shards]
        flat = [names[i % world][i // world] for i in range(world * len(shards[0]))]
        assert flat[:7] == [f for _, f, _ in rows] and flat[7:] == [f for _, f, _ in rows][:len(flat) - 7]
