"""-m gpu: the host -> device input pipeline (fabric_amd/input_pipeline.py; reference train.py:83-85).  Every batch must arrive
intact and in order whatever the ring depth, for pinned and pageable sources, while the consumer is still busy with earlier
batches on its own stream (slot reuse is ordered by events, not by luck), and a training loop fed through it must produce
exactly the parameters of the loop fed with resident tensors."""
import pytest
import torch

from fabric_amd import BiDateNet
from fabric_amd.input_pipeline import DeviceFeeder
from fabric_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('depth', [2, 3, 5])
@pytest.mark.parametrize('pinned', [True, False])
def test_batches_arrive_intact_and_in_order(depth, pinned):
    n, shape = 11, (4, 13, 64, 64)
    g = torch.Generator().manual_seed(3)
    host = []
    for i in range(n):
        a = torch.randn(shape, generator=g) + i
        b = torch.randn(shape, generator=g) - i
        y = torch.randint(0, 2, (4, 64, 64), generator=g, dtype=torch.uint8)
        host.append(tuple(t.pin_memory() if pinned else t for t in (a, b, y)))
    feeder = DeviceFeeder('cuda', depth=depth, stage_threads=3)
    s = torch.cuda.Stream()
    got = []
    with torch.cuda.stream(s):
        for k, (a, b, y) in enumerate(feeder(iter(host))):
            assert a.is_cuda and a.shape == shape and y.dtype == torch.uint8
            torch.cuda._sleep(3_000_000)                       # the consumer lags: later copies must not overwrite what it still reads
            got.append((a.double().sum() + 2 * b.double().sum() + y.double().sum()).clone())
    torch.cuda.synchronize()
    for k, (a, b, y) in enumerate(host):
        want = a.double().sum() + 2 * b.double().sum() + y.double().sum()
        assert abs(got[k].item() - want.item()) <= 1e-6 * abs(want.item()) + 1e-6, k
    assert len(got) == n
    assert list(feeder(iter([]))) == []


def test_fed_training_equals_resident_training():
    torch.manual_seed(5)
    B, steps = 4, 6
    batches = [(torch.randn(B, 13, 32, 32), torch.randn(B, 13, 32, 32), (torch.rand(B, 32, 32) < 0.2).to(torch.uint8)) for _ in range(steps)]
    sd0 = {k: v.clone() for k, v in BiDateNet(13, 2).state_dict().items()}
    out = []
    for fed in (False, True):
        model = BiDateNet(13, 2, precision='bf16')
        model.load_state_dict(sd0)
        ts = TrainStep(model.cuda().train(), lr=0.05)
        with torch.cuda.stream(ts.stream()):
            if fed:
                for b in DeviceFeeder('cuda')(iter(batches)):
                    ts.step(*b)
            else:
                for b in batches:
                    ts.step(*(t.cuda() for t in b))
        torch.cuda.synchronize()
        out.append(ts.flat_params.cpu().clone())
    assert torch.equal(out[0], out[1])


def test_feeder_refuses_a_cpu_device():
    with pytest.raises(RuntimeError, match='ROCm device'):
        DeviceFeeder('cpu')
