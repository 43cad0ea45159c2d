"""UNet2DModel caches its (state-dict key, tensor) list between forwards; every way of replacing a Parameter object must
drop that cache (a parent's load_state_dict(assign=True), attribute assignment, register_parameter,
torch.func.functional_call and a swapped sub-module used to leave the plan running the OLD weights without an error)."""
import pytest
import torch
from torch import nn

import drivescenegen_amd as d
from tests.common import CFG1, noisy_inputs, synth_weights


class _Wrapper(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net


def _mutations():
    def assign_through_parent(net):
        w = _Wrapper(net)
        sd = {k: v.detach().clone() * 1.5 for k, v in w.state_dict().items()}
        w.load_state_dict(sd, assign=True)

    def setattr_parameter(net):
        net.conv_in.weight = nn.Parameter(net.conv_in.weight.detach() * 1.5)

    def register(net):
        net.conv_out.register_parameter("bias", nn.Parameter(net.conv_out.bias.detach() + 0.25))

    def swap_module(net):
        old = net.conv_norm_out
        new = nn.GroupNorm(old.num_groups, old.num_channels, eps=old.eps).to(old.weight.device)
        with torch.no_grad():
            new.weight.copy_(old.weight * 1.5)
            new.bias.copy_(old.bias)
        net.conv_norm_out = new
    return [assign_through_parent, setattr_parameter, register, swap_module]


@pytest.mark.parametrize("mutate", _mutations(), ids=lambda f: f.__name__)
def test_cache_validity_check_sees_every_replacement(mutate):
    net = synth_weights(d.UNet2DModel(**CFG1))
    net._build_plan_items()
    assert net._plan_items_valid()
    mutate(net)
    assert not net._plan_items_valid()
    net._build_plan_items()
    assert net._plan_items_valid()
    assert [k for k, _ in net._plan_items] == list(net.state_dict().keys())
    live = dict(net.state_dict(keep_vars=True))
    assert all(t is live[k] for k, t in net._plan_items)


@pytest.mark.gpu
@pytest.mark.parametrize("mutate", _mutations(), ids=lambda f: f.__name__)
def test_forward_runs_the_replaced_weights(mutate):
    """the plan's output after a replacement == a fresh model built from the same state dict (bitwise), != the old output"""
    net = synth_weights(d.UNet2DModel(**CFG1)).to("cuda:0").eval().requires_grad_(False)
    x = noisy_inputs(CFG1, 2).to("cuda:0")
    before = net(x, 37).sample.clone()
    mutate(net)
    net.requires_grad_(False)        # (a fresh nn.Parameter asks for gradients: stay on the inference plan)
    after = net(x, 37).sample
    fresh = d.UNet2DModel(**CFG1).to("cuda:0").eval().requires_grad_(False)
    fresh.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()})
    want = fresh(x, 37).sample
    assert not torch.equal(after, before)
    assert torch.equal(after, want)


@pytest.mark.gpu
def test_functional_call_uses_the_given_weights():
    net = synth_weights(d.UNet2DModel(**CFG1)).to("cuda:0").eval().requires_grad_(False)
    x = noisy_inputs(CFG1, 1).to("cuda:0")
    base = net(x, 5).sample.clone()
    other = {k: (v.detach() * 1.25) for k, v in net.state_dict().items()}
    with torch.no_grad():
        got = torch.func.functional_call(net, other, (x, 5)).sample.clone()
    fresh = d.UNet2DModel(**CFG1).to("cuda:0").eval().requires_grad_(False)
    fresh.load_state_dict(other)
    assert torch.equal(got, fresh(x, 5).sample)
    assert torch.equal(net(x, 5).sample, base)      # and the module's own weights are back afterwards
