"""The SURVEY 8f "next" rows (noise masks, apply_mask, NASC) at BASELINE configs[1] size
(4 x 500 000 x 2000, 4 G samples) through size-independent properties with closed-form answers:
fields that are constant along ping_time or range, injected impulse / attenuated pings, dB-offset
linearity.  The oracle cannot run at this size; small-size parity is in test_gpu_masks*.py / test_gpu_nasc.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C, P, S = 4, 500_000, 2000
STEP = 0.2  # metres per sample


@pytest.fixture(scope="module")
def env():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import gc

    gc.collect()
    torch.cuda.empty_cache()  # blocks cached by earlier full-size modules
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2**30:
        pytest.skip("needs ~200 GB of free HBM")
    from echopype_amd import ops

    prof = -60.0 - 20.0 * torch.linspace(0, 1, S, dtype=torch.float64, device="cuda")  # depends on range only
    sv = prof.expand(C, P, S).contiguous()
    depth = (1.0 + STEP * torch.arange(S, dtype=torch.float64, device="cuda")).expand(C, P, S).contiguous()
    return torch, ops, sv, depth, prof


def _count(torch, m):
    return int(sum(int(m[c].sum(dtype=torch.int64)) for c in range(m.shape[0])))


def test_masks_on_a_ping_invariant_field(env):
    torch, ops, sv, depth, prof = env
    # no ping differs from its neighbours: no impulse, no attenuation
    up = ops.range_bin_smooth(sv[:1], nper=25)
    exp_up = 10 * torch.log10((10 ** (prof / 10)).reshape(-1, 25).mean(dim=1)).repeat_interleave(25)
    assert float((up[0, 12345] - exp_up).abs().max()) < 1e-10 and float((up[0, -1] - exp_up).abs().max()) < 1e-10
    up_all = torch.empty_like(sv)
    for c in range(C):
        up_all[c] = ops.range_bin_smooth(sv[c:c + 1], nper=25)[0]
    assert _count(torch, ops.impulse_mask(up_all, 2, 10.0)) == 0
    del up_all
    assert _count(torch, ops.attenuated_mask(sv, depth, 100.0, 200.0, 15, -6.0)) == 0


def test_injected_impulse_and_attenuated_pings_are_the_only_ones_masked(env):
    torch, ops, sv, depth, prof = env
    sv2 = sv.clone()
    spikes = torch.arange(1000, P - 1000, 5000, device="cuda")
    faded = torch.arange(3500, P - 1000, 5000, device="cuda")
    sv2[:, spikes] += 30.0
    sv2[:, faded] -= 12.0
    up = torch.empty_like(sv2)
    for c in range(C):
        up[c] = ops.range_bin_smooth(sv2[c:c + 1], nper=25)[0]
    m = ops.impulse_mask(up, 2, 10.0)
    del up
    assert _count(torch, m) == C * spikes.numel() * S
    assert bool(m[:, spikes].all())
    m = ops.attenuated_mask(sv2, depth, 100.0, 200.0, 15, -6.0)
    assert _count(torch, m) == C * faded.numel() * S
    assert bool(m[:, faded].all())
    # apply_mask: keep = not masked; exactly the faded pings become the fill value, the rest is untouched
    keep = (m == 0).to(torch.uint8)
    out = ops.apply_mask(sv2, keep, fill_value=-999.0)
    assert bool((out[:, faded] == -999.0).all())
    assert _count(torch, out == -999.0) == C * faded.numel() * S
    assert bool((out[:, spikes] == sv2[:, spikes]).all())
    again = ops.apply_mask(out, keep, fill_value=-999.0)  # idempotent
    assert all(bool((again[c] == out[c]).all()) for c in range(C))


def test_transient_pool_of_a_range_invariant_field_is_the_field(env):
    torch, ops, sv, depth, prof = env
    flat = torch.full((1, P, S), -70.0, dtype=torch.float64, device="cuda")
    flat[0, ::7, ::3] = float("nan")  # holes do not change a nanmean of equal values
    pooled, mask = ops.pool_sv(flat, 100, 25, 50, threshold=12.0)
    assert bool(torch.isnan(pooled[0, :, :100]).all())
    assert float((pooled[0, :, 100:] + 70.0).abs().max()) < 1e-10
    assert _count(torch, mask) == 0
    flat[0, 250_000:250_003, 1000:1010] = -40.0  # a 3 ping x 10 sample transient, 30 dB above
    _, mask = ops.pool_sv(flat, 100, 25, 50, threshold=12.0, want_pooled=False)
    assert _count(torch, mask) == 30 and bool(mask[0, 250_000:250_003, 1000:1010].all())


def test_nasc_closed_form_and_linearity(env):
    torch, ops, sv, depth, prof = env
    n_d, rbin = 500, 10.0
    starts = torch.from_numpy(np.linspace(0, P, n_d + 1).astype(np.int32)).cuda()
    n_r = len(np.arange(0, float(depth.max()) + rbin, rbin)) - 1
    nasc = ops.nasc(sv, depth, starts, n_d, rbin, n_r)
    # every distance bin sees the same profile: NASC[c, d, r] = mean(sv in r) * (sum of steps in r) * 4 pi 1852^2
    d1 = depth[0, 0].cpu().numpy()
    lin = 10 ** (prof.cpu().numpy() / 10)
    b = np.floor(d1 / rbin).astype(int)
    dd = np.r_[np.diff(d1), 0.0]
    exp = np.array([lin[b == r].mean() * dd[b == r].sum() for r in range(n_r)]) * 4 * np.pi * 1852**2
    got = nasc.cpu().numpy()
    np.testing.assert_allclose(got, np.broadcast_to(exp, got.shape), rtol=1e-10)
    up10 = ops.nasc(sv + 10.0, depth, starts, n_d, rbin, n_r)
    np.testing.assert_allclose(up10.cpu().numpy(), 10.0 * got, rtol=1e-12)
